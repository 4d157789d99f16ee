// LDS-blocked kernels for the scone / ebli layer shape: ONE group, identity + two value arrays on a shared
// pattern (slots: identity, S_lower, S_upper), ns = 4, channel widths 16 or 32 (MFMA) and 1 (first layer).
//
// Execution plan (built once per operator on the host, build_block_plan):
//   the output rows are cut into blocks of <= 64 consecutive rows whose gather sources (<= 128 distinct source
//   rows) are staged once per slab into LDS; the block's CSR slice is re-expressed as a padded ELL tile with
//   LOCAL source slots (uint8) so the inner loop touches LDS only.  Rows are expected in a locality order
//   (Hilbert order of the edge midpoints, scone_gcn_amd/complex.py) so a block's sources are ~2x its rows.
//
// Kernel structure: one 8-wave workgroup per CU, grid-strided over blocks with an XCD-contiguous mapping.
//   for block: ELL tile + source list -> LDS
//     for slab: [s_waitcnt vmcnt(0); s_barrier]  -> slab s has landed in buffer s&1, everybody left buffer (s+1)&1
//               issue LDS-DMA (global_load_lds, 16 B/lane, per-lane source = gather) of slab s+1 into the other buffer
//               compute slab s: each wave gathers its 8 rows x 4 trajectories with lane = (point, channel slice)
//               from the XOR-swizzled LDS image and feeds [self | lower | upper] straight into f32 MFMA against
//               weights held in registers; the activation epilogue's stores are deferred by one slab so the next
//               vmcnt(0) never waits on stores that were just issued.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <type_traits>
#include <utility>

#include "scn_internal.h"

namespace scn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- diagnostic build only (-DSCN_STAMPS): per-segment cycle sums of fwd_c32, never compiled into the product
#ifdef SCN_STAMPS
__device__ unsigned long long g_stamps[8];
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_DECL unsigned long long t0_ = 0, t1_ = 0, acc_[6] = {0, 0, 0, 0, 0, 0}
#define STAMP_ADD(i) do { STAMP(t1_); acc_[i] += t1_ - t0_; t0_ = t1_; } while (0)
#define STAMP_START() STAMP(t0_)
#define STAMP_FLUSH() do { if ((threadIdx.x & 63) == 0) { unsigned long long s_ = 0; for (int i_ = 0; i_ < 6; ++i_) { atomicAdd(&g_stamps[i_], acc_[i_]); s_ += acc_[i_]; } atomicMax(&g_stamps[6], s_); atomicAdd(&g_stamps[7], 1ull); } } while (0)
#else
#define STAMP_DECL
#define STAMP_ADD(i)
#define STAMP_START()
#define STAMP_FLUSH()
#endif

// ------------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------------
void free_block_plan(scn_conv_s* c) {
    for (void* p : c->plan.allocs)
        if (p) (void)hipFree(p);
    c->plan.allocs.clear();
    c->plan.built = false;
}

template <typename T>
static int upload(scn_conv_s* c, const std::vector<T>& h, const T** out) {
    void* d = nullptr;
    size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    SCN_HIP_TRY(hipMalloc(&d, bytes));
    c->plan.allocs.push_back(d);
    if (!h.empty()) SCN_HIP_TRY(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)d;
    return SCN_OK;
}

int build_assignments(scn_conv_s* c);

// Greedy block cut shared by the plan builder and scn_plan_refine_order: starting at row r0, take consecutive rows (< r_end)
// while the block stays within BK_R rows, BK_SRC distinct source rows (the rows themselves included when the operator has
// an identity slot) and BK_ELL_CAP padded ELL entries.  `mark` is a per-column stamp array (block id), `cur` gets the sources.
static void grow_block(const int32_t* rowptr, const int32_t* col, int r_end, bool has_id, int r0, int bid,
                       std::vector<int32_t>& mark, std::vector<int32_t>& cur, int& rows, int& w) {
    cur.clear();
    rows = 0;
    w = 0;
    while (r0 + rows < r_end && rows < BK_R) {
        const int r = r0 + rows;
        const int j0 = rowptr[r], j1 = rowptr[r + 1];
        int add = 0;
        for (int j = j0; j < j1; ++j)
            if (mark[col[j]] != bid) ++add;
        bool self_new = has_id && mark[r] != bid;
        for (int j = j0; j < j1 && self_new; ++j)
            if (col[j] == r) self_new = false;
        if (self_new) ++add;
        const int nw = std::max(w, (j1 - j0 + 1) & ~1);          // padded width, even
        if ((int)cur.size() + add > BK_SRC || nw * (rows + 1) > BK_ELL_CAP || nw > 254) break;
        for (int j = j0; j < j1; ++j)
            if (mark[col[j]] != bid) { mark[col[j]] = bid; cur.push_back(col[j]); }
        if (has_id && mark[r] != bid) { mark[r] = bid; cur.push_back(r); }
        w = nw;
        ++rows;
    }
}

int build_block_plan(scn_conv_s* c) {
    // only the scone/ebli shape gets a plan; everything else runs the generic kernels
    // single-group operators get a plan: identity + 2 value arrays is the scone/ebli layer (MFMA kernels); one value
    // array without identity is a bare shift (Bunch terms) served by the blocked SpMM
    if (c->n_groups != 1) return SCN_OK;
    const Group& G = c->g[0];
    if (G.n_vals < 1) return SCN_OK;
    const bool has_v1 = G.n_vals > 1, has_id = G.identity == 1;
    const int n_rows = c->n_rows;
    std::vector<int32_t> blk_row0, src_ptr(1, 0), src_rows, ell_ptr;
    std::vector<uint8_t> blk_rows, width, tile_w, tile_w4, tile_wu, tile_wu4, ell_slot, self_slot;
    std::vector<float2> ell_v;
    std::vector<float> block_gather;                              // per block: sum over the 8-row groups of their ELL width
    std::vector<int32_t> mark(G.n_cols, -1), local(G.n_cols, 0), cur;
    cur.reserve(BK_SRC + 64);
    int64_t total_src = 0;
    int wmax = 0;
    int bid = 0;
    const uint8_t* hint = c->block_start.empty() ? nullptr : c->block_start.data();
    for (int r0 = 0; r0 < n_rows;) {
        int rows = 0, w = 0, r_limit = n_rows;
        if (hint)                                               // a hinted boundary closes the block (limits still apply)
            for (int r = r0 + 1; r < std::min(n_rows, r0 + BK_R + 1); ++r)
                if (hint[r]) { r_limit = r; break; }
        grow_block(G.h_rowptr.data(), G.h_col.data(), r_limit, has_id, r0, bid, mark, cur, rows, w);
        if (rows == 0) return SCN_OK;   // a single row does not fit: no plan (generic kernels will be used)
        std::sort(cur.begin(), cur.end());
        for (size_t i = 0; i < cur.size(); ++i) local[cur[i]] = (int32_t)i;
        blk_row0.push_back(r0);
        blk_rows.push_back((uint8_t)rows);
        src_rows.insert(src_rows.end(), cur.begin(), cur.end());
        src_ptr.push_back((int32_t)src_rows.size());
        ell_ptr.push_back((int32_t)ell_slot.size());
        width.push_back((uint8_t)w);
        wmax = std::max(wmax, w);
        const size_t base = ell_slot.size();                 // entries are [row][w], w even, zero padded
        ell_slot.resize(base + (size_t)w * rows, 0);
        ell_v.resize(base + (size_t)w * rows, float2{0.f, 0.f});
        uint8_t tw[BK_WAVES], tw4[2 * BK_WAVES], twu[BK_WAVES], twu4[2 * BK_WAVES];
        for (int i = 0; i < BK_WAVES; ++i) tw[i] = tw4[2 * i] = tw4[2 * i + 1] = twu[i] = twu4[2 * i] = twu4[2 * i + 1] = 0;
        for (int i = 0; i < BK_R; ++i) {
            uint8_t ss = 0;
            if (i < rows) {
                const int r = r0 + i;
                const int j0 = G.h_rowptr[r], j1 = G.h_rowptr[r + 1];
                // entries the second operator takes part in come first, so the gather can stop feeding it after `twu`
                int k = 0, n_up = 0;
                for (int pass = 0; pass < 2; ++pass)
                    for (int j = j0; j < j1; ++j) {
                        const bool up = has_v1 && G.h_val1[j] != 0.f;
                        if (up != (pass == 0)) continue;
                        const size_t e = base + (size_t)i * w + k++;
                        ell_slot[e] = (uint8_t)local[G.h_col[j]];
                        ell_v[e] = float2{G.h_val0[j], has_v1 ? G.h_val1[j] : 0.f};
                        n_up += up;
                    }
                tw[i >> 3] = std::max<uint8_t>(tw[i >> 3], (uint8_t)((j1 - j0 + 1) & ~1));
                tw4[i >> 2] = std::max<uint8_t>(tw4[i >> 2], (uint8_t)((j1 - j0 + 1) & ~1));
                twu[i >> 3] = std::max<uint8_t>(twu[i >> 3], (uint8_t)((n_up + 1) & ~1));
                twu4[i >> 2] = std::max<uint8_t>(twu4[i >> 2], (uint8_t)((n_up + 1) & ~1));
                ss = has_id ? (uint8_t)local[r] : 0;
            }
            self_slot.push_back(ss);
        }
        tile_w.insert(tile_w.end(), tw, tw + BK_WAVES);
        tile_w4.insert(tile_w4.end(), tw4, tw4 + 2 * BK_WAVES);
        {
            float gather = 0.f;
            for (int i = 0; i < BK_WAVES; ++i) gather += tw[i];
            block_gather.push_back(gather);
        }
        tile_wu.insert(tile_wu.end(), twu, twu + BK_WAVES);
        tile_wu4.insert(tile_wu4.end(), twu4, twu4 + 2 * BK_WAVES);
        total_src += (int64_t)cur.size();
        r0 += rows;
        ++bid;
    }
    BlockPlan& P = c->plan;
    P.dev.n_blocks = bid;
    P.dev.ell_w_max = wmax;
    int st;
    if ((st = upload(c, blk_row0, &P.dev.blk_row0)) != SCN_OK) return st;
    if ((st = upload(c, blk_rows, &P.dev.blk_rows)) != SCN_OK) return st;
    if ((st = upload(c, src_ptr, &P.dev.src_ptr)) != SCN_OK) return st;
    if ((st = upload(c, src_rows, &P.dev.src_rows)) != SCN_OK) return st;
    if ((st = upload(c, ell_ptr, &P.dev.ell_ptr)) != SCN_OK) return st;
    if ((st = upload(c, width, &P.dev.width)) != SCN_OK) return st;
    if ((st = upload(c, tile_w, &P.dev.tile_w)) != SCN_OK) return st;
    if ((st = upload(c, tile_w4, &P.dev.tile_w4)) != SCN_OK) return st;
    if ((st = upload(c, tile_wu, &P.dev.tile_wu)) != SCN_OK) return st;
    if ((st = upload(c, tile_wu4, &P.dev.tile_wu4)) != SCN_OK) return st;
    if ((st = upload(c, ell_slot, &P.dev.ell_slot)) != SCN_OK) return st;
    std::vector<uint16_t> ell_enc(ell_slot.size());
    for (size_t i = 0; i < ell_slot.size(); ++i) ell_enc[i] = (uint16_t)((ell_slot[i] << 9) | ((ell_slot[i] & 3) << 5));   // slot*512 | slot part of swz32
    if ((st = upload(c, ell_enc, &P.dev.ell_enc)) != SCN_OK) return st;
    if ((st = upload(c, ell_v, &P.dev.ell_v)) != SCN_OK) return st;
    if ((st = upload(c, self_slot, &P.dev.self_slot)) != SCN_OK) return st;
    P.mean_src_per_row = (double)total_src / std::max(1, n_rows);
    {   // one slab of a block costs a constant part (MFMA, epilogue, barriers: ~60 % on average, tools/stamps.py) + its gather
        double mean = 0.0;
        for (float g : block_gather) mean += g;
        mean /= std::max<size_t>(1, block_gather.size());
        P.h_cost.resize(block_gather.size());
        for (size_t i = 0; i < block_gather.size(); ++i) P.h_cost[i] = (float)(1.5 * mean + block_gather[i]);
    }
    P.h_row0 = blk_row0;
    P.h_row0.push_back(n_rows);
    P.built = true;
    return build_assignments(c);
}

// ------------------------------------------------------------------------------------------------
// LDS layout and pipeline helpers
// ------------------------------------------------------------------------------------------------
// [buf0: BK_SRC*PIECE][buf1: BK_SRC*PIECE][ell_v: BK_ELL_CAP float2][srcrows: BK_SRC i32][slot: BK_ELL_CAP u8][self: BK_R u8][extra]
struct Smem {
    char* buf0;
    int buf_stride;
    __device__ __forceinline__ char* buf(int i) const { return buf0 + i * buf_stride; }
    float2* v;
    int32_t* srcrows;
    uint8_t* slot;
    uint8_t* self;
};
__host__ __device__ static inline size_t smem_bytes(int piece, int extra = 0) {
    size_t b = (size_t)2 * BK_SRC * piece + (size_t)BK_ELL_CAP * 9 + BK_SRC * 4 + BK_R;
    return ((b + 15) / 16) * 16 + extra;
}
__device__ __forceinline__ Smem carve(char* base, int piece) {
    Smem s;
    s.buf0 = base;
    s.buf_stride = BK_SRC * piece;
    s.v = (float2*)(base + 2 * BK_SRC * piece);
    s.srcrows = (int32_t*)(s.v + BK_ELL_CAP);
    s.slot = (uint8_t*)(s.srcrows + BK_SRC);
    s.self = s.slot + BK_ELL_CAP;
    return s;
}

struct BlockMeta { int row0, rows, nsrc, w; };

__device__ __forceinline__ BlockMeta load_block(const PlanDev& P, int b, const Smem& sm) {
    BlockMeta m;
    m.row0 = P.blk_row0[b];
    m.rows = P.blk_rows[b];
    const int sp0 = P.src_ptr[b];
    m.nsrc = P.src_ptr[b + 1] - sp0;
    m.w = P.width[b];
    const int ep = P.ell_ptr[b];
    for (int i = threadIdx.x; i < m.nsrc; i += BK_THREADS) sm.srcrows[i] = P.src_rows[sp0 + i];
    for (int i = threadIdx.x; i < m.w * m.rows; i += BK_THREADS) {
        sm.slot[i] = P.ell_slot[ep + i];
        sm.v[i] = P.ell_v[ep + i];
    }
    if (threadIdx.x < BK_R) sm.self[threadIdx.x] = P.self_slot[(size_t)b * BK_R + threadIdx.x];
    return m;
}

// two consecutive ELL entries of a row: slots and (val0, val1) pairs, one 2-byte and one 16-byte LDS read
struct EllPair { int s0, s1; f32x4 v; };
__device__ __forceinline__ EllPair ell_load(const Smem& sm, int idx) {
    EllPair e;
    const uint32_t ss = *(const uint16_t*)(sm.slot + idx);
    e.s0 = ss & 255;
    e.s1 = ss >> 8;
    e.v = *(const f32x4*)(sm.v + idx);
    return e;
}

// XOR swizzle of the 16-byte chunk index inside a staged piece (involution for a fixed slot): keeps the
// lane = (point, channel slice) gather spread over the LDS banks.
//   512-B pieces (C=32): chunk = n*8 + h*4 + q ; 256-B pieces (C=16): chunk = n*4 + g.
__device__ __forceinline__ int swz32(int slot, int chunk) {
    return chunk ^ ((((slot >> 1) & 1) << 2) | ((slot & 1) << 1) | ((chunk >> 4) & 1));
}
__device__ __forceinline__ int swz16(int slot, int chunk) { return chunk ^ (slot & 3); }

// ---- C = 32 (512-byte pieces): LDS carve with 16-bit pre-encoded ELL slots, and the gather built on it ----
// [buf0][buf1][ell_v: BK_ELL_CAP float2][srcrows: BK_SRC i32][enc: BK_ELL_CAP u16][self: BK_R u8][extra]
// streamed tensors of the backward (aux read once, dx written once per launch): non-temporal, so that the L2 keeps the staged
// halo rows neighbouring blocks share: -1.4 % (SCN_NO_NT: plain accesses, for A/B runs).  The forward's output stores are
// left plain: non-temporal they cost it 1-3 %.
#ifdef SCN_NO_NT
#define SCN_ST_STREAM(ptr, val) (*(ptr) = (val))
#define SCN_LD_STREAM(ptr) (*(ptr))
#else
#define SCN_ST_STREAM(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define SCN_LD_STREAM(ptr) __builtin_nontemporal_load(ptr)
#endif

struct SmemC32 {
    char* buf0;
    __device__ __forceinline__ char* buf(int i) const { return buf0 + i * (BK_SRC * 512); }
    float2* v;
    int32_t* srcrows;
    uint16_t* enc;
    uint8_t* self;
};
__host__ __device__ static inline size_t smem_bytes_c32(int extra = 0) {
    size_t b = (size_t)2 * BK_SRC * 512 + (size_t)BK_ELL_CAP * 10 + BK_SRC * 4 + BK_R;
    return ((b + 15) / 16) * 16 + extra;
}
__device__ __forceinline__ SmemC32 carve_c32(char* base) {
    SmemC32 s;
    s.buf0 = base;
    s.v = (float2*)(base + 2 * BK_SRC * 512);
    s.srcrows = (int32_t*)(s.v + BK_ELL_CAP);
    s.enc = (uint16_t*)(s.srcrows + BK_SRC);
    s.self = (uint8_t*)(s.enc + BK_ELL_CAP);
    return s;
}
static_assert(BK_SRC * 512 == 65536 && BK_SRC <= 128, "gather_c32 XORs slot*512 (< 2^16) with the buffer bit 2^16");

template <int NT>
__device__ __forceinline__ BlockMeta load_block_c32(const PlanDev& P, int b, const SmemC32& sm) {
    BlockMeta m;
    m.row0 = P.blk_row0[b];
    m.rows = P.blk_rows[b];
    const int sp0 = P.src_ptr[b];
    m.nsrc = P.src_ptr[b + 1] - sp0;
    m.w = P.width[b];
    const int ep = P.ell_ptr[b];
    for (int i = threadIdx.x; i < m.nsrc; i += NT) sm.srcrows[i] = P.src_rows[sp0 + i];
    for (int i = threadIdx.x; i < m.w * m.rows; i += NT) {
        sm.enc[i] = P.ell_enc[ep + i];
        sm.v[i] = P.ell_v[ep + i];
    }
    if (threadIdx.x < BK_R) sm.self[threadIdx.x] = P.self_slot[(size_t)b * BK_R + threadIdx.x];
    return m;
}

// Gather of one lane's NQ 16-byte chunks of row `row` from the staged 512-byte pieces: identity term gs, and the ELL row
// (entry pairs) accumulated into gl (val0 operator) and gu (val1 operator).
//  * cb[q] = (this lane's swizzled chunk index << 4) | (buffer index << 16); the LDS address of a chunk is cb[q] ^ enc with
//    enc = slot*512 | (slot&3)*32 straight from the plan: one v_xor per read instead of five address instructions -- these
//    kernels are VALU-issue bound (profiles/r01_pmc_fwd_bwd_c32_bf16.txt), so instruction count is what matters;
//  * all 2*NQ LDS reads of an entry pair are issued before the first FMA and the next pair's slots / values are fetched
//    behind them: left to itself hipcc serialises read -> s_waitcnt lgkmcnt(0) -> use under register pressure.
template <int NQ>
__device__ __forceinline__ void read_self_c32(const SmemC32& sm, int row, const uint32_t (&cb)[NQ], f32x4 (&gs)[NQ]) {
    const uint32_t slot = sm.self[row];
    const uint32_t enc = (slot << 9) | ((slot & 3) << 5);
#pragma unroll
    for (int q = 0; q < NQ; ++q) gs[q] = *(const f32x4*)(sm.buf0 + (cb[q] ^ enc));
}

template <int NQ, bool DUAL = true, bool SELF = true>     // DUAL = false: operator with one value array, gu is left untouched
__device__ __forceinline__ void gather_c32(const SmemC32& sm, int row, int w, int tw, int twu, const uint32_t (&cb)[NQ],   // SELF = false: gs is left untouched (read_self_c32 later)
                                           f32x4 (&gs)[NQ], f32x4 (&gl)[NQ], f32x4 (&gu)[NQ]) {
    const char* lds = sm.buf0;
    const int rb = row * w;                                   // w is even: entry pairs are 4-byte aligned
    uint32_t ss = *(const uint32_t*)(sm.enc + rb);
    f32x4 v = *(const f32x4*)(sm.v + rb);
    {
        const uint32_t slot = sm.self[row];
        const uint32_t enc = (slot << 9) | ((slot & 3) << 5);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (SELF) gs[q] = *(const f32x4*)(lds + (cb[q] ^ enc));
            gl[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (DUAL) gu[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // the plan puts the entries of the second operator first (its pattern is a subset): entries t >= twu only feed gl
    auto pair = [&](int t, auto both) {
        const uint32_t e0 = ss & 0xffffu, e1 = ss >> 16;
        const f32x4 vc = v;
        f32x4 d0[NQ], d1[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) d0[q] = *(const f32x4*)(lds + (cb[q] ^ e0));
#pragma unroll
        for (int q = 0; q < NQ; ++q) d1[q] = *(const f32x4*)(lds + (cb[q] ^ e1));
        if (t + 2 < tw) {
            ss = *(const uint32_t*)(sm.enc + rb + t + 2);
            v = *(const f32x4*)(sm.v + rb + t + 2);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            gl[q] += vc[0] * d0[q];
            if (decltype(both)::value) gu[q] += vc[1] * d0[q];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            gl[q] += vc[2] * d1[q];
            if (decltype(both)::value) gu[q] += vc[3] * d1[q];
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    int t = 0;
    if (DUAL)
        for (; t < twu; t += 2) pair(t, std::true_type{});
    for (; t < tw; t += 2) pair(t, std::false_type{});
}

// LDS-DMA of nsrc pieces of slab-base Xs into buf: LDS image is lane-linear, the swizzle goes on the SOURCE chunk.
template <int PIECE, int SWZ>
__device__ __forceinline__ void dma_stage(const char* Xs, char* buf, const Smem& sm, int nsrc) {
    constexpr int CPP = PIECE / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = nsrc * CPP;
    for (int base = wave * 64; base < total; base += BK_THREADS) {
        const int c = base + lane;
        if (c < total) {
            const int slot = c / CPP, pos = c % CPP;
            int d = pos;
            if (SWZ == 32) d = swz32(slot, pos);
            if (SWZ == 16) d = swz16(slot, pos);
            const char* g = Xs + (size_t)sm.srcrows[slot] * PIECE + d * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(buf + base * 16), 16, 0, 0);
        }
    }
}
// runtime piece size (dual SpMM), no swizzle
__device__ __forceinline__ void dma_stage_rt(const char* Xs, char* buf, const Smem& sm, int nsrc, int piece, int cpp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = nsrc * cpp;
    for (int base = wave * 64; base < total; base += BK_THREADS) {
        const int c = base + lane;
        if (c < total) {
            const int slot = c / cpp, pos = c - slot * cpp;
            const char* g = Xs + (size_t)sm.srcrows[slot] * piece + pos * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(buf + base * 16), 16, 0, 0);
        }
    }
}

__device__ __forceinline__ void wait_all_and_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wait_vm_and_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// XCD-contiguous block range of this workgroup: blocks b = first, first+stride, ... < last
__device__ __forceinline__ void block_range(int n_blocks, int& first, int& last, int& stride) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    stride = gridDim.x >> 3;
    const int b0 = (int)((int64_t)n_blocks * xcd / 8);
    last = (int)((int64_t)n_blocks * (xcd + 1) / 8);
    first = b0 + j;
}

#define SCN_SLAB_RANGE() \
    const int slab0 = (int)((int64_t)blockIdx.y * n_slabs / gridDim.y), slab1 = (int)((int64_t)(blockIdx.y + 1) * n_slabs / gridDim.y)

// Work units of a kernel that accepts a WorkList `wl`: dense = (every block) x (this workgroup's slab range); listed = the
// listed blocks, each with its own slab list (gridDim.y == 1).  SCN_UNIT_BEGIN opens the unit loop and defines b (block),
// k0 / n_it (first entry / trip count) and SLAB_AT(it); all of it is wave-uniform (scalar loads).
#define SCN_UNIT_RANGE()                                                             \
    const bool listed = wl.block != nullptr;                                         \
    int u_, u_end_, u_stride_;                                                       \
    block_range(listed ? wl.n_work : P.n_blocks, u_, u_end_, u_stride_);             \
    SCN_SLAB_RANGE()
#define SCN_UNIT_BEGIN()                                                             \
    for (; u_ < u_end_; u_ += u_stride_) {                                           \
        const int b = listed ? wl.block[u_] : (P.assign ? P.assign[u_] : u_);        \
        const int k0 = listed ? wl.ptr[u_] : slab0;                                  \
        const int n_it = listed ? wl.ptr[u_ + 1] - k0 : slab1 - slab0;               \
        if (n_it <= 0) continue;
#define SLAB_AT(it) (listed ? wl.slab[k0 + (it)] : k0 + (it))

// ------------------------------------------------------------------------------------------------
// dual SpMM on K-float pieces (K % 4 == 0, K <= 128): ya = val0-operator * x, yb = val1-operator * x
// 16 waves per workgroup (no MFMA, few registers): thread = (row, 16-byte chunk), two items per thread;
// results are stored one slab late so the per-slab vmcnt(0) never waits on fresh stores.
// ------------------------------------------------------------------------------------------------
constexpr int Y_STRIDE = 4;      // floats per point of the shifted first-layer input y = (x, S_lo x, S_up x, 0): 16-byte records
constexpr int SP_THREADS = 1024;
constexpr int SP_ITEMS = 2;           // BK_R * 32 chunks / SP_THREADS
constexpr int SP_SLAB_GROUP = 8;      // slabs a workgroup processes per visit of a block

// A staged piece of the two-buffer SpMM is VIRTUAL: `batch` consecutive slabs of one source row side by side (K floats
// each, cpk = K/4 chunks), so that a narrow operand -- the 1-channel tensors of the first / last Bunch layer are K = 4 floats
// per row and slab: 64 work items per block -- fills the workgroup: chunk pos of a piece = chunk pos % cpk of slab pos / cpk.
// batch = 1 is the plain layout.
__device__ __forceinline__ void dma_stage_sp(const char* X, int slab0, int n_avail, size_t slab_bytes, char* buf, const Smem& sm,
                                             int nsrc, int k4, int cpk, int vcpp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = nsrc * vcpp;
    for (int base = wave * 64; base < total; base += SP_THREADS) {
        const int c = base + lane;
        if (c < total) {
            const int slot = c / vcpp, pos = c - slot * vcpp;
            const int sl = pos / cpk, sub = pos - sl * cpk;
            if (sl < n_avail) {
                const char* g = X + (size_t)(slab0 + sl) * slab_bytes + (size_t)sm.srcrows[slot] * k4 + sub * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(buf + base * 16), 16, 0, 0);
            }
        }
    }
}

// (plain pieces of one slab at Xs)
__device__ __forceinline__ void dma_stage_sp(const char* Xs, char* buf, const Smem& sm, int nsrc, int piece, int cpp) {
    dma_stage_sp(Xs, 0, 1, 0, buf, sm, nsrc, piece, cpp, cpp);
}

template <bool DUAL>
__global__ __launch_bounds__(SP_THREADS, 4) void spmm_blocked_kernel(PlanDev P, const float* __restrict__ X,
                                                                     float* __restrict__ ya, float* __restrict__ yb,
                                                                     int n_rows, int n_cols, int n_slabs, int K, int batch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int k4 = K * 4, cpk = K / 4;                              // one slab's piece
    const int piece = k4 * batch, cpp = cpk * batch;                // the staged (virtual) piece
    const size_t in_slab_bytes = (size_t)n_cols * k4;
    const Smem sm = carve(smem, piece);
    uint8_t* tws = (uint8_t*)(smem + smem_bytes(piece));           // [BK_WAVES] per-row-group widths
    int b0, b_end, b_stride;
    block_range(P.n_blocks, b0, b_end, b_stride);
    const int n_units = (n_slabs + batch - 1) / batch;              // a unit = `batch` slabs (the last one may hold fewer)
    const int unit_lo = (int)((int64_t)blockIdx.y * n_units / gridDim.y), unit_hi = (int)((int64_t)(blockIdx.y + 1) * n_units / gridDim.y);
    if (unit_lo >= unit_hi) return;
    f32x4 pa[SP_ITEMS], pb[SP_ITEMS];
    int pend_slab0 = 0, pend_row0 = 0, pend_avail = 0;
    int pend_total = -1;
    auto store_pending = [&]() {
#pragma unroll
        for (int k = 0; k < SP_ITEMS; ++k) {
            const int idx = threadIdx.x + k * SP_THREADS;
            if (idx < pend_total) {
                const int r = idx / cpp, ch = idx - r * cpp;
                const int sl = ch / cpk, sub = ch - sl * cpk;
                if (sl < pend_avail) {
                    const size_t off = ((size_t)(pend_slab0 + sl) * n_rows + pend_row0 + r) * K + sub * 4;
                    // non-temporal: the outputs would otherwise evict the halo rows neighbouring blocks are about to read
                    __builtin_nontemporal_store(pa[k], (f32x4*)(ya + off));
                    if (DUAL) __builtin_nontemporal_store(pb[k], (f32x4*)(yb + off));
                }
            }
        }
    };
    // unit groups outside, blocks inside: all workgroups sweep their blocks for the same 8 slabs before moving on, which
    // keeps more of the halo rows that neighbouring blocks share within reach of the caches (-4 % at |E| = 1M; 4 or 16
    // slabs per group are worse, and so is giving a workgroup a contiguous block range)
    for (int u0 = unit_lo; u0 < unit_hi; u0 += SP_SLAB_GROUP)
    for (int bi = b0, u1 = min(u0 + SP_SLAB_GROUP, unit_hi); bi < b_end; bi += b_stride) {
        const int b = P.assign ? P.assign[bi] : bi;
        wait_all_and_barrier();
        BlockMeta m;
        {   // load_block with this kernel's thread count
            m.row0 = P.blk_row0[b];
            m.rows = P.blk_rows[b];
            const int sp0 = P.src_ptr[b];
            m.nsrc = P.src_ptr[b + 1] - sp0;
            m.w = P.width[b];
            const int ep = P.ell_ptr[b];
            for (int i = threadIdx.x; i < m.nsrc; i += SP_THREADS) sm.srcrows[i] = P.src_rows[sp0 + i];
            for (int i = threadIdx.x; i < m.w * m.rows; i += SP_THREADS) {
                sm.slot[i] = P.ell_slot[ep + i];
                sm.v[i] = P.ell_v[ep + i];
            }
            if (threadIdx.x < BK_WAVES) tws[threadIdx.x] = P.tile_w[b * BK_WAVES + threadIdx.x];
        }
        __syncthreads();
        const int total = m.rows * cpp;
        const bool wave_uniform_rows = cpp >= 8 && (64 % cpp) == 0;
        auto avail = [&](int u) { return min(batch, n_slabs - u * batch); };
        dma_stage_sp((const char*)X, u0 * batch, avail(u0), in_slab_bytes, sm.buf(0), sm, m.nsrc, k4, cpk, cpp);
        for (int u = u0; u < u1; ++u) {
            const int cur_off = ((u - u0) & 1) * sm.buf_stride;
            wait_vm_and_barrier();
            if (u + 1 < u1)
                dma_stage_sp((const char*)X, (u + 1) * batch, avail(u + 1), in_slab_bytes, sm.buf((u + 1 - u0) & 1), sm, m.nsrc, k4,
                             cpk, cpp);
            if (pend_total >= 0) store_pending();
#pragma unroll
            for (int k = 0; k < SP_ITEMS; ++k) {
                const int idx = threadIdx.x + k * SP_THREADS;
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
                if (idx < total) {
                    const int r = idx / cpp, ch = idx - r * cpp;
                    int tw = tws[r >> 3];
                    if (wave_uniform_rows) tw = __builtin_amdgcn_readfirstlane(tw);   // the wave's lanes share one row group
                    const int rb = r * m.w;
                    const char* cbase = sm.buf0 + cur_off + ch * 16;
#pragma unroll 2
                    for (int t = 0; t < tw; t += 2) {
                        const EllPair e = ell_load(sm, rb + t);
                        const f32x4 d0 = *(const f32x4*)(cbase + e.s0 * piece);
                        const f32x4 d1 = *(const f32x4*)(cbase + e.s1 * piece);
                        a0 += e.v[0] * d0;
                        if (DUAL) a1 += e.v[1] * d0;
                        a0 += e.v[2] * d1;
                        if (DUAL) a1 += e.v[3] * d1;
                    }
                }
                pa[k] = a0;
                pb[k] = a1;
            }
            pend_slab0 = u * batch;
            pend_avail = avail(u);
            pend_row0 = m.row0;
            pend_total = total;
        }
    }
    if (pend_total >= 0) store_pending();
}

// ------------------------------------------------------------------------------------------------
// dual SpMM, K = 128 or 64, as a RING of four half-piece stages: the staged unit is half a row piece (K/2 floats of every
// source row), so the same 128 KB of LDS holds four stages instead of two buffers and three stages' LDS-DMAs are in flight
// while one is consumed (the two-buffer kernel has one slab in flight at a time: DMA-only it runs 9.1 ms against 7.8 for
// the traffic).  NI = LDS-DMA instructions per wave and stage (BK_SRC * K/8 chunks / 1024 lanes), issued
// by EVERY wave with clamped source addresses so that `s_waitcnt vmcnt(NI * stages still in flight)` means the same thing in
// every wave; loads complete in order among loads, so counting only them is safe with the result stores in between.
// ------------------------------------------------------------------------------------------------
constexpr int RING_SLAB_GROUP = 16;   // slabs per block visit: the ring's three-stage fill makes a visit dearer than in the two-buffer kernel (8: +1 %, 32: +1 %)

template <int NI>
__device__ __forceinline__ void wait_ring(int later) {           // `later` = stages issued after the one needed (wave-uniform)
    if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <bool DUAL, int NI>
__global__ __launch_bounds__(SP_THREADS, 4) void spmm_ring_kernel(PlanDev P, const float* __restrict__ X,
                                                                  float* __restrict__ ya, float* __restrict__ yb,
                                                                  int n_rows, int n_cols, int n_slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int K = 64 * NI, PIECE = K * 4, HALF = PIECE / 2, CPH = HALF / 16;     // chunks per half piece: 16 or 8
    constexpr int STAGE = BK_SRC * HALF;                                             // 32 KB / 16 KB; 4 stages = both buffers
    static_assert(BK_SRC * CPH == NI * SP_THREADS, "every lane issues exactly NI LDS-DMA chunks per stage");
    const Smem sm = carve(smem, PIECE);
    uint8_t* tws = (uint8_t*)(smem + smem_bytes(PIECE));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int b0, b_end, b_stride;
    block_range(P.n_blocks, b0, b_end, b_stride);
    const int slab_lo = (int)((int64_t)blockIdx.y * n_slabs / gridDim.y), slab_hi = (int)((int64_t)(blockIdx.y + 1) * n_slabs / gridDim.y);
    if (slab_lo >= slab_hi) return;
    f32x4 pa = {0.f, 0.f, 0.f, 0.f}, pb = pa;
    float *pend_a = nullptr, *pend_b = nullptr;
    for (int slab0 = slab_lo; slab0 < slab_hi; slab0 += RING_SLAB_GROUP)
    for (int bi = b0, slab1 = min(slab0 + RING_SLAB_GROUP, slab_hi); bi < b_end; bi += b_stride) {
        const int b = P.assign ? P.assign[bi] : bi;
        wait_all_and_barrier();
        BlockMeta m;
        {
            m.row0 = P.blk_row0[b];
            m.rows = P.blk_rows[b];
            const int sp0 = P.src_ptr[b];
            m.nsrc = P.src_ptr[b + 1] - sp0;
            m.w = P.width[b];
            const int ep = P.ell_ptr[b];
            for (int i = threadIdx.x; i < m.nsrc; i += SP_THREADS) sm.srcrows[i] = P.src_rows[sp0 + i];
            for (int i = threadIdx.x; i < m.w * m.rows; i += SP_THREADS) {
                sm.slot[i] = P.ell_slot[ep + i];
                sm.v[i] = P.ell_v[ep + i];
            }
            if (threadIdx.x < BK_WAVES) tws[threadIdx.x] = P.tile_w[b * BK_WAVES + threadIdx.x];
        }
        __syncthreads();
        // this lane's NI chunks of a stage: chunk c = slot * CPH + pos, clamped into the block's sources
        // (a block of rows without entries -- isolated nodes cluster in the holes of the complex -- has no sources at all:
        // nothing is staged for it, its rows are written as zeros; has_src is uniform over the workgroup)
        uint32_t goff[NI];
        const bool has_src = m.nsrc > 0;
        const int total = m.nsrc * CPH;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = has_src ? min((i * (SP_THREADS / 64) + wave) * 64 + lane, total - 1) : 0;
            goff[i] = has_src ? (uint32_t)sm.srcrows[c / CPH] * PIECE + (c % CPH) * 16 : 0u;
        }
        const int n_stage = 2 * (slab1 - slab0);
        auto issue = [&](int q) {                                   // stage q = (slab slab0 + q/2, half q&1)
            const char* src = (const char*)X + (size_t)(slab0 + (q >> 1)) * n_cols * PIECE + (q & 1) * HALF;
            char* dst = sm.buf0 + (q & 3) * STAGE;
#pragma unroll
            for (int i = 0; i < NI; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + goff[i]),
                                                 (__attribute__((address_space(3))) void*)(dst + ((i * (SP_THREADS / 64) + wave) * 64) * 16),
                                                 16, 0, 0);
        };
        // SCN_SPMM_FLOOR (diagnostic builds only, tools/spmm_ceiling.sh; results are WRONG by design): 1 = LDS-DMA + stores, no
        // gather (what the memory system alone takes); 2 = gather + stores on whatever LDS holds, no LDS-DMA
#if defined(SCN_SPMM_FLOOR) && SCN_SPMM_FLOOR == 2
        const bool do_dma = false;
#else
        const bool do_dma = has_src;
#endif
        if (do_dma)
            for (int q = 0; q < 3 && q < n_stage; ++q) issue(q);
        // this thread's item of every stage: (row r, 16-byte chunk ch of the half piece)
        const int idx = threadIdx.x, r = idx / CPH, ch = idx - r * CPH;
        const bool live = r < m.rows;
#if defined(SCN_SPMM_FLOOR) && SCN_SPMM_FLOOR == 1
        const int tw = 0;
#else
        const int tw = live ? __builtin_amdgcn_readfirstlane(tws[min(r, BK_R - 1) >> 3]) : 0;   // CPH >= 8: a wave's lanes share a row group
#endif
        const int rb = r * m.w;
        for (int q = 0; q < n_stage; ++q) {
            wait_ring<NI>(min(n_stage - 1 - q, 2));
            if (do_dma && q + 3 < n_stage) issue(q + 3);
            if (pend_a) {                                           // non-temporal, one stage late (see spmm_blocked_kernel)
                __builtin_nontemporal_store(pa, (f32x4*)pend_a);
                if (DUAL) __builtin_nontemporal_store(pb, (f32x4*)pend_b);
            }
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
            if (live) {
                const char* cbase = sm.buf0 + (q & 3) * STAGE + ch * 16;
#pragma unroll 2
                for (int t = 0; t < tw; t += 2) {
                    const EllPair e = ell_load(sm, rb + t);
                    const f32x4 d0 = *(const f32x4*)(cbase + e.s0 * HALF);
                    const f32x4 d1 = *(const f32x4*)(cbase + e.s1 * HALF);
                    a0 += e.v[0] * d0;
                    if (DUAL) a1 += e.v[1] * d0;
                    a0 += e.v[2] * d1;
                    if (DUAL) a1 += e.v[3] * d1;
                }
            }
            pa = a0;
            pb = a1;
            const size_t off = ((size_t)(slab0 + (q >> 1)) * n_rows + m.row0 + r) * K + (q & 1) * (K / 2) + ch * 4;
            pend_a = live ? ya + off : nullptr;
            pend_b = live && DUAL ? yb + off : nullptr;
        }
    }
    if (pend_a) {
        __builtin_nontemporal_store(pa, (f32x4*)pend_a);
        if (DUAL) __builtin_nontemporal_store(pb, (f32x4*)pend_b);
    }
}

// ------------------------------------------------------------------------------------------------
// forward, C_in = C_out = 32   (v_mfma_f32_32x32x2_f32; lane = (point p = lane&31, channel half h = lane>>5))
// (f32 MFMA and VALU do not overlap on a gfx950 SIMD -- tools/ubench/mfma_valu_overlap.hip -- so the waves are not
// staggered; what does overlap with the MFMA chain is vector-memory issue, see c32_mfma_epilogue.)
// ------------------------------------------------------------------------------------------------
// out^T tile = W^T Z^T: the MFMA takes the weights as A and the gathered point vectors as B, so the accumulator
// has the POINT on the lane and 16 channels in registers -> the epilogue stores four 16-byte chunks per lane.
//   D[i = channel][j = point]: lane (p, h) holds channels (r&3) + 8*(r>>2) + 4*h, r = 0..15, of point p.
template <int ACT, typename Side>
__device__ __forceinline__ void c32_mfma_epilogue(const f32x4 (&zs)[4], const f32x4 (&zl)[4], const f32x4 (&zu)[4],
                                                  const float (&Bw)[3][16], f32x16& pend, Side&& side) {
    // `side(k)` (k = 0..11) issues one independent vector-memory instruction after every 4th MFMA: the wave would sit
    // on the accumulator dependency anyway, so the memory pipe's address processing hides under the MFMA chain.
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Bw[0][s], zs[s >> 2][s & 3], acc, 0, 0, 0);
        if ((s & 3) == 3) side(s >> 2);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Bw[1][s], zl[s >> 2][s & 3], acc, 0, 0, 0);
        if ((s & 3) == 3) side(4 + (s >> 2));
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Bw[2][s], zu[s >> 2][s & 3], acc, 0, 0, 0);
        if ((s & 3) == 3) side(8 + (s >> 2));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) pend[r] = act_apply_fast(ACT, acc[r]);
}

// ptr = &out[point of this lane][4*h]; chunk g holds channels 8g + 4h .. +3
__device__ __forceinline__ void c32_store_tile(const f32x16& pend, float* ptr, bool valid) {
    if (valid) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(f32x4*)(ptr + 8 * g) = f32x4{pend[4 * g], pend[4 * g + 1], pend[4 * g + 2], pend[4 * g + 3]};
    }
}

template <int ACT>
__global__ __launch_bounds__(BK_THREADS, 2) void fwd_c32_kernel(PlanDev P, const float* __restrict__ X,
                                                                const float* __restrict__ W0,
                                                                const float* __restrict__ W1,
                                                                const float* __restrict__ W2,
                                                                float* __restrict__ out, int n_rows, int n_cols,
                                                                int n_slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PIECE = 512, CPP = 32, NDMA = BK_SRC * CPP / BK_THREADS;   // 8 LDS-DMA instructions per wave
    const Smem sm = carve(smem, PIECE);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 31, h = lane >> 5, n = p & 3, rt = wave * 8 + (p >> 2);
    STAMP_DECL;
    // weights: k-step s of segment g contracts channel 16*h + s ; W_g[(16h+s)][co = p]
    float Bw[3][16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        Bw[0][s] = W0[(16 * h + s) * 32 + p];
        Bw[1][s] = W1[(16 * h + s) * 32 + p];
        Bw[2][s] = W2[(16 * h + s) * 32 + p];
    }
    int b, b_end, b_stride;
    block_range(P.n_blocks, b, b_end, b_stride);
    SCN_SLAB_RANGE();
    if (slab0 >= slab1) return;
    f32x16 pend;                   // finished tile waiting to be stored (one slab late)
    float* pend_ptr = nullptr;
    bool pend_valid = false;
    f32x4 zs[4], zl[4], zu[4];     // gathered tile
    float* z_ptr = nullptr;
    bool z_valid = false;
    int cq[4];                     // lane-constant part of the swizzled chunk index
#pragma unroll
    for (int q = 0; q < 4; ++q) cq[q] = (n * 8 + h * 4 + q) ^ (n >> 1);
    const size_t slab_bytes = (size_t)n_cols * PIECE;
    for (; b < b_end; b += b_stride) {
        wait_all_and_barrier();
        const BlockMeta m = load_block(P, b, sm);
        __syncthreads();
        const int tw = P.tile_w[b * BK_WAVES + wave];
        const int rtc = rt < m.rows ? rt : m.rows - 1;
        // per-lane source offsets of this wave's LDS-DMA instructions (the same for every slab of the block)
        uint32_t goff[NDMA];
        const int total = m.nsrc * CPP;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int c = (i * BK_WAVES + wave) * 64 + lane;
            const int slot = c / CPP, pos = c % CPP;
            goff[i] = c < total ? (uint32_t)sm.srcrows[slot] * PIECE + swz32(slot, pos) * 16 : 0u;
        }
        auto stage = [&](int slab, char* buf) {
            const char* Xs = (const char*)X + (size_t)slab * slab_bytes;
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
                const int base = (i * BK_WAVES + wave) * 64;
                if (base + lane < total)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Xs + goff[i]),
                                                     (__attribute__((address_space(3))) void*)(buf + base * 16), 16, 0, 0);
            }
        };
        stage(slab0, sm.buf(0));
        for (int slab = slab0; slab < slab1; ++slab) {
            const char* cur = sm.buf((slab - slab0) & 1);
            STAMP_START();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP_ADD(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            STAMP_ADD(1);
            STAMP_ADD(2);
            STAMP_ADD(3);
            auto do_mfma = [&](int) {
                // the previous tile's four 16-byte stores and the eight LDS-DMA loads of the next slab ride inside the chain
                const f32x16 prev = pend;
                float* const prev_ptr = pend_ptr;
                const bool prev_valid = pend_valid && pend_ptr != nullptr;
                const bool more = slab + 1 < slab1;
                const char* Xn = (const char*)X + (size_t)(slab + 1) * slab_bytes;
                char* nbuf = sm.buf((slab + 1 - slab0) & 1);
                c32_mfma_epilogue<ACT>(zs, zl, zu, Bw, pend, [&](int k) {
                    if (k < NDMA) {
                        const int base = (k * BK_WAVES + wave) * 64;
                        if (more && base + lane < total)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Xn + goff[k]),
                                                             (__attribute__((address_space(3))) void*)(nbuf + base * 16), 16, 0, 0);
                    } else if (prev_valid) {
                        const int g = k - NDMA;
                        *(f32x4*)(prev_ptr + 8 * g) = f32x4{prev[4 * g], prev[4 * g + 1], prev[4 * g + 2], prev[4 * g + 3]};
                    }
                });
                pend_ptr = z_ptr;
                pend_valid = z_valid;
            };
            STAMP_ADD(5);
            {
                const int slot = sm.self[rtc];
                const int sx = (((slot >> 1) & 1) << 2) | ((slot & 1) << 1);
                const char* base = cur + slot * PIECE;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    zs[q] = *(const f32x4*)(base + ((cq[q] ^ sx) << 4));
                    zl[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    zu[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                // one-entry look-ahead: the reads of entry k+1 are in flight while entry k is accumulated
                const int rb = rtc * m.w;
                f32x4 d[4];
                EllPair e = ell_load(sm, rb);
                if (tw > 0) {
                    const int x0 = (((e.s0 >> 1) & 1) << 2) | ((e.s0 & 1) << 1);
                    const char* b0 = cur + e.s0 * PIECE;
#pragma unroll
                    for (int q = 0; q < 4; ++q) d[q] = *(const f32x4*)(b0 + ((cq[q] ^ x0) << 4));
                }
                for (int t = 0; t < tw; t += 2) {
                    f32x4 d1[4];
                    {
                        const int x1 = (((e.s1 >> 1) & 1) << 2) | ((e.s1 & 1) << 1);
                        const char* b1 = cur + e.s1 * PIECE;
#pragma unroll
                        for (int q = 0; q < 4; ++q) d1[q] = *(const f32x4*)(b1 + ((cq[q] ^ x1) << 4));
                    }
                    const f32x4 v = e.v;
                    if (t + 2 < tw) e = ell_load(sm, rb + t + 2);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        zl[q] += v[0] * d[q];
                        zu[q] += v[1] * d[q];
                    }
                    if (t + 2 < tw) {
                        const int x0 = (((e.s0 >> 1) & 1) << 2) | ((e.s0 & 1) << 1);
                        const char* b0 = cur + e.s0 * PIECE;
#pragma unroll
                        for (int q = 0; q < 4; ++q) d[q] = *(const f32x4*)(b0 + ((cq[q] ^ x0) << 4));
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        zl[q] += v[2] * d1[q];
                        zu[q] += v[3] * d1[q];
                    }
                }
            }
            z_ptr = out + (((size_t)slab * n_rows + m.row0 + rt) * BK_NS + n) * 32 + 4 * h;
            z_valid = rt < m.rows;
            STAMP_ADD(4);
            do_mfma(slab);
            STAMP_ADD(5);
        }
    }
    STAMP_FLUSH();
    if (pend_ptr) c32_store_tile(pend, pend_ptr, pend_valid);
}

// ------------------------------------------------------------------------------------------------
// forward, C_in = C_out = 32 on bf16 MFMA with an EXACT three-way split of both operands.
//   x = hi + mid + lo with hi/mid/lo the successive top-8-bit truncations of the fp32 value (24 mantissa bits = 3 x 8,
//   so the split is exact); z*w is evaluated as hh + hm + mh + hl + lh + mm (the dropped terms are < 2^-23 |z w|), each
//   bf16 product is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32 -> same accuracy class as the fp32 MFMA.
// Why: on gfx950 the fp32 MFMA blocks the SIMD's VALU (tools/ubench/mfma_valu_overlap.hip: times add), while the bf16 MFMA
// co-executes with non-packed VALU work of the partner wave (mfma_bf16_valu_overlap.hip) and does the 96x32 contraction
// in 36 x 32 cycles instead of 48 x 64.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Split3 { bf16x8 hi, mid, lo; };

__device__ __forceinline__ uint32_t pack_hi16(float a, float b) {   // (bf16 bits of a, bf16 bits of b) by truncation
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

__device__ __forceinline__ Split3 split3(const float (&x)[8]) {
    u32x4 h, m, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        h[i] = pack_hi16(a, b);
        const float ra = a - trunc_bf16(a), rb = b - trunc_bf16(b);
        m[i] = pack_hi16(ra, rb);
        const float sa = ra - trunc_bf16(ra), sb = rb - trunc_bf16(rb);
        l[i] = pack_hi16(sa, sb);
    }
    Split3 s;
    s.hi = __builtin_bit_cast(bf16x8, h);
    s.mid = __builtin_bit_cast(bf16x8, m);
    s.lo = __builtin_bit_cast(bf16x8, l);
    return s;
}

// six bf16 MFMAs = one exact-split fp32 product block (A = weights, B = gathered points)
__device__ __forceinline__ f32x16 mfma_split(const Split3& w, const Split3& z, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.lo, z.hi, acc, 0, 0, 0);     // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, z.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.mid, z.mid, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.mid, z.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, z.mid, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, z.hi, acc, 0, 0, 0);
    return acc;
}

// ------------------------------------------------------------------------------------------------
// forward, C_in = C_out = 32, SIXTEEN waves per workgroup (4 per SIMD, <= 128 VGPRs):
//   wave = 4 rows x 4 trajectories = 16 points, lane = (point = lane&15, channel octet kq = lane>>4),
//   v_mfma_f32_16x16x32_bf16 with the exact three-way split; the split weights live in LDS (18 KB, lane-linear A fragments).
// The 8-wave kernels are latency-bound (every pipe < 40 % busy, tools/stamps.py); twice the resident waves at half the
// per-wave work is the cheapest way to hide the LDS round trips of the gather.
// ------------------------------------------------------------------------------------------------
constexpr int W16_THREADS = 1024, W16_WAVES = 16;
constexpr int W16_WFRAG_BYTES = 3 * 2 * 3 * 64 * 16;      // [segment][co tile][split][lane] x 8 bf16

// EXT0: the operator has ONE value array and the layer is  act(X0 W0 + X W1 + (S X) W2)  -- segment 0 comes from a second
// tensor X0 (the lane's own rows, straight from HBM), segment 1 is the staged tensor's own row, segment 2 its gathered shift
// (Ebli on large complexes: X0 = H, X = S H, so the third term is S^2 H without ever forming S^2).
template <int ACT, bool EXT0 = false>
__global__ __launch_bounds__(W16_THREADS, 4) void fwd_c32_w16_kernel(PlanDev P, const float* __restrict__ X,
                                                                     const float* __restrict__ X0,
                                                                     const float* __restrict__ W0,
                                                                     const float* __restrict__ W1,
                                                                     const float* __restrict__ W2,
                                                                     float* __restrict__ out, int n_rows, int n_cols,
                                                                     int n_slabs, WorkList wl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PIECE = 512, CPP = 32, NDMA = BK_SRC * CPP / W16_THREADS;   // 4 LDS-DMA instructions per wave
    const SmemC32 sm = carve_c32(smem);
    char* wfrag = smem + smem_bytes_c32();
    uint8_t* tws = (uint8_t*)(wfrag + W16_WFRAG_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pt = lane & 15, kq = lane >> 4, n = pt & 3, rt = wave * 4 + (pt >> 2);
    STAMP_DECL;
    // split the weights once: fragment (g, ct, split, lane) = W_g[8*kq' + j][16*ct + i], i = lane&15, kq' = lane>>4
    for (int f = tid; f < 3 * 2 * 64; f += W16_THREADS) {
        const int g = f / 128, ct = (f >> 6) & 1, l = f & 63;
        const float* Wg = g == 0 ? W0 : (g == 1 ? W1 : W2);
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = Wg[(8 * (l >> 4) + j) * 32 + 16 * ct + (l & 15)];
        const Split3 sp = split3(w);
        char* base = wfrag + ((g * 2 + ct) * 3) * 1024 + l * 16;
        *(bf16x8*)(base) = sp.hi;
        *(bf16x8*)(base + 1024) = sp.mid;
        *(bf16x8*)(base + 2048) = sp.lo;
    }
    SCN_UNIT_RANGE();
    if (!listed && slab0 >= slab1) return;
    f32x4 pend[2];
    float* pend_ptr = nullptr;
    bool pend_valid = false;
    // chunk index of this lane's two 16-byte chunks inside a piece: n*8 + kq*2 + q, swizzled like the 8-wave kernels
    uint32_t cqs[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) cqs[q] = (uint32_t)((n * 8 + kq * 2 + q) ^ (n >> 1)) << 4;
    const size_t slab_bytes = (size_t)n_cols * PIECE;
    SCN_UNIT_BEGIN()
        wait_all_and_barrier();
        BlockMeta m;
        {
            m.row0 = P.blk_row0[b];
            m.rows = P.blk_rows[b];
            const int sp0 = P.src_ptr[b];
            m.nsrc = P.src_ptr[b + 1] - sp0;
            m.w = P.width[b];
            const int ep = P.ell_ptr[b];
            for (int i = tid; i < m.nsrc; i += W16_THREADS) sm.srcrows[i] = P.src_rows[sp0 + i];
            for (int i = tid; i < m.w * m.rows; i += W16_THREADS) {
                sm.enc[i] = P.ell_enc[ep + i];
                sm.v[i] = P.ell_v[ep + i];
            }
            if (tid < BK_R) sm.self[tid] = P.self_slot[(size_t)b * BK_R + tid];
            if (tid < W16_WAVES) tws[tid] = P.tile_w4[b * W16_WAVES + tid];
        }
        __syncthreads();
        const int tw = tws[wave];                    // width of this wave's own 4 rows
        const int twu = EXT0 ? 0 : P.tile_wu4[b * W16_WAVES + wave];
        const int rtc = rt < m.rows ? rt : m.rows - 1;
        // LDS-DMA instruction k of this wave moves chunks c = (k * 16 + wave) * 64 + lane: slot = 2 * (k * 16 + wave) + (lane >> 5),
        // position lane & 31; the swizzled source chunk depends on slot & 3 = ((wave & 1) << 1) | (lane >> 5) only: ONE lane
        // constant for every k, the slot's source row comes from LDS at a lane constant + 128 * k bytes
        const int total = m.nsrc * CPP;
        const uint32_t dchunk = (uint32_t)swz32(((wave & 1) << 1) | (lane >> 5), lane & 31) * 16;
        const int32_t* my_src = sm.srcrows + 2 * wave + (lane >> 5);          // + 32 * k
        auto dma = [&](int k, const char* Xs, char* buf) {
            const int base = (k * W16_WAVES + wave) * 64;
            if (base + lane < total)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Xs + ((uint32_t)my_src[32 * k] * PIECE + dchunk)),
                                                 (__attribute__((address_space(3))) void*)(buf + base * 16), 16, 0, 0);
        };
        auto slab_base = [&](int slab) {               // wave-uniform: SGPR base + 32-bit lane offset in the LDS-DMA
            const uint64_t v = (uint64_t)((const char*)X + (size_t)slab * slab_bytes);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
            return (const char*)(((uint64_t)hi << 32) | lo);
        };
#pragma unroll
        for (int i = 0; i < NDMA; ++i) dma(i, slab_base(SLAB_AT(0)), sm.buf(0));
#ifndef SCN_FWD_NO_STAGGER
        // STAGGER.  A slab visit has two phases per wave: GATHER (LDS reads + FMAs -- with all sixteen waves in it at once the
        // LDS pipe is the limit) and CONTRACT (split + MFMA + activation: matrix pipe and VALU).  Waves 8-15 run half a visit
        // behind waves 0-7 (every SIMD hosts two of each): while one half gathers slab v the other contracts slab v - 1.  The
        // loop walks HALF-STEPS hs; a wave gathers at hs = 2v + late and contracts at hs = 2v + 1 + late; the slab barrier stays
        // at the even half-steps.  A late wave carries its gathered z across the barrier; its share of the next LDS-DMA is one
        // visit further ahead (slab v + 2), the early waves' as before (slab v + 1).
        const int late = wave >> 3;
        if (late && n_it > 1) {                        // the late waves' share of slab 1 has no contraction to ride in
#pragma unroll
            for (int i = 0; i < NDMA; ++i) dma(i, slab_base(SLAB_AT(1)), sm.buf(1));
        }
#else
        const int late = 0;
#endif
        f32x4 z[3][2];                                 // [segment][chunk q]: channels 8*kq + 4*q .. +3 of this lane's point
        for (int hs = 0; hs <= 2 * n_it; ++hs) {
            if ((hs & 1) == 0) {
                STAMP_START();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                STAMP_ADD(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                STAMP_ADD(1);
            }
            const int hv = hs - late;
            if (hv < 0 || hv >= 2 * n_it) continue;
            const int it = hv >> 1;
            const int slab = SLAB_AT(it);
            if ((hv & 1) == 0) {
                // ---------------- GATHER
                const uint32_t cb[2] = {cqs[0] | (uint32_t)((it & 1) << 16), cqs[1] | (uint32_t)((it & 1) << 16)};
                if (EXT0) {
                    const float* x0 = X0 + (((size_t)slab * n_rows + m.row0 + rtc) * BK_NS + n) * 32 + 8 * kq;
                    z[0][0] = *(const f32x4*)(x0);           // requested before the gather: its latency hides under it
                    z[0][1] = *(const f32x4*)(x0 + 4);
                    f32x4 unused[2];
                    gather_c32<2, false>(sm, rtc, m.w, tw, 0, cb, z[1], z[2], unused);
                } else {
                    gather_c32<2>(sm, rtc, m.w, tw, twu, cb, z[0], z[1], z[2]);
                }
                STAMP_ADD(2);
                continue;
            }
            // ---------------- CONTRACT: out^T tile (16 channels x 16 points) x 2 channel tiles; the LDS-DMA of the next slab and
            // the previous tile's two 16-byte stores ride inside the chains
            const int vdma = it + 1 + late;
            const bool more = vdma < n_it;
            const char* Xn = slab_base(more ? SLAB_AT(vdma) : slab);
            char* nbuf = sm.buf(vdma & 1);
            {
                const f32x4 prev0 = pend[0], prev1 = pend[1];
                float* const prev_ptr = pend_ptr;
                const bool prev_valid = pend_valid && pend_ptr != nullptr;
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    float x8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) x8[j] = z[g][j >> 2][j & 3];
                    const Split3 zs = split3(x8);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const char* wb = wfrag + ((g * 2 + ct) * 3) * 1024 + lane * 16;
                        const bf16x8 wh = *(const bf16x8*)(wb), wm = *(const bf16x8*)(wb + 1024), wl = *(const bf16x8*)(wb + 2048);
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, zs.hi, acc[ct], 0, 0, 0);
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, zs.lo, acc[ct], 0, 0, 0);
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, zs.mid, acc[ct], 0, 0, 0);
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, zs.hi, acc[ct], 0, 0, 0);
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, zs.mid, acc[ct], 0, 0, 0);
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, zs.hi, acc[ct], 0, 0, 0);
                        const int k = g * 2 + ct;                         // 6 side slots: 4 DMA + 2 stores
                        if (k < NDMA) {
                            if (more) dma(k, Xn, nbuf);
                        } else if (prev_valid) {
                            *(f32x4*)(prev_ptr + 16 * (k - NDMA)) = (k - NDMA) == 0 ? prev0 : prev1;
                        }
                    }
                }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pend[ct][r] = act_apply_fast(ACT, acc[ct][r]);
            }
            // D: column = lane&15 = point, row = 4*kq + r = channel within the 16-channel tile
            pend_ptr = out + (((size_t)slab * n_rows + m.row0 + rt) * BK_NS + n) * 32 + 4 * kq;
            pend_valid = rt < m.rows;
            STAMP_ADD(3);
        }
    }
    STAMP_FLUSH();
    if (pend_ptr && pend_valid) {
        *(f32x4*)(pend_ptr) = pend[0];
        *(f32x4*)(pend_ptr + 16) = pend[1];
    }
}

// ------------------------------------------------------------------------------------------------
// forward, C_in = C_out = 16, sixteen waves, exact bf16 three-way split -- the C=32 machinery on TWO slabs at a time:
// a staged 512-byte piece is [slab A: 4 trajectories x 16 channels | slab B: the same], so the plan, the LDS image, the
// XOR addressing and gather_c32 are shared with the C=32 kernels (the LDS-DMA takes a per-lane source address, which is
// all the pairing needs).  wave = 4 rows; lane = (point = lane&15 -> row, trajectory n; channel quad g = lane>>4) holds
// chunk q = the same point of slab A (q=0) and slab B (q=1): two 16-point MFMA tiles.
//   per tile: v_mfma_f32_16x16x32_bf16 over K = [segment 0 | segment 1] (4 + 4 channels per lane) and
//             v_mfma_f32_16x16x16_bf16 over segment 2, each as the six products of the split; weights stay in registers.
// ------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
struct Split3x4 { s16x4 hi, mid, lo; };
__device__ __forceinline__ Split3x4 split3_4(const f32x4 x) {
    uint32_t h[2], m[2], l[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        h[i] = pack_hi16(a, b);
        const float ra = a - trunc_bf16(a), rb = b - trunc_bf16(b);
        m[i] = pack_hi16(ra, rb);
        const float sa = ra - trunc_bf16(ra), sb = rb - trunc_bf16(rb);
        l[i] = pack_hi16(sa, sb);
    }
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    Split3x4 s;
    s.hi = __builtin_bit_cast(s16x4, u32x2{h[0], h[1]});
    s.mid = __builtin_bit_cast(s16x4, u32x2{m[0], m[1]});
    s.lo = __builtin_bit_cast(s16x4, u32x2{l[0], l[1]});
    return s;
}

template <int ACT>
__global__ __launch_bounds__(W16_THREADS, 4) void fwd_c16_w16_kernel(PlanDev P, const float* __restrict__ X,
                                                                     const float* __restrict__ W0,
                                                                     const float* __restrict__ W1,
                                                                     const float* __restrict__ W2,
                                                                     float* __restrict__ out, int n_rows, int n_cols,
                                                                     int n_slabs, WorkList wl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CPP = 32, NDMA = BK_SRC * CPP / W16_THREADS;   // 4 LDS-DMA instructions per wave and slab pair
    const SmemC32 sm = carve_c32(smem);
    uint8_t* tws = (uint8_t*)(smem + smem_bytes_c32());
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pt = lane & 15, g = lane >> 4, n = pt & 3, rt = wave * 4 + (pt >> 2);
    // A fragments: lane (out channel i = lane&15, k-group g): K=32 block = W0 rows 4g..4g+3 then W1 rows 4g..4g+3; K=16: W2
    Split3 wa;
    Split3x4 wb;
    {
        float w[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w[j] = W0[(4 * g + j) * 16 + pt];
            w[4 + j] = W1[(4 * g + j) * 16 + pt];
        }
        wa = split3(w);
        f32x4 w2;
#pragma unroll
        for (int j = 0; j < 4; ++j) w2[j] = W2[(4 * g + j) * 16 + pt];
        wb = split3_4(w2);
    }
    SCN_UNIT_RANGE();
    if (!listed && slab0 >= slab1) return;
    f32x4 pend[2];
    float* pend_ptr[2] = {nullptr, nullptr};
    // chunk of tile q inside a piece: q*16 + n*4 + g, swizzled like the DMA image (swz32: pos ^ bit 4 of the chunk ^ slot part)
    uint32_t cqs[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) cqs[q] = (uint32_t)((q * 16 + n * 4 + g) ^ q) << 4;
    const size_t slab_bytes = (size_t)n_cols * 256;
    SCN_UNIT_BEGIN()
        wait_all_and_barrier();
        BlockMeta m;
        {
            m.row0 = P.blk_row0[b];
            m.rows = P.blk_rows[b];
            const int sp0 = P.src_ptr[b];
            m.nsrc = P.src_ptr[b + 1] - sp0;
            m.w = P.width[b];
            const int ep = P.ell_ptr[b];
            for (int i = tid; i < m.nsrc; i += W16_THREADS) sm.srcrows[i] = P.src_rows[sp0 + i];
            for (int i = tid; i < m.w * m.rows; i += W16_THREADS) {
                sm.enc[i] = P.ell_enc[ep + i];
                sm.v[i] = P.ell_v[ep + i];
            }
            if (tid < BK_R) sm.self[tid] = P.self_slot[(size_t)b * BK_R + tid];
            if (tid < W16_WAVES) tws[tid] = P.tile_w4[b * W16_WAVES + tid];
        }
        __syncthreads();
        const int tw = tws[wave];
        const int twu = P.tile_wu4[b * W16_WAVES + wave];
        const int rtc = rt < m.rows ? rt : m.rows - 1;
        uint32_t goff[NDMA];
        const bool second = (lane >> 4) & 1;          // this lane's chunks come from slab B: bit 4 of d = bit 4 of pos = lane bit 4
        const int total = m.nsrc * CPP;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int c = (i * W16_WAVES + wave) * 64 + lane;
            const int slot = c / CPP, d = swz32(slot, c % CPP);
            goff[i] = c < total ? (uint32_t)sm.srcrows[slot] * 256 + (d & 15) * 16 : 0u;
        }
        auto dma = [&](int k, const char* XA, const char* XB, char* buf) {
            const int base = (k * W16_WAVES + wave) * 64;
            if (base + lane < total)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((second ? XB : XA) + goff[k]),
                                                 (__attribute__((address_space(3))) void*)(buf + base * 16), 16, 0, 0);
        };
        const int n_pairs = (n_it + 1) >> 1;
        {
            const char* XA = (const char*)X + (size_t)SLAB_AT(0) * slab_bytes;
            const char* XB = n_it > 1 ? (const char*)X + (size_t)SLAB_AT(1) * slab_bytes : XA;
#pragma unroll
            for (int i = 0; i < NDMA; ++i) dma(i, XA, XB, sm.buf(0));
        }
        for (int ip = 0; ip < n_pairs; ++ip) {
            const bool has_b = 2 * ip + 1 < n_it;
            const int slab_a = SLAB_AT(2 * ip), slab_b = has_b ? SLAB_AT(2 * ip + 1) : slab_a;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const bool more = ip + 1 < n_pairs;
            const char* XnA = (const char*)X + (size_t)(more ? SLAB_AT(2 * ip + 2) : slab_a) * slab_bytes;
            const char* XnB = (more && 2 * ip + 3 < n_it) ? (const char*)X + (size_t)SLAB_AT(2 * ip + 3) * slab_bytes : XnA;
            char* nbuf = sm.buf((ip + 1) & 1);
            f32x4 z[3][2];                 // [segment][tile q]: channels 4g..4g+3 of this lane's point in slab A / B
            {
                const uint32_t cb[2] = {cqs[0] | (uint32_t)((ip & 1) << 16), cqs[1] | (uint32_t)((ip & 1) << 16)};
                gather_c32<2>(sm, rtc, m.w, tw, twu, cb, z[0], z[1], z[2]);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)                        // the previous pair's results: a whole gather phase old by the
                if (pend_ptr[q]) *(f32x4*)(pend_ptr[q]) = pend[q];   // time the next vmcnt(0) comes
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float x8[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { x8[j] = z[0][q][j]; x8[4 + j] = z[1][q][j]; }
                const Split3 za = split3(x8);
                const Split3x4 zb = split3_4(z[2][q]);
                // two accumulators: a chain of dependent MFMAs stays within one instruction shape
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acb = acc;
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa.lo, za.hi, acc, 0, 0, 0);     // small terms first
                acb = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wb.lo, zb.hi, acb, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa.hi, za.lo, acc, 0, 0, 0);
                acb = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wb.hi, zb.lo, acb, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa.mid, za.mid, acc, 0, 0, 0);
                acb = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wb.mid, zb.mid, acb, 0, 0, 0);
                if (more) dma(2 * q, XnA, XnB, nbuf);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa.mid, za.hi, acc, 0, 0, 0);
                acb = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wb.mid, zb.hi, acb, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa.hi, za.mid, acc, 0, 0, 0);
                acb = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wb.hi, zb.mid, acb, 0, 0, 0);
                if (more) dma(2 * q + 1, XnA, XnB, nbuf);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa.hi, za.hi, acc, 0, 0, 0);
                acb = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wb.hi, zb.hi, acb, 0, 0, 0);
                acc += acb;
#pragma unroll
                for (int r = 0; r < 4; ++r) pend[q][r] = act_apply_fast(ACT, acc[r]);
            }
            // D: column = lane&15 = point, row = 4*g + r = output channel
            const bool valid = rt < m.rows;
            pend_ptr[0] = valid ? out + (((size_t)slab_a * n_rows + m.row0 + rt) * BK_NS + n) * 16 + 4 * g : nullptr;
            pend_ptr[1] = valid && has_b ? out + (((size_t)slab_b * n_rows + m.row0 + rt) * BK_NS + n) * 16 + 4 * g : nullptr;
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (pend_ptr[q]) *(f32x4*)(pend_ptr[q]) = pend[q];
}

// ------------------------------------------------------------------------------------------------
// forward, C_in = C_out = 16   (v_mfma_f32_16x16x4_f32; lane = (point p = lane&15, channel quad g = lane>>4);
// a wave's 8 rows form two 16-point tiles (rows 0-3 / 4-7) with independent accumulators)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BK_THREADS, 2) void fwd_c16_kernel(PlanDev P, const float* __restrict__ X,
                                                                const float* __restrict__ W0,
                                                                const float* __restrict__ W1,
                                                                const float* __restrict__ W2,
                                                                float* __restrict__ out, int n_rows, int n_cols,
                                                                int n_slabs, int act, WorkList wl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PIECE = 256;
    const Smem sm = carve(smem, PIECE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 15, g = lane >> 4, n = p & 3;
    const int rtA = wave * 8 + (p >> 2), rtB = rtA + 4;
    float Bw[3][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        Bw[0][s] = W0[(4 * g + s) * 16 + p];
        Bw[1][s] = W1[(4 * g + s) * 16 + p];
        Bw[2][s] = W2[(4 * g + s) * 16 + p];
    }
    SCN_UNIT_RANGE();
    if (!listed && slab0 >= slab1) return;
    const int chunk = n * 4 + g;
    // D layout: column = lane&15 (channel), row = 4*g + r (point): row-in-quad = g, trajectory = r
    const int prA = wave * 8 + g, prB = prA + 4;
    f32x4 pendA, pendB;
    float* pend_ptr = nullptr;
    int pend_rows = 0;
    SCN_UNIT_BEGIN()
        wait_all_and_barrier();
        const BlockMeta m = load_block(P, b, sm);
        __syncthreads();
        const int tw = P.tile_w[b * BK_WAVES + wave];
        const int rA = rtA < m.rows ? rtA : m.rows - 1, rB = rtB < m.rows ? rtB : m.rows - 1;
        dma_stage<PIECE, 16>((const char*)X + (size_t)SLAB_AT(0) * n_cols * PIECE, sm.buf(0), sm, m.nsrc);
        for (int it = 0; it < n_it; ++it) {
            const int slab = SLAB_AT(it);
            const char* cur = sm.buf(it & 1);
            wait_vm_and_barrier();
            if (it + 1 < n_it)
                dma_stage<PIECE, 16>((const char*)X + (size_t)SLAB_AT(it + 1) * n_cols * PIECE, sm.buf((it + 1) & 1), sm,
                                     m.nsrc);
            if (pend_ptr) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (prA < pend_rows) pend_ptr[(prA * BK_NS + r) * 16] = pendA[r];
                    if (prB < pend_rows) pend_ptr[(prB * BK_NS + r) * 16] = pendB[r];
                }
            }
            f32x4 zsA, zsB, zlA = {0.f, 0.f, 0.f, 0.f}, zuA = zlA, zlB = zlA, zuB = zlA;
            {
                const int sA = sm.self[rA], sB = sm.self[rB];
                zsA = *(const f32x4*)(cur + sA * PIECE + swz16(sA, chunk) * 16);
                zsB = *(const f32x4*)(cur + sB * PIECE + swz16(sB, chunk) * 16);
            }
            {
                const int rbA = rA * m.w, rbB = rB * m.w;
                EllPair nA = ell_load(sm, rbA), nB = ell_load(sm, rbB);
                for (int t = 0; t < tw; t += 2) {
                    const EllPair eA = nA, eB = nB;
                    if (t + 2 < tw) { nA = ell_load(sm, rbA + t + 2); nB = ell_load(sm, rbB + t + 2); }
                    const f32x4 dA0 = *(const f32x4*)(cur + eA.s0 * PIECE + swz16(eA.s0, chunk) * 16);
                    const f32x4 dA1 = *(const f32x4*)(cur + eA.s1 * PIECE + swz16(eA.s1, chunk) * 16);
                    const f32x4 dB0 = *(const f32x4*)(cur + eB.s0 * PIECE + swz16(eB.s0, chunk) * 16);
                    const f32x4 dB1 = *(const f32x4*)(cur + eB.s1 * PIECE + swz16(eB.s1, chunk) * 16);
                    __builtin_amdgcn_sched_barrier(0);      // all four reads (and the next pair's metadata) in flight before the FMAs
                    zlA += eA.v[0] * dA0; zuA += eA.v[1] * dA0; zlA += eA.v[2] * dA1; zuA += eA.v[3] * dA1;
                    zlB += eB.v[0] * dB0; zuB += eB.v[1] * dB0; zlB += eB.v[2] * dB1; zuB += eB.v[3] * dB1;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = accA;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                accA = __builtin_amdgcn_mfma_f32_16x16x4f32(zsA[s], Bw[0][s], accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(zsB[s], Bw[0][s], accB, 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                accA = __builtin_amdgcn_mfma_f32_16x16x4f32(zlA[s], Bw[1][s], accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(zlB[s], Bw[1][s], accB, 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                accA = __builtin_amdgcn_mfma_f32_16x16x4f32(zuA[s], Bw[2][s], accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(zuB[s], Bw[2][s], accB, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pendA[r] = act_apply_fast(act, accA[r]);
                pendB[r] = act_apply_fast(act, accB[r]);
            }
            pend_ptr = out + ((size_t)slab * n_rows + m.row0) * (BK_NS * 16) + p;
            pend_rows = m.rows;
        }
    }
    if (pend_ptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (prA < pend_rows) pend_ptr[(prA * BK_NS + r) * 16] = pendA[r];
            if (prB < pend_rows) pend_ptr[(prB * BK_NS + r) * 16] = pendB[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward, C_in = 1 -> C_out = C (first layer, TE:143-147 with flow (E,1)): three gathered scalars per point,
// then out = act(x*w0 + lo*w1 + up*w2) written as coalesced float4.
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(BK_THREADS, 6) void fwd_c1_kernel(PlanDev P, const float* __restrict__ X,
                                                               const float* __restrict__ W0,
                                                               const float* __restrict__ W1,
                                                               const float* __restrict__ W2,
                                                               float* __restrict__ out, float* __restrict__ Yout,
                                                               int n_rows, int n_cols, int n_slabs, int act, WorkList wl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PIECE = 16, CQ = C / 4;
    const Smem sm = carve(smem, PIECE);
    float* Z = (float*)(smem + smem_bytes(PIECE));                    // [2][BK_R*BK_NS][3]
    const int tid = threadIdx.x;
    const int cq = tid % CQ;                                          // constant per thread: 512 % CQ == 0
    const f32x4 w0 = *(const f32x4*)(W0 + cq * 4), w1 = *(const f32x4*)(W1 + cq * 4), w2 = *(const f32x4*)(W2 + cq * 4);
    SCN_UNIT_RANGE();
    if (!listed && slab0 >= slab1) return;
    SCN_UNIT_BEGIN()
        wait_all_and_barrier();
        const BlockMeta m = load_block(P, b, sm);
        __syncthreads();
        dma_stage<PIECE, 0>((const char*)X + (size_t)SLAB_AT(0) * n_cols * PIECE, sm.buf(0), sm, m.nsrc);
        for (int it = 0; it < n_it; ++it) {
            const int slab = SLAB_AT(it);
            const float* st = (const float*)sm.buf(it & 1);
            float* Zs = Z + (it & 1) * (BK_R * BK_NS * 3);
            wait_vm_and_barrier();
            if (it + 1 < n_it)
                dma_stage<PIECE, 0>((const char*)X + (size_t)SLAB_AT(it + 1) * n_cols * PIECE, sm.buf((it + 1) & 1), sm,
                                    m.nsrc);
            if (tid < m.rows * BK_NS) {
                const int rt = tid >> 2, n = tid & 3;
                float zs = st[sm.self[rt] * 4 + n], zl = 0.f, zu = 0.f;
                const int tw = P.tile_w[b * BK_WAVES + (rt >> 3)];
                const int rb = rt * m.w;
                for (int t = 0; t < tw; t += 2) {
                    const EllPair e = ell_load(sm, rb + t);
                    const float d0 = st[e.s0 * 4 + n], d1 = st[e.s1 * 4 + n];
                    zl = fmaf(e.v[0], d0, zl);
                    zu = fmaf(e.v[1], d0, zu);
                    zl = fmaf(e.v[2], d1, zl);
                    zu = fmaf(e.v[3], d1, zu);
                }
                Zs[tid * 3] = zs; Zs[tid * 3 + 1] = zl; Zs[tid * 3 + 2] = zu;
            }
            __syncthreads();   // Z visible (the DMA in flight is drained here too: acceptable for this light kernel)
            if (Yout) {        // the shifted input (x, S_lo x, S_up x, 0) per point, as the first-layer weight gradient takes it
                float* yb = Yout + ((size_t)slab * n_rows + m.row0) * (BK_NS * Y_STRIDE);
                for (int i = tid; i < m.rows * BK_NS; i += BK_THREADS)
                    *(f32x4*)(yb + 4 * i) = f32x4{Zs[3 * i], Zs[3 * i + 1], Zs[3 * i + 2], 0.f};
            }
            float* o = out + ((size_t)slab * n_rows + m.row0) * (BK_NS * C);
            const int total = m.rows * BK_NS * CQ;
            for (int idx = tid; idx < total; idx += BK_THREADS) {
                const int pt = idx / CQ;
                const float zs = Zs[pt * 3], zl = Zs[pt * 3 + 1], zu = Zs[pt * 3 + 2];
                f32x4 v = zs * w0 + zl * w1 + zu * w2;
                v[0] = act_apply_fast(act, v[0]); v[1] = act_apply_fast(act, v[1]);
                v[2] = act_apply_fast(act, v[2]); v[3] = act_apply_fast(act, v[3]);
                __builtin_nontemporal_store(v, (f32x4*)(o + (size_t)idx * 4));   // streamed out: keep the L2 for the gathered input
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, c_dz = c_aux = 32
//   G = [dz | S_lower^T dz | S_upper^T dz] gathered like the forward; dx = (G @ [W0;W1;W2]^T) * act'(aux);
//   dW_g += aux^T G_g via MFMA with the points on the contraction axis (G transposed through a per-wave LDS patch,
//   16 points at a time).
// ------------------------------------------------------------------------------------------------
// Ordering between this wave's own LDS writes and reads of the transpose patch: DS operations of one wave execute in
// issue order, so no s_waitcnt is needed for cross-lane visibility -- only a compiler fence that keeps the order.
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

constexpr int T32_STRIDE = 36;    // floats per point row of the transpose patch (32 + 4 pad: 144 B)
constexpr int T16_STRIDE = 20;    // floats per point row (16 + 4 pad: 80 B)

__global__ __launch_bounds__(BK_THREADS, 2) void bwd_c32_kernel(PlanDev P, const float* __restrict__ DZ,
                                                                const float* __restrict__ W0,
                                                                const float* __restrict__ W1,
                                                                const float* __restrict__ W2,
                                                                const float* __restrict__ aux, float* __restrict__ dx,
                                                                float* __restrict__ partial, int n_rows, int n_cols,
                                                                int n_slabs, int act) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PIECE = 512, CPP = 32, NDMA = BK_SRC * CPP / BK_THREADS;
    const Smem sm = carve(smem, PIECE);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* patch = (float*)(smem + smem_bytes(PIECE)) + wave * (16 * T32_STRIDE);
    const int p = lane & 31, h = lane >> 5, n = p & 3, rt = wave * 8 + (p >> 2);
    STAMP_DECL;
    // dgrad B operands: k-step s of segment g contracts dz channel 16*h + s against W_g[ca = p][c]
    float Bt[3][16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        Bt[0][s] = W0[p * 32 + 16 * h + s];
        Bt[1][s] = W1[p * 32 + 16 * h + s];
        Bt[2][s] = W2[p * 32 + 16 * h + s];
    }
    f32x16 dWacc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) dWacc[g][r] = 0.f;
    int b, b_end, b_stride;
    block_range(P.n_blocks, b, b_end, b_stride);
    SCN_SLAB_RANGE();
    int cq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cq[q] = (n * 8 + h * 4 + q) ^ (n >> 1);
    const size_t slab_bytes = (size_t)n_cols * PIECE;
    if (slab0 < slab1)
    for (; b < b_end; b += b_stride) {
        wait_all_and_barrier();
        const BlockMeta m = load_block(P, b, sm);
        __syncthreads();
        const int tw = P.tile_w[b * BK_WAVES + wave];
        const int rtc = rt < m.rows ? rt : m.rows - 1;
        uint32_t goff[NDMA];                          // per-lane LDS-DMA source offsets, constant over the block's slabs
        const int total = m.nsrc * CPP;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int c = (i * BK_WAVES + wave) * 64 + lane;
            const int slot = c / CPP, pos = c % CPP;
            goff[i] = c < total ? (uint32_t)sm.srcrows[slot] * PIECE + swz32(slot, pos) * 16 : 0u;
        }
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int base = (i * BK_WAVES + wave) * 64;
            if (base + lane < total)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)DZ + (size_t)slab0 * slab_bytes + goff[i]),
                                                 (__attribute__((address_space(3))) void*)(sm.buf(0) + base * 16), 16, 0, 0);
        }
        const int rows_left = m.rows - wave * 8;      // rows of this wave's tile that exist (<= 0: none)
        for (int slab = slab0; slab < slab1; ++slab) {
            const char* cur = sm.buf((slab - slab0) & 1);
            // aux / dx address of point pt(r) = (r&3) + 8*(r>>2) + 4*h, channel ca = p : tbase + pt*32
            const size_t tuni = (((size_t)slab * n_rows + m.row0 + wave * 8) * BK_NS) * 32;   // wave-uniform tile base
            const float* ap = aux + (rows_left > 0 ? tuni : 0);   // waves without rows read (and discard) element 0
            float* dp = dx ? dx + tuni : nullptr;
            const int L0 = p + 128 * h;                       // lane part of the element offset: pt*32 + p
            STAMP_START();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP_ADD(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            STAMP_ADD(1);
            f32x4 gs[4], gl[4], gu[4];
            {
                const int slot = sm.self[rtc];
                const int sx = (((slot >> 1) & 1) << 2) | ((slot & 1) << 1);
                const char* base = cur + slot * PIECE;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    gs[q] = *(const f32x4*)(base + ((cq[q] ^ sx) << 4));
                    gl[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    gu[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                const int rb = rtc * m.w;
                EllPair en = ell_load(sm, rb);
                for (int t = 0; t < tw; t += 2) {
                    const EllPair e = en;
                    if (t + 2 < tw) en = ell_load(sm, rb + t + 2);
                    const int x0 = (((e.s0 >> 1) & 1) << 2) | ((e.s0 & 1) << 1);
                    const int x1 = (((e.s1 >> 1) & 1) << 2) | ((e.s1 & 1) << 1);
                    const char* b0 = cur + e.s0 * PIECE;
                    const char* b1 = cur + e.s1 * PIECE;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 d0 = *(const f32x4*)(b0 + ((cq[q] ^ x0) << 4));
                        gl[q] += e.v[0] * d0;
                        gu[q] += e.v[1] * d0;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 d1 = *(const f32x4*)(b1 + ((cq[q] ^ x1) << 4));
                        gl[q] += e.v[2] * d1;
                        gu[q] += e.v[3] * d1;
                    }
                }
            }
            STAMP_ADD(2);
            // dgrad chain: dH[point][ca] = sum_k G[point][k] * W[ca][k]   (D: column = lane&31 = ca, rows = points).
            // Vector-memory issue rides inside the chain (the wave sits on the accumulator dependency anyway):
            // first the LDS-DMA of the next slab, then this tile's 16 aux values (used right after the chain).
            float a[16];
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const bool more = slab + 1 < slab1;
            const char* Xn = (const char*)DZ + (size_t)(slab + 1) * slab_bytes;
            char* nbuf = sm.buf((slab + 1 - slab0) & 1);
#pragma unroll
            for (int s = 0; s < 48; ++s) {
                const int g = s >> 4, ss = s & 15;
                const f32x4* G = g == 0 ? gs : (g == 1 ? gl : gu);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(G[ss >> 2][ss & 3], Bt[g][ss], acc, 0, 0, 0);
                if ((s & 1) == 1) {
                    const int k = s >> 1;                         // 24 side slots
                    if (k < NDMA) {
                        const int base = (k * BK_WAVES + wave) * 64;
                        if (more && base + lane < total)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Xn + goff[k]),
                                                             (__attribute__((address_space(3))) void*)(nbuf + base * 16), 16, 0, 0);
                    } else {
                        const int r = k - NDMA;
                        const bool ok = 2 * (r >> 2) + h < rows_left;          // row-in-tile of point pt(r,h) exists
                        const float v = ap[ok ? L0 + ((r & 3) + 8 * (r >> 2)) * 32 : 0];
                        a[r] = ok ? v : 0.f;
                    }
                }
            }
            STAMP_ADD(3);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= act_grad_from_output(act, a[r]);
            // dW_g += aux^T G_g : A[i = ca][k = point] = a[s], B[k = point][j = c] = G_g[pt(s,h)][c = p].
            // Points 0..15 are k-steps s = 0..7, points 16..31 are s = 8..15; the patch holds 16 points at a time.
            // The 16 dx stores ride inside these chains.
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const f32x4* src = g == 0 ? gs : (g == 1 ? gl : gu);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if ((p >> 4) == half) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) *(f32x4*)(patch + (p & 15) * T32_STRIDE + 16 * h + 4 * q) = src[q];
                    }
                    wave_lds_sync();
#pragma unroll
                    for (int s8 = 0; s8 < 8; ++s8) {
                        const int s = half * 8 + s8;
                        const int pt16 = (s8 & 3) + 8 * (s8 >> 2) + 4 * h;      // point index within this half (0..15)
                        dWacc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], patch[pt16 * T32_STRIDE + p], dWacc[g], 0, 0, 0);
                        const int k = (g * 2 + half) * 8 + s8;                   // 48 MFMAs, a store after every third
                        if (dp && (k % 3) == 2) {
                            const int r = k / 3;
                            if (2 * (r >> 2) + h < rows_left) dp[L0 + ((r & 3) + 8 * (r >> 2)) * 32] = acc[r];
                        }
                    }
                    wave_lds_sync();
                }
            }
            STAMP_ADD(4);
        }
    }
    STAMP_FLUSH();
    // reduce the eight waves' dW tiles in a fixed order and emit this workgroup's partial [ca][slot*32 + c]
    wait_all_and_barrier();
    float* red = (float*)sm.buf(0);                             // 8 waves * 3072 floats = 96 KB (both stage buffers)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ca = (r & 3) + 8 * (r >> 2) + 4 * h;
            red[wave * 3072 + ca * 96 + g * 32 + p] = dWacc[g][r];
        }
    __syncthreads();
    float* outp = partial + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 3072;
    for (int i = threadIdx.x; i < 3072; i += BK_THREADS) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < BK_WAVES; ++w) s += red[w * 3072 + i];
        outp[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// backward, c_dz = c_aux = 32 on the bf16 MFMA with the exact three-way split (default; see the note above fwd_c32_w16).
// Same block / slab pipeline and lane roles as bwd_c32_kernel; what changes is the contraction (v_mfma_f32_32x32x16_bf16):
//   dgrad      D[pt][ca]  = sum_{g,c} G_g[pt][c] W_g[ca][c]   A = split(G_g) in its gather layout (lane = point p, channels
//                           16h+8t+j), B = split(W_g) fragments kept in LDS (18 KB, where the fp32 kernel has its patch)
//   transpose  T_g[pt][c] = sum_k G_g[pt][k] I[k][c]           the same A fragments against a 0/1 selection matrix: the D
//                           layout hands every lane 16 points of ONE channel (exact: hi+mid+lo re-add to the fp32 value),
//                           which is the B operand the weight gradient needs -- no LDS round trip
//   dW_g       D[ca][c]  += sum_pt aux[pt][ca] T_g[pt][c]      A = split(aux tile) straight from the dgrad D layout
// 90 MFMAs x 32 cycles per 32-point tile instead of 96 x 64.  MFMA and VALU time add up on a SIMD (tools/ubench), so
// the VALU instruction count (5.5 per split value) matters as much as the MFMA count.
// ------------------------------------------------------------------------------------------------
constexpr int B32_WFRAG_BYTES = 3 * 2 * 3 * 64 * 16;      // [segment][k-step][split][lane] x 8 bf16
static_assert(B32_WFRAG_BYTES + 2048 <= 160 * 1024 - (2 * BK_SRC * 512 + BK_ELL_CAP * 10 + BK_SRC * 4 + BK_R + 16), "weight + selection fragments fit behind the staging buffers");

// EXT0 (see fwd_c32_w16_kernel): segment 0 of G is the upstream gradient itself (DZ0, own rows from HBM), the staged tensor
// DZ is S^T dz: segment 1 its own row, segment 2 its gathered shift (S^T)^2 dz.
// PAIR: C = 16 on TWO slabs per visit -- the staged 512-byte piece is a point's 32 VIRTUAL channels (slab A's 16, slab B's 16:
// the LDS-DMA's per-lane source address does the pairing), the weights are the block-diagonal diag(W, W), and aux / dx /
// the weight-gradient blocks are addressed per slab; everything between is the C = 32 kernel as it stands.
// y broadcast of the FIRST variant below: lane 32 h + 4 (R & 7) + g of yv0 (R < 8) / yv1 holds y[pt(R, h)][g]; a bit-mask
// ds_swizzle with and = 0, or = j makes every lane read lane j of its own half (the pattern must be a literal: templates).
template <int R>
__device__ __forceinline__ void first_dw_step(float yv0, float yv1, float dx, float (&dwf)[3]) {
    const int src = __float_as_int(R < 8 ? yv0 : yv1);
    dwf[0] = fmaf(__int_as_float(__builtin_amdgcn_ds_swizzle(src, (4 * (R & 7) + 0) << 5)), dx, dwf[0]);
    dwf[1] = fmaf(__int_as_float(__builtin_amdgcn_ds_swizzle(src, (4 * (R & 7) + 1) << 5)), dx, dwf[1]);
    dwf[2] = fmaf(__int_as_float(__builtin_amdgcn_ds_swizzle(src, (4 * (R & 7) + 2) << 5)), dx, dwf[2]);
}
template <int... Rs>
__device__ __forceinline__ void first_dw_tile(std::integer_sequence<int, Rs...>, float yv0, float yv1, const f32x16& dx,
                                              float (&dwf)[3]) {
    (first_dw_step<Rs>(yv0, yv1, dx[Rs], dwf), ...);
}

// PAIR form of the broadcast: a half wave is two 16-lane groups (slab A's channels, slab B's), each with its own points' y;
// lane 16 s + j of register q holds value 16 q + j of its slab's 48 values [r][g]; and = 0x10 keeps a lane in its group.
template <int R>
__device__ __forceinline__ void first_dw_step_pair(const float (&yv)[3], float dx, float (&dwf)[3]) {
#define SCN_SWZ_PAIR(G) \
    __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(yv[(3 * R + G) / 16]), 0x10 | (((3 * R + G) % 16) << 5)))
    dwf[0] = fmaf(SCN_SWZ_PAIR(0), dx, dwf[0]);
    dwf[1] = fmaf(SCN_SWZ_PAIR(1), dx, dwf[1]);
    dwf[2] = fmaf(SCN_SWZ_PAIR(2), dx, dwf[2]);
#undef SCN_SWZ_PAIR
}
template <int... Rs>
__device__ __forceinline__ void first_dw_tile_pair(std::integer_sequence<int, Rs...>, const float (&yv)[3], const f32x16& dx,
                                                   float (&dwf)[3]) {
    (first_dw_step_pair<Rs>(yv, dx[Rs], dwf), ...);
}

// FIRST: this layer's input is the FIRST layer's output (aux = H1 = act(y . W_first), y = the shifted 1-channel input saved by
// scn_conv_forward_first).  The input gradient dx = dL/d(pre-activation of layer 1) is then needed for one thing only -- the
// first layer's weight gradient dW_first[g][c] = sum_p y[p][g] dx[p][c] -- so it is contracted with y right here, in registers
// (y is broadcast across the lanes of a half wave with ds_swizzle: 48 cross-lane reads + 48 FMAs per tile), and never written:
// no 4*E*C-byte dx tensor, no separate streaming kernel over it.  DZ0 carries y in this variant.
template <int ACT, bool EXT0 = false, bool PAIR = false, bool FIRST = false>
__global__ __launch_bounds__(BK_THREADS, 2) void bwd_c32_bf16_kernel(PlanDev P, const float* __restrict__ DZ,
                                                                     const float* __restrict__ DZ0,
                                                                     const float* __restrict__ W0,
                                                                     const float* __restrict__ W1,
                                                                     const float* __restrict__ W2,
                                                                     const float* __restrict__ aux,
                                                                     float* __restrict__ dx, float* __restrict__ partial,
                                                                     int n_rows, int n_cols, int n_slabs, WorkList wl,
                                                                     float* __restrict__ partial_first = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(!(EXT0 && PAIR), "the power form exists for C = 32 only");
    static_assert(!FIRST || !EXT0, "the fused first-layer gradient exists for the plain forms (C = 32, and C = 16 on slab pairs)");
    constexpr int PIECE = 512, CPP = 32, NDMA = BK_SRC * CPP / BK_THREADS;
    constexpr int CH = PAIR ? 16 : 32;                       // channels of a stored point
    const SmemC32 sm = carve_c32(smem);
    char* wfrag = smem + smem_bytes_c32();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 31, h = lane >> 5, n = p & 3, rt = wave * 8 + (p >> 2);
    STAMP_DECL;
    // dgrad B fragments: (g, t, lane) holds W_g[ca = lane&31][c = 16*(lane>>5) + 8t + j], j = 0..7
    for (int f = threadIdx.x; f < 3 * 2 * 64; f += BK_THREADS) {
        const int g = f / 128, t = (f >> 6) & 1, l = f & 63;
        const float* Wg = g == 0 ? W0 : (g == 1 ? W1 : W2);
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ca = l & 31, cc = 16 * (l >> 5) + 8 * t + j;
            w[j] = !PAIR ? Wg[ca * 32 + cc] : ((ca >> 4) == (cc >> 4) ? Wg[(ca & 15) * 16 + (cc & 15)] : 0.f);
        }
        const Split3 sp = split3(w);
        char* base = wfrag + ((g * 2 + t) * 3) * 1024 + l * 16;
        *(bf16x8*)(base) = sp.hi;
        *(bf16x8*)(base + 1024) = sp.mid;
        *(bf16x8*)(base + 2048) = sp.lo;
    }
    // selection fragments: k-step t, B[k = 8h + j][n = p] = (p == 16h + 8t + j); kept in LDS behind the weight fragments
    // (two lane-linear B fragments; read back right where a transpose needs them -- registers are what this kernel lacks)
    char* isel = wfrag + B32_WFRAG_BYTES;
    if (threadIdx.x < 128) {
        const int t = threadIdx.x >> 6, l = threadIdx.x & 63, pp = l & 31, hh = l >> 5;
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c0 = 16 * hh + 8 * t + 2 * i;
            v[i] = (pp == c0 ? 0x3F80u : 0u) | (pp == c0 + 1 ? 0x3F800000u : 0u);
        }
        *(u32x4*)(isel + t * 1024 + l * 16) = v;
    }
    f32x16 dWacc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) dWacc[g][r] = 0.f;
    float dwf[3] = {0.f, 0.f, 0.f};                           // FIRST: this lane's share of dW_first[g][c = p]
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    SCN_UNIT_RANGE();
    uint32_t cqs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cqs[q] = (uint32_t)((n * 8 + h * 4 + q) ^ (n >> 1)) << 4;
    const size_t slab_bytes = (size_t)n_cols * (PAIR ? 256 : PIECE);
    const bool second = PAIR && (((lane >> 2) ^ wave) & 1);       // this lane's LDS-DMA chunks come from slab B (see dma_k)
    if (listed || slab0 < slab1)
    SCN_UNIT_BEGIN()
        wait_all_and_barrier();
        const BlockMeta m = load_block_c32<BK_THREADS>(P, b, sm);
        __syncthreads();
        const int tw = P.tile_w[b * BK_WAVES + wave];
        const int twu = EXT0 ? 0 : P.tile_wu[b * BK_WAVES + wave];
        const int rtc = rt < m.rows ? rt : m.rows - 1;
        // LDS-DMA instruction k of this wave moves chunks c = (k * 8 + wave) * 64 + lane: slot = 2 * (k * 8 + wave) + (lane >> 5),
        // position lane & 31.  The swizzled source chunk only depends on slot & 3 = ((wave & 1) << 1) | (lane >> 5), i.e. it is ONE
        // lane constant for every k, and the slot's source row comes from LDS at a lane constant + 64 * k: no per-k offsets in
        // registers.  PAIR: virtual chunk n*8 + q*4 + g <- slab q, chunk n*4 + g; q = bit 2 of d = ((lane>>2) ^ wave) & 1 for every k.
        const int total = m.nsrc * CPP;
        const int dq = swz32(((wave & 1) << 1) | (lane >> 5), lane & 31);
        const uint32_t dchunk = PAIR ? (uint32_t)(((dq >> 3) * 4 + (dq & 3)) * 16) : (uint32_t)(dq * 16);
        const int32_t* my_src = sm.srcrows + 2 * wave + (lane >> 5);          // + 16 * k
        auto dma_k = [&](int k, const char* Xs, char* buf) {
            const int base = (k * BK_WAVES + wave) * 64;
            if (base + lane < total) {
                const uint32_t off = (uint32_t)my_src[16 * k] * (PAIR ? 256u : (uint32_t)PIECE) + dchunk;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Xs + off),
                                                 (__attribute__((address_space(3))) void*)(buf + base * 16), 16, 0, 0);
            }
        };
        const int n_vis = PAIR ? (n_it + 1) >> 1 : n_it;          // visits: slabs, or slab pairs
        auto slab_of = [&](int vis, int q) {                     // q-th slab of a visit (the lone last slab stands in for its pair)
            if (!PAIR) return SLAB_AT(vis);
            return SLAB_AT(2 * vis + q < n_it ? 2 * vis + q : 2 * vis);
        };
        auto src_base = [&](int vis) {
            if (!PAIR) {    // wave-uniform: say so, and the LDS-DMA takes (SGPR base + 32-bit lane offset) instead of a 64-bit VGPR address
                const uint64_t v = (uint64_t)((const char*)DZ + (size_t)slab_of(vis, 0) * slab_bytes);
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
                return (const char*)(((uint64_t)hi << 32) | lo);
            }
            return (const char*)DZ + (size_t)slab_of(vis, second ? 1 : 0) * slab_bytes;
        };
#pragma unroll
        for (int i = 0; i < NDMA; ++i) dma_k(i, src_base(0), sm.buf(0));
        const int rows_left = m.rows - wave * 8;
        f32x4 G[3][4];
        for (int it = 0; it < n_vis; ++it) {
            const int slab = slab_of(it, PAIR ? p >> 4 : 0);          // PAIR: lanes p >= 16 hold slab B's channels
            const bool lane_live = !PAIR || p < 16 || 2 * it + 1 < n_it;
            const uint32_t bufbit = (uint32_t)((it & 1) << 16);
            const uint32_t cbs[4] = {cqs[0] | bufbit, cqs[1] | bufbit, cqs[2] | bufbit, cqs[3] | bufbit};
            STAMP_START();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP_ADD(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            STAMP_ADD(1);
            const size_t tuni = (((size_t)slab * n_rows + m.row0 + wave * 8) * BK_NS) * CH;
            const float* ap = aux + (rows_left > 0 ? tuni : 0);
            float* dp = dx && lane_live ? dx + tuni : nullptr;
            const int L0 = (PAIR ? p & 15 : p) + 4 * CH * h;
            // the tile's 16 aux values and the first weight fragments are requested before the gather so that their latency
            // (HBM / LDS) is covered by it; they are issued ahead of the next slab's LDS-DMA, so waiting for them later leaves
            // the DMA in flight
            float a[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = 2 * (r >> 2) + h < rows_left && lane_live;
                const float v = SCN_LD_STREAM(ap + (ok ? L0 + ((r & 3) + 8 * (r >> 2)) * CH : 0));
                a[r] = ok ? v : 0.f;
            }
            // FIRST: the tile's y records, one float per lane and register, placed for a ds_swizzle broadcast inside each half wave:
            // lane 32 h + 4 (r & 7) + g holds y[pt(r, h)][g], r < 8 in yv0 and r >= 8 (the points 16 higher) in yv1
            float yv0 = 0.f, yv1 = 0.f;
            float yvp[3] = {0.f, 0.f, 0.f};                    // PAIR form: see first_dw_step_pair (`slab` is this lane group's slab)
            if (FIRST && !PAIR) {
                const int r7 = (lane >> 2) & 7, ypt = (r7 & 3) + 8 * (r7 >> 2) + 4 * h;
                const float* yt = DZ0 + (((size_t)slab * n_rows + m.row0 + wave * 8) * BK_NS) * Y_STRIDE + ypt * Y_STRIDE + (lane & 3);
                if ((ypt >> 2) < rows_left) yv0 = yt[0];
                if ((ypt >> 2) + 4 < rows_left) yv1 = yt[16 * Y_STRIDE];
            }
            if (FIRST && PAIR) {
                const float* yt = DZ0 + (((size_t)slab * n_rows + m.row0 + wave * 8) * BK_NS) * Y_STRIDE;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int idx = 16 * q + (lane & 15), r = idx / 3, g = idx - 3 * r;
                    const int ypt = (r & 3) + 8 * (r >> 2) + 4 * h;
                    if ((ypt >> 2) < rows_left && lane_live) yvp[q] = yt[ypt * Y_STRIDE + g];
                }
            }
            constexpr int GSEQ[3] = {1, 2, 0};                           // segment order
            bf16x8 wn[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) wn[i] = *(const bf16x8*)(wfrag + lane * 16 + ((GSEQ[0] * 2 + 0) * 3 + i) * 1024);
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- GATHER: G[1], G[2]; the identity segment is four LDS reads, made right before it is needed, so
            // it never shares the register file with the other two
            if (EXT0) {
                const float* g0 = DZ0 + (((size_t)slab * n_rows + m.row0 + rtc) * BK_NS + n) * 32 + 16 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) G[0][q] = *(const f32x4*)(g0 + 4 * q);
                f32x4 unused[4];
                gather_c32<4, false>(sm, rtc, m.w, tw, 0, cbs, G[1], G[2], unused);
            } else {
                __builtin_amdgcn_s_setprio(3);
                gather_c32<4, true, false>(sm, rtc, m.w, tw, twu, cbs, G[0], G[1], G[2]);
                __builtin_amdgcn_s_setprio(1);
            }
            STAMP_ADD(2);
            // ---------------- CONTRACT
            const int vdma = it + 1;                                     // the visit whose slab this contraction stages
            const bool more = vdma < n_vis;
            const char* Xn = src_base(more ? vdma : it);
            char* nbuf = sm.buf(vdma & 1);
            auto side = [&](int k) {                                     // (all eight right after the barrier instead: +2 %)
                if (k < NDMA && more) dma_k(k, Xn, nbuf);
            };
            // Per segment g: [transpose + dgrad] over its two k-steps, then straight into dW_g.  The transpose keeps the three
            // parts of the split in THREE accumulators (T_hi, T_mid, T_lo = hi / mid / lo of G_g, one channel per lane): each
            // holds exactly bf16-representable values, so the B operand of the weight gradient is a pack (v_perm, one per two
            // values) instead of a second split3 of their sum.
            f32x16 acc;
            Split3 At[2];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int g = GSEQ[u];
                if (u == 2 && !EXT0) read_self_c32<4>(sm, rtc, cbs, G[0]);
                f32x16 Th, Tm, Tl;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float x8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) x8[j] = G[g][2 * t + (j >> 2)][j & 3];
                    const Split3 zs = split3(x8);
                    const bf16x8 wh = wn[0], wm = wn[1], wl = wn[2];
                    const bf16x8 It = *(const bf16x8*)(isel + t * 1024 + lane * 16);
                    const int k0 = (u * 2 + t) * 2;                          // two LDS-DMA side slots per group (8 used)
                    Tl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zs.lo, It, t == 0 ? zero16 : Tl, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zs.lo, wh, (u == 0 && t == 0) ? zero16 : acc, 0, 0, 0);
                    if (u * 2 + t < 5) {                                     // next group's fragments, one group ahead
                        const int un = (u * 2 + t + 1) >> 1, tn = (u * 2 + t + 1) & 1;
                        const char* wb = wfrag + ((GSEQ[un] * 2 + tn) * 3) * 1024 + lane * 16;
#pragma unroll
                        for (int i = 0; i < 3; ++i) wn[i] = *(const bf16x8*)(wb + i * 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    side(k0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zs.hi, wl, acc, 0, 0, 0);
                    side(k0 + 1);
                    Tm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zs.mid, It, t == 0 ? zero16 : Tm, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zs.mid, wm, acc, 0, 0, 0);
                    Th = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zs.hi, It, t == 0 ? zero16 : Th, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zs.mid, wh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zs.hi, wm, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zs.hi, wh, acc, 0, 0, 0);
                }
                if (u == 0) {                                         // dW A fragments from the aux tile (k-step t <-> points pt(8t + j, h))
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        float x8[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) x8[j] = a[8 * t + j];
                        At[t] = split3(x8);
                    }
                }
                if (u == 2) {
                    __builtin_amdgcn_s_setprio(0);
                    STAMP_ADD(3);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] *= act_grad_from_output(ACT, a[r]);   // dX = acc * act'(aux)
                    if (FIRST && !PAIR)                                  // dW_first[g][c = p] += y[pt(r, h)][g] * dx[pt(r, h)][p]
                        first_dw_tile(std::make_integer_sequence<int, 16>{}, yv0, yv1, acc, dwf);
                    if (FIRST && PAIR) first_dw_tile_pair(std::make_integer_sequence<int, 16>{}, yvp, acc, dwf);
                }
                // dW_g += aux^T T_g
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    u32x4 ph, pm, pl;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        ph[i] = pack_hi16(Th[8 * t + 2 * i], Th[8 * t + 2 * i + 1]);
                        pm[i] = pack_hi16(Tm[8 * t + 2 * i], Tm[8 * t + 2 * i + 1]);
                        pl[i] = pack_hi16(Tl[8 * t + 2 * i], Tl[8 * t + 2 * i + 1]);
                    }
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, ph), bm = __builtin_bit_cast(bf16x8, pm),
                                 bl = __builtin_bit_cast(bf16x8, pl);
                    const Split3& at = At[t];
                    auto store = [&](int r) {                            // only in the last segment: acc is complete there
                        if (!FIRST && u == 2 && dp && 2 * (r >> 2) + h < rows_left) SCN_ST_STREAM(dp + L0 + ((r & 3) + 8 * (r >> 2)) * CH, acc[r]);
                    };
                    const int r0 = 8 * t;
                    dWacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at.lo, bh, dWacc[g], 0, 0, 0);
                    store(r0);
                    dWacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at.hi, bl, dWacc[g], 0, 0, 0);
                    store(r0 + 1); store(r0 + 2);
                    dWacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at.mid, bm, dWacc[g], 0, 0, 0);
                    store(r0 + 3);
                    dWacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at.mid, bh, dWacc[g], 0, 0, 0);
                    store(r0 + 4); store(r0 + 5);
                    dWacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at.hi, bm, dWacc[g], 0, 0, 0);
                    store(r0 + 6);
                    dWacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at.hi, bh, dWacc[g], 0, 0, 0);
                    store(r0 + 7);
                }
            }
            STAMP_ADD(4);
        }
    }
    STAMP_FLUSH();
    wait_all_and_barrier();
    float* red = (float*)sm.buf(0);
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ca = (r & 3) + 8 * (r >> 2) + 4 * h;
            red[wave * 3072 + ca * 96 + g * 32 + p] = dWacc[g][r];
        }
    __syncthreads();
    if (PAIR) {                                               // dW = the two diagonal blocks of the virtual 32x32 gradient
        float* outp = partial + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 768;
        for (int i = threadIdx.x; i < 768; i += BK_THREADS) {
            const int ca = i / 48, rem = i - ca * 48, g = rem >> 4, cc = rem & 15;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < BK_WAVES; ++w)
                s += red[w * 3072 + ca * 96 + g * 32 + cc] + red[w * 3072 + (16 + ca) * 96 + g * 32 + 16 + cc];
            outp[i] = s;
        }
    } else {
        float* outp = partial + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 3072;
        for (int i = threadIdx.x; i < 3072; i += BK_THREADS) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < BK_WAVES; ++w) s += red[w * 3072 + i];
            outp[i] = s;
        }
    }
    if (FIRST) {                                               // dW_first partial of this workgroup: [g][c], fixed order over (wave, h)
                                                               // (PAIR: c = 16 s + channel, the two slabs are folded by the reduce kernel)
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 3; ++g) red[(wave * 2 + h) * 96 + g * 32 + p] = dwf[g];
        __syncthreads();
        if (threadIdx.x < 96) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 2 * BK_WAVES; ++w) s += red[w * 96 + threadIdx.x];
            partial_first[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 96 + threadIdx.x] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, c_dz = c_aux = 16  (16x16x4 MFMA, two 16-point tiles per wave)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BK_THREADS, 2) void bwd_c16_kernel(PlanDev P, const float* __restrict__ DZ,
                                                                const float* __restrict__ W0,
                                                                const float* __restrict__ W1,
                                                                const float* __restrict__ W2,
                                                                const float* __restrict__ aux, float* __restrict__ dx,
                                                                float* __restrict__ partial, int n_rows, int n_cols,
                                                                int n_slabs, int act, WorkList wl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PIECE = 256;
    const Smem sm = carve(smem, PIECE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* patch = (float*)(smem + smem_bytes(PIECE)) + wave * (2 * 16 * T16_STRIDE);
    const int p = lane & 15, g4 = lane >> 4, n = p & 3;
    const int rtA = wave * 8 + (p >> 2), rtB = rtA + 4;
    float Bt[3][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        Bt[0][s] = W0[p * 16 + 4 * g4 + s];
        Bt[1][s] = W1[p * 16 + 4 * g4 + s];
        Bt[2][s] = W2[p * 16 + 4 * g4 + s];
    }
    f32x4 dWacc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) dWacc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    SCN_UNIT_RANGE();
    const int chunk = n * 4 + g4;
    const int prA = wave * 8 + g4, prB = prA + 4;       // D-layout rows: point 4*g4 + r -> row-in-quad g4, trajectory r
    if (listed || slab0 < slab1)
    SCN_UNIT_BEGIN()
        wait_all_and_barrier();
        const BlockMeta m = load_block(P, b, sm);
        __syncthreads();
        const int tw = P.tile_w[b * BK_WAVES + wave];
        const int rA = rtA < m.rows ? rtA : m.rows - 1, rB = rtB < m.rows ? rtB : m.rows - 1;
        dma_stage<PIECE, 16>((const char*)DZ + (size_t)SLAB_AT(0) * n_cols * PIECE, sm.buf(0), sm, m.nsrc);
        float nA[4], nB[4];                                  // aux of the next slab (fetched one slab ahead)
        {
            const size_t tb = ((size_t)SLAB_AT(0) * n_rows + m.row0) * (BK_NS * 16) + p;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                nA[r] = prA < m.rows ? aux[tb + (prA * BK_NS + r) * 16] : 0.f;
                nB[r] = prB < m.rows ? aux[tb + (prB * BK_NS + r) * 16] : 0.f;
            }
        }
        for (int it = 0; it < n_it; ++it) {
            const int slab = SLAB_AT(it);
            const char* cur = sm.buf(it & 1);
            const size_t tbase = ((size_t)slab * n_rows + m.row0) * (BK_NS * 16) + p;
            wait_vm_and_barrier();
            float aA[4], aB[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { aA[r] = nA[r]; aB[r] = nB[r]; }
            if (it + 1 < n_it) {
                const size_t tb = ((size_t)SLAB_AT(it + 1) * n_rows + m.row0) * (BK_NS * 16) + p;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    nA[r] = prA < m.rows ? aux[tb + (prA * BK_NS + r) * 16] : 0.f;
                    nB[r] = prB < m.rows ? aux[tb + (prB * BK_NS + r) * 16] : 0.f;
                }
                dma_stage<PIECE, 16>((const char*)DZ + (size_t)SLAB_AT(it + 1) * n_cols * PIECE, sm.buf((it + 1) & 1), sm,
                                     m.nsrc);
            }
            f32x4 G[3][2];
            {
                const int sA = sm.self[rA], sB = sm.self[rB];
                G[0][0] = *(const f32x4*)(cur + sA * PIECE + swz16(sA, chunk) * 16);
                G[0][1] = *(const f32x4*)(cur + sB * PIECE + swz16(sB, chunk) * 16);
                G[1][0] = G[1][1] = G[2][0] = G[2][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            {
                const int rbA = rA * m.w, rbB = rB * m.w;
                EllPair nA = ell_load(sm, rbA), nB = ell_load(sm, rbB);
                for (int t = 0; t < tw; t += 2) {
                    const EllPair eA = nA, eB = nB;
                    if (t + 2 < tw) { nA = ell_load(sm, rbA + t + 2); nB = ell_load(sm, rbB + t + 2); }
                    const f32x4 dA0 = *(const f32x4*)(cur + eA.s0 * PIECE + swz16(eA.s0, chunk) * 16);
                    const f32x4 dA1 = *(const f32x4*)(cur + eA.s1 * PIECE + swz16(eA.s1, chunk) * 16);
                    const f32x4 dB0 = *(const f32x4*)(cur + eB.s0 * PIECE + swz16(eB.s0, chunk) * 16);
                    const f32x4 dB1 = *(const f32x4*)(cur + eB.s1 * PIECE + swz16(eB.s1, chunk) * 16);
                    __builtin_amdgcn_sched_barrier(0);
                    G[1][0] += eA.v[0] * dA0; G[2][0] += eA.v[1] * dA0; G[1][0] += eA.v[2] * dA1; G[2][0] += eA.v[3] * dA1;
                    G[1][1] += eB.v[0] * dB0; G[2][1] += eB.v[1] * dB0; G[1][1] += eB.v[2] * dB1; G[2][1] += eB.v[3] * dB1;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (dx) {
                f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = accA;
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        accA = __builtin_amdgcn_mfma_f32_16x16x4f32(G[g][0][s], Bt[g][s], accA, 0, 0, 0);
                        accB = __builtin_amdgcn_mfma_f32_16x16x4f32(G[g][1][s], Bt[g][s], accB, 0, 0, 0);
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (prA < m.rows) dx[tbase + (prA * BK_NS + r) * 16] = accA[r] * act_grad_from_output(act, aA[r]);
                    if (prB < m.rows) dx[tbase + (prB * BK_NS + r) * 16] = accB[r] * act_grad_from_output(act, aB[r]);
                }
            }
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                *(f32x4*)(patch + p * T16_STRIDE + 4 * g4) = G[g][0];
                *(f32x4*)(patch + 16 * T16_STRIDE + p * T16_STRIDE + 4 * g4) = G[g][1];
                wave_lds_sync();
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int pt = 4 * g4 + s;
                    dWacc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(aA[s], patch[pt * T16_STRIDE + p], dWacc[g], 0, 0, 0);
                    dWacc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(aB[s], patch[16 * T16_STRIDE + pt * T16_STRIDE + p], dWacc[g], 0, 0, 0);
                }
                wave_lds_sync();
            }
        }
    }
    wait_all_and_barrier();
    float* red = (float*)sm.buf(0);                             // 8 waves * 768 floats
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * 768 + (4 * g4 + r) * 48 + g * 16 + p] = dWacc[g][r];
    __syncthreads();
    float* outp = partial + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 768;
    for (int i = threadIdx.x; i < 768; i += BK_THREADS) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < BK_WAVES; ++w) s += red[w * 768 + i];
        outp[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// backward of the first layer: c_dz = C, c_aux = 1, dW only:  dW_g[0][c] += sum_p x[p] * G_g[p][c]
// thread = (row, trajectory, channel quad); per-thread accumulators reduced once at the end.
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(SP_THREADS, 4) void bwd_c1_kernel(PlanDev P, const float* __restrict__ DZ,
                                                               const float* __restrict__ aux,
                                                               float* __restrict__ partial, int n_rows, int n_cols,
                                                               int n_slabs) {
    // 16 waves, thread = (row, 16-byte chunk) like the dual SpMM: BK_R * CPP <= SP_ITEMS * SP_THREADS items per slab
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PIECE = BK_NS * C * 4, CPP = PIECE / 16, CQ = C / 4;
    const Smem sm = carve(smem, PIECE);
    uint8_t* tws = (uint8_t*)(smem + smem_bytes(PIECE));
    const int tid = threadIdx.x;
    f32x4 acc[3];
    acc[0] = acc[1] = acc[2] = f32x4{0.f, 0.f, 0.f, 0.f};
    int b, b_end, b_stride;
    block_range(P.n_blocks, b, b_end, b_stride);
    SCN_SLAB_RANGE();
    if (slab0 < slab1)
    for (; b < b_end; b += b_stride) {
        wait_all_and_barrier();
        BlockMeta m;
        {
            m.row0 = P.blk_row0[b];
            m.rows = P.blk_rows[b];
            const int sp0 = P.src_ptr[b];
            m.nsrc = P.src_ptr[b + 1] - sp0;
            m.w = P.width[b];
            const int ep = P.ell_ptr[b];
            for (int i = tid; i < m.nsrc; i += SP_THREADS) sm.srcrows[i] = P.src_rows[sp0 + i];
            for (int i = tid; i < m.w * m.rows; i += SP_THREADS) {
                sm.slot[i] = P.ell_slot[ep + i];
                sm.v[i] = P.ell_v[ep + i];
            }
            if (tid < BK_R) sm.self[tid] = P.self_slot[(size_t)b * BK_R + tid];
            if (tid < BK_WAVES) tws[tid] = P.tile_w[b * BK_WAVES + tid];
        }
        __syncthreads();
        dma_stage_sp((const char*)DZ + (size_t)slab0 * n_cols * PIECE, sm.buf(0), sm, m.nsrc, PIECE, CPP);
        const int total = m.rows * CPP;                           // (row, chunk) pairs; chunk = n*CQ + cq
        float xn[SP_ITEMS];                                       // x of the next slab (fetched one slab ahead)
        {
            const float* xs = aux + ((size_t)slab0 * n_rows + m.row0) * BK_NS;
#pragma unroll
            for (int k = 0; k < SP_ITEMS; ++k) {
                const int idx = tid + k * SP_THREADS;
                xn[k] = idx < total ? xs[(idx / CPP) * BK_NS + (idx % CPP) / CQ] : 0.f;
            }
        }
        for (int slab = slab0; slab < slab1; ++slab) {
            const char* cur = sm.buf((slab - slab0) & 1);
            wait_vm_and_barrier();
            float xv[SP_ITEMS];
#pragma unroll
            for (int k = 0; k < SP_ITEMS; ++k) xv[k] = xn[k];
            if (slab + 1 < slab1) {
                const float* xs = aux + ((size_t)(slab + 1) * n_rows + m.row0) * BK_NS;
#pragma unroll
                for (int k = 0; k < SP_ITEMS; ++k) {
                    const int idx = tid + k * SP_THREADS;
                    xn[k] = idx < total ? xs[(idx / CPP) * BK_NS + (idx % CPP) / CQ] : 0.f;
                }
                dma_stage_sp((const char*)DZ + (size_t)(slab + 1) * n_cols * PIECE, sm.buf((slab + 1 - slab0) & 1), sm, m.nsrc,
                             PIECE, CPP);
            }
#pragma unroll
            for (int k = 0; k < SP_ITEMS; ++k) {
                const int idx = tid + k * SP_THREADS;
                if (idx < total) {
                    const int r = idx / CPP, ch = idx - r * CPP;
                    const int tw = __builtin_amdgcn_readfirstlane(tws[r >> 3]);   // CPP in {16, 32}: a wave spans <= 4 rows of one group
                    f32x4 gl = {0.f, 0.f, 0.f, 0.f}, gu = gl;
                    const char* cb = cur + ch * 16;
                    const f32x4 gsv = *(const f32x4*)(cb + sm.self[r] * PIECE);
                    const int rb = r * m.w;
#pragma unroll 2
                    for (int t = 0; t < tw; t += 2) {
                        const EllPair e = ell_load(sm, rb + t);
                        const f32x4 d0 = *(const f32x4*)(cb + e.s0 * PIECE);
                        const f32x4 d1 = *(const f32x4*)(cb + e.s1 * PIECE);
                        gl += e.v[0] * d0;
                        gu += e.v[1] * d0;
                        gl += e.v[2] * d1;
                        gu += e.v[3] * d1;
                    }
                    acc[0] += xv[k] * gsv;
                    acc[1] += xv[k] * gl;
                    acc[2] += xv[k] * gu;
                }
            }
        }
    }
    // threads with equal cq = tid % CQ hold the same channels: reduce over them in a fixed order
    wait_all_and_barrier();
    f32x4* red = (f32x4*)sm.buf(0);                               // [3][SP_THREADS]
#pragma unroll
    for (int g = 0; g < 3; ++g) red[g * SP_THREADS + tid] = acc[g];
    __syncthreads();
    if (tid < 3 * CQ) {
        const int g = tid / CQ, cq = tid - g * CQ;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int t = cq; t < SP_THREADS; t += CQ) s += red[g * SP_THREADS + t];
        float* outp = partial + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (3 * C) + g * C + cq * 4;
        outp[0] = s[0]; outp[1] = s[1]; outp[2] = s[2]; outp[3] = s[3];
    }
}

// ------------------------------------------------------------------------------------------------
// First-layer weight gradient with the shift moved to the cheap side (scn_conv_dw_first):
//   dW_g[0][c] = sum_p (S_g x)[p] * dz[p][c]     instead of     sum_p x[p] * (S_g^T dz)[p][c]
// gather3_c1_kernel computes Y[p] = (x, S_lo x, S_up x)[p] from the 16-byte input pieces (the gather of fwd_c1, 12 bytes per
// point out); dw_first_stream_kernel then reads dz exactly once, coalesced, with no LDS gather at all -- an HBM stream.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BK_THREADS, 2) void gather3_c1_kernel(PlanDev P, const float* __restrict__ X,
                                                                   float* __restrict__ Y, int n_rows, int n_cols,
                                                                   int n_slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PIECE = 16;
    const Smem sm = carve(smem, PIECE);
    const int tid = threadIdx.x;
    int b, b_end, b_stride;
    block_range(P.n_blocks, b, b_end, b_stride);
    SCN_SLAB_RANGE();
    if (slab0 >= slab1) return;
    for (; b < b_end; b += b_stride) {
        wait_all_and_barrier();
        const BlockMeta m = load_block(P, b, sm);
        __syncthreads();
        dma_stage<PIECE, 0>((const char*)X + (size_t)slab0 * n_cols * PIECE, sm.buf(0), sm, m.nsrc);
        for (int slab = slab0; slab < slab1; ++slab) {
            const float* st = (const float*)sm.buf((slab - slab0) & 1);
            wait_vm_and_barrier();
            if (slab + 1 < slab1)
                dma_stage<PIECE, 0>((const char*)X + (size_t)(slab + 1) * n_cols * PIECE, sm.buf((slab + 1 - slab0) & 1), sm,
                                    m.nsrc);
            if (tid < m.rows * BK_NS) {
                const int rt = tid >> 2, n = tid & 3;
                float zs = st[sm.self[rt] * 4 + n], zl = 0.f, zu = 0.f;
                const int tw = P.tile_w[b * BK_WAVES + (rt >> 3)];
                const int rb = rt * m.w;
                for (int t = 0; t < tw; t += 2) {
                    const EllPair e = ell_load(sm, rb + t);
                    const float d0 = st[e.s0 * 4 + n], d1 = st[e.s1 * 4 + n];
                    zl = fmaf(e.v[0], d0, zl);
                    zu = fmaf(e.v[1], d0, zu);
                    zl = fmaf(e.v[2], d1, zl);
                    zu = fmaf(e.v[3], d1, zu);
                }
                *(f32x4*)(Y + (((size_t)slab * n_rows + m.row0) * BK_NS + tid) * Y_STRIDE) = f32x4{zs, zl, zu, 0.f};
            }
        }
    }
}

constexpr int DWS_THREADS = 256, DWS_BLOCKS = 1024;
template <int C>
__global__ __launch_bounds__(DWS_THREADS) void dw_first_stream_kernel(const float* __restrict__ Y,
                                                                      const float* __restrict__ DZ,
                                                                      float* __restrict__ partial, int64_t n_points) {
    constexpr int CQ = C / 4;                                     // channel quads per point; DWS_THREADS % CQ == 0
    __shared__ f32x4 red[3 * DWS_THREADS];
    const int tid = threadIdx.x;
    const int64_t total = n_points * CQ, stride = (int64_t)gridDim.x * DWS_THREADS;
    f32x4 acc[3];
    acc[0] = acc[1] = acc[2] = f32x4{0.f, 0.f, 0.f, 0.f};
    int64_t i = (int64_t)blockIdx.x * DWS_THREADS + tid;
    for (; i + 3 * stride < total; i += 4 * stride) {            // four independent 16-byte loads in flight per thread
        f32x4 d[4];
        float y[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = i + u * stride;
            d[u] = *(const f32x4*)(DZ + j * 4);
            const f32x4 yq = *(const f32x4*)(Y + (j / CQ) * Y_STRIDE);
            y[u][0] = yq[0]; y[u][1] = yq[1]; y[u][2] = yq[2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] += y[u][0] * d[u];
            acc[1] += y[u][1] * d[u];
            acc[2] += y[u][2] * d[u];
        }
    }
    for (; i < total; i += stride) {
        const f32x4 d = *(const f32x4*)(DZ + i * 4);
        const float* yp = Y + (i / CQ) * Y_STRIDE;
        acc[0] += yp[0] * d;
        acc[1] += yp[1] * d;
        acc[2] += yp[2] * d;
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) red[g * DWS_THREADS + tid] = acc[g];
    __syncthreads();
    if (tid < 3 * CQ) {                                           // threads with equal tid % CQ hold the same channels
        const int g = tid / CQ, cq = tid - g * CQ;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int t = cq; t < DWS_THREADS; t += CQ) sum += red[g * DWS_THREADS + t];
        float* outp = partial + (size_t)blockIdx.x * (3 * C) + g * C + cq * 4;
        outp[0] = sum[0]; outp[1] = sum[1]; outp[2] = sum[2]; outp[3] = sum[3];
    }
}

// Zero-skipping variants of the two streaming pieces: one work item = (listed block, one of its slabs) = a contiguous run of
// rows*ns points.  dw_first_list_kernel accumulates like dw_first_stream_kernel over the listed items only (everything else
// of dz is exactly zero); clear_list_kernel writes zeros over the listed items of a [S][rows][ns][c] tensor, which returns
// a buffer of the zero-skipping mode to its all-zero state.
template <int C>
__global__ __launch_bounds__(DWS_THREADS) void dw_first_list_kernel(PlanDev P, WorkList wl, const float* __restrict__ Y,
                                                                    const float* __restrict__ DZ,
                                                                    float* __restrict__ partial, int n_rows) {
    constexpr int CQ = C / 4;
    __shared__ f32x4 red[3 * DWS_THREADS];
    const int tid = threadIdx.x;
    f32x4 acc[3];
    acc[0] = acc[1] = acc[2] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int u = blockIdx.x; u < wl.n_work; u += gridDim.x) {
        const int b = wl.block[u];
        const int row0 = P.blk_row0[b], rows = P.blk_rows[b];
        const int count = rows * BK_NS * CQ;                      // float4 items of the block's rows in one slab
        for (int k = wl.ptr[u]; k < wl.ptr[u + 1]; ++k) {
            const size_t p0 = ((size_t)wl.slab[k] * n_rows + row0) * BK_NS;   // first point
            for (int i = tid; i < count; i += DWS_THREADS) {     // i % CQ == tid % CQ: DWS_THREADS % CQ == 0
                const f32x4 d = *(const f32x4*)(DZ + (p0 * CQ + i) * 4);
                const float* yp = Y + (p0 + i / CQ) * Y_STRIDE;
                acc[0] += yp[0] * d;
                acc[1] += yp[1] * d;
                acc[2] += yp[2] * d;
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) red[g * DWS_THREADS + tid] = acc[g];
    __syncthreads();
    if (tid < 3 * CQ) {
        const int g = tid / CQ, cq = tid - g * CQ;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int t = cq; t < DWS_THREADS; t += CQ) sum += red[g * DWS_THREADS + t];
        float* outp = partial + (size_t)blockIdx.x * (3 * C) + g * C + cq * 4;
        outp[0] = sum[0]; outp[1] = sum[1]; outp[2] = sum[2]; outp[3] = sum[3];
    }
}

__global__ __launch_bounds__(256) void clear_list_kernel(PlanDev P, WorkList wl, float* __restrict__ T, int n_rows,
                                                         int row_floats) {
    for (int u = blockIdx.x; u < wl.n_work; u += gridDim.x) {
        const int b = wl.block[u];
        const int row0 = P.blk_row0[b], count = P.blk_rows[b] * row_floats / 4;
        for (int k = wl.ptr[u]; k < wl.ptr[u + 1]; ++k) {
            f32x4* t = (f32x4*)(T + ((size_t)wl.slab[k] * n_rows + row0) * row_floats);
            for (int i = threadIdx.x; i < count; i += 256) t[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// dW_slot[0][cc] += sum over the stream kernel's partials: one wave per output element, lane l adds partials l, l+64, ...
// and the 64 lane sums are folded in a fixed butterfly order (deterministic).
__global__ __launch_bounds__(64) void dw_first_reduce_kernel(const float* __restrict__ partial, int n_partials, int c,
                                                             float* __restrict__ dW0, float* __restrict__ dW1,
                                                             float* __restrict__ dW2) {
    const int o = blockIdx.x, lane = threadIdx.x;                 // o = slot * c + cc
    float s = 0.f;
    for (int b = lane; b < n_partials; b += 64) s += partial[(size_t)b * (3 * c) + o];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) {
        const int slot = o / c, cc = o - slot * c;
        float* d = slot == 0 ? dW0 : (slot == 1 ? dW1 : dW2);
        d[cc] += s;
    }
}

// dW_slot[i] += sum over partials (fixed order).  layout: partial[b][ca*3c + slot*c + cc]
__global__ void blocked_dw_reduce(const float* __restrict__ partial, int n_partials, int c_aux, int c,
                                  float* __restrict__ dW0, float* __restrict__ dW1, float* __restrict__ dW2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = c_aux * 3 * c;
    if (i >= per) return;
    float s = 0.f;
    for (int b = 0; b < n_partials; ++b) s += partial[(size_t)b * per + i];
    const int ca = i / (3 * c), rem = i - ca * 3 * c, slot = rem / c, cc = rem - slot * c;
    float* d = slot == 0 ? dW0 : (slot == 1 ? dW1 : dW2);
    if (d) d[ca * c + cc] += s;
}

// ------------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------------
// dynamic LDS above 64 KB has to be opted into per kernel
#define SCN_ENSURE_LDS(kernel, bytes)                                                                          \
    do {                                                                                                       \
        if ((bytes) > 160 * 1024) return SCN_ERR_UNSUPPORTED;                                                  \
        if ((bytes) > 64 * 1024)                                                                               \
            SCN_HIP_TRY(hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                            (int)(bytes)));                                                    \
    } while (0)

static bool scone_shape(const scn_conv_s* c) {
    return c->plan.built && c->n_groups == 1 && c->g[0].identity == 1 && c->g[0].n_vals == 2;
}

// persistent grid: workgroups per CU by LDS footprint, blocks strided XCD-contiguously; small operators split slabs
// Cost-balanced STATIC assignment of blocks to workgroups for a launch with gx workgroups per slab range: workgroup j of XCD
// x visits positions b0 + j, b0 + j + stride, ... of its XCD's range (block_range); within every round of `stride` positions
// the most expensive blocks go to the workgroups that have the least so far.  Blocks of a round still run at the same time
// (halo rows stay shared in L2) and the assignment is fixed, so the weight gradient stays bitwise reproducible; the busiest
// workgroup had 2.3 % (forward) / 3.4 % (backward) more than the mean with the plain stride.
static const int32_t* balanced_assignment(const scn_conv_s* c, int gx) {   // tables are built with the plan, for every grid launch_grid can pick
    static const bool off = getenv("SCN_NO_BALANCE") != nullptr;            // A/B switch: plain strided assignment
    if (off) return nullptr;
    auto it = c->plan.assign_by_grid.find(gx);
    return it != c->plan.assign_by_grid.end() ? it->second : nullptr;
}

static int build_assignment(scn_conv_s* c, int gx) {
    BlockPlan& B = c->plan;
    const int nb = B.dev.n_blocks, stride = gx / 8;
    if (stride <= 0 || nb <= gx || (int)B.h_cost.size() != nb || B.assign_by_grid.count(gx)) return SCN_OK;
    std::vector<int32_t> assign(nb);
    std::vector<double> load(stride);
    std::vector<int> wg(stride), blk(stride);
    for (int x = 0; x < 8; ++x) {
        const int b0 = (int)((int64_t)nb * x / 8), last = (int)((int64_t)nb * (x + 1) / 8);
        std::fill(load.begin(), load.end(), 0.0);
        for (int r = b0; r < last; r += stride) {
            const int cnt = std::min(stride, last - r);
            for (int i = 0; i < cnt; ++i) { wg[i] = i; blk[i] = r + i; }
            std::stable_sort(wg.begin(), wg.begin() + cnt, [&](int a, int b) { return load[a] < load[b]; });
            std::stable_sort(blk.begin(), blk.begin() + cnt, [&](int a, int b) { return B.h_cost[a] > B.h_cost[b]; });
            for (int i = 0; i < cnt; ++i) {
                assign[r + wg[i]] = blk[i];
                load[wg[i]] += B.h_cost[blk[i]];
            }
        }
    }
    const int32_t* dev = nullptr;
    const int st = upload(c, assign, &dev);
    if (st == SCN_OK) B.assign_by_grid[gx] = dev;
    return st;
}

int build_assignments(scn_conv_s* c) {                      // every grid.x launch_grid can produce for this plan
    const int nb = c->plan.dev.n_blocks;
    for (int cap : {256, 512, 768})
        for (int gy = 1; gy <= 32; gy *= 2) {
            const int gx = std::max(8, std::min(cap / gy, ((nb + 7) / 8) * 8) / 8 * 8);
            const int st = build_assignment(c, gx);
            if (st != SCN_OK) return st;
        }
    return SCN_OK;
}

static void launch_grid(const scn_conv_s* c, int n_slabs, size_t lds, dim3& grid, int max_per_cu = 2) {
    const int nb = c->plan.dev.n_blocks;
    const int per_cu = (max_per_cu >= 3 && lds <= 160 * 1024 / max_per_cu) ? max_per_cu : (lds <= 80 * 1024 ? 2 : 1);
    const int cap = 256 * per_cu;
    // gx workgroups stride over the blocks (a multiple of 8: one share per XCD), gy split the slabs.  Pick the split whose
    // busiest workgroup has the least (blocks x slabs) to do: at |E| = 50k (830 blocks) 256 x 1 leaves a 4-vs-3 block tail,
    // 64 x 4 is even; at |E| = 1M the answer stays 256 x 1.
    const int nb_xcd = (nb + 7) / 8;
    long best = -1;
    int bx = 8, by = 1;
    for (int gy = 1; gy <= 32 && gy <= std::max(1, n_slabs); gy *= 2) {
        int gx = std::max(8, std::min(cap / gy, ((nb + 7) / 8) * 8) / 8 * 8);
        // a block visit costs its slabs plus about half a slab of prologue (ELL tile, DMA offsets)
        const long work = (long)((nb_xcd + gx / 8 - 1) / (gx / 8)) * (2 * ((n_slabs + gy - 1) / gy) + 1);
        const long cost = work * 64 + gy;                                                    // ties: fewer slab splits
        if (best < 0 || cost < best) { best = cost; bx = gx; by = gy; }
    }
    grid = dim3(bx, by);
}

bool blocked_forward_supported(const scn_conv_s* c, int ns, const int32_t* c_in, int c_out) {
    if (!scone_shape(c) || ns != BK_NS) return false;
    const int ci = c_in[0];
    return (ci == 32 && c_out == 32) || (ci == 16 && c_out == 16) || (ci == 1 && (c_out == 16 || c_out == 32));
}

int blocked_forward(scn_conv_s* c, int n_slabs, int ns, const float* const* src, const int32_t* c_in,
                    const float* const* W, int c_out, int act, float* out, float* y_out, const WorkList* wlp,
                    hipStream_t st) {
    PlanDev P = c->plan.dev;
    const WorkList wl = wlp ? *wlp : WorkList{0, nullptr, nullptr, nullptr};
    dim3 grid;
    const int ci = c_in[0];
    const int nr = c->n_rows, nc = c->g[0].n_cols;
    if (ci == 32) {
        const size_t lds = smem_bytes(512);
        SCN_ENSURE_LDS(fwd_c32_kernel<SCN_ACT_NONE>, lds);
        SCN_ENSURE_LDS(fwd_c32_kernel<SCN_ACT_TANH>, lds);
        SCN_ENSURE_LDS(fwd_c32_kernel<SCN_ACT_RELU>, lds);
        SCN_ENSURE_LDS(fwd_c32_kernel<SCN_ACT_LEAKY_RELU>, lds);
        launch_grid(c, n_slabs, lds, grid);
        P.assign = balanced_assignment(c, grid.x);
#define SCN_LAUNCH_FWD32(A)                                                                                       \
    hipLaunchKernelGGL(fwd_c32_kernel<A>, grid, dim3(BK_THREADS), lds, st, P, src[0], W[0], W[1], W[2], out, nr, nc, \
                       n_slabs)
        static const bool f32_mfma = getenv("SCN_F32_MFMA") != nullptr;   // A/B switch: fp32-MFMA variant
        if (wl.block) grid.y = 1;                                          // a work list carries its own slab lists
        if (!f32_mfma || wl.block) {                                       // default: the 16-wave bf16x3 kernel
            const size_t lds16 = smem_bytes_c32(W16_WFRAG_BYTES + 16);
#define SCN_LAUNCH_FWD32W(A)                                                                                      \
    do {                                                                                                          \
        SCN_ENSURE_LDS(fwd_c32_w16_kernel<A>, lds16);                                                             \
        hipLaunchKernelGGL(fwd_c32_w16_kernel<A>, grid, dim3(W16_THREADS), lds16, st, P, src[0], (const float*)nullptr,   \
                           W[0], W[1], W[2], out, nr, nc, n_slabs, wl);                                           \
    } while (0)
            switch (act) {
                case SCN_ACT_TANH: SCN_LAUNCH_FWD32W(SCN_ACT_TANH); break;
                case SCN_ACT_RELU: SCN_LAUNCH_FWD32W(SCN_ACT_RELU); break;
                case SCN_ACT_LEAKY_RELU: SCN_LAUNCH_FWD32W(SCN_ACT_LEAKY_RELU); break;
                default: SCN_LAUNCH_FWD32W(SCN_ACT_NONE); break;
            }
            SCN_LAUNCH_CHECK();
            return SCN_OK;
        }
        switch (act) {
            case SCN_ACT_TANH: SCN_LAUNCH_FWD32(SCN_ACT_TANH); break;
            case SCN_ACT_RELU: SCN_LAUNCH_FWD32(SCN_ACT_RELU); break;
            case SCN_ACT_LEAKY_RELU: SCN_LAUNCH_FWD32(SCN_ACT_LEAKY_RELU); break;
            default: SCN_LAUNCH_FWD32(SCN_ACT_NONE); break;
        }
    } else if (ci == 16) {
        static const bool f32_mfma16 = getenv("SCN_F32_MFMA") != nullptr;  // A/B switch: the 8-wave fp32-MFMA variant
        if (!f32_mfma16) {                                                 // default: 16 waves, bf16x3, two slabs per visit
            const size_t lds16 = smem_bytes_c32(16);
            launch_grid(c, n_slabs, lds16, grid);
            P.assign = balanced_assignment(c, grid.x);
            if (wl.block) grid.y = 1;
#define SCN_LAUNCH_FWD16W(A)                                                                                      \
    do {                                                                                                          \
        SCN_ENSURE_LDS(fwd_c16_w16_kernel<A>, lds16);                                                             \
        hipLaunchKernelGGL(fwd_c16_w16_kernel<A>, grid, dim3(W16_THREADS), lds16, st, P, src[0], W[0], W[1], W[2], out, nr,  \
                           nc, n_slabs, wl);                                                                      \
    } while (0)
            switch (act) {
                case SCN_ACT_TANH: SCN_LAUNCH_FWD16W(SCN_ACT_TANH); break;
                case SCN_ACT_RELU: SCN_LAUNCH_FWD16W(SCN_ACT_RELU); break;
                case SCN_ACT_LEAKY_RELU: SCN_LAUNCH_FWD16W(SCN_ACT_LEAKY_RELU); break;
                default: SCN_LAUNCH_FWD16W(SCN_ACT_NONE); break;
            }
            SCN_LAUNCH_CHECK();
            return SCN_OK;
        }
        const size_t lds = smem_bytes(256);
        SCN_ENSURE_LDS(fwd_c16_kernel, lds);
        launch_grid(c, n_slabs, lds, grid);
        P.assign = balanced_assignment(c, grid.x);
        if (wl.block) grid.y = 1;
        hipLaunchKernelGGL(fwd_c16_kernel, grid, dim3(BK_THREADS), lds, st, P, src[0], W[0], W[1], W[2], out, nr, nc,
                           n_slabs, act, wl);
    } else {
        const size_t lds = smem_bytes(16, 2 * BK_R * BK_NS * 12);
        launch_grid(c, n_slabs, lds, grid, 3);      // 8-wave workgroups at 64 VGPRs and 23 KB of LDS: three per CU (4.45 -> 3.9 ms; four: 4.8)
        P.assign = balanced_assignment(c, grid.x);
        if (wl.block) grid.y = 1;
        if (c_out == 32)
            hipLaunchKernelGGL(fwd_c1_kernel<32>, grid, dim3(BK_THREADS), lds, st, P, src[0], W[0], W[1], W[2], out, y_out, nr,
                               nc, n_slabs, act, wl);
        else
            hipLaunchKernelGGL(fwd_c1_kernel<16>, grid, dim3(BK_THREADS), lds, st, P, src[0], W[0], W[1], W[2], out, y_out, nr,
                               nc, n_slabs, act, wl);
    }
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

bool blocked_backward_supported(const scn_conv_s* c, int ns, const int32_t* c_dz, int c_aux, bool has_dx) {
    if (!scone_shape(c) || ns != BK_NS) return false;
    const int cd = c_dz[0];
    if ((cd == 32 && c_aux == 32) || (cd == 16 && c_aux == 16)) return true;
    return (cd == 16 || cd == 32) && c_aux == 1 && !has_dx;
}

static size_t bwd_lds(int cd, int c_aux) {
    if (c_aux == 32) return smem_bytes_c32(B32_WFRAG_BYTES + 2048);   // weight + selection fragments (>= the fp32 variant's smem_bytes(512) + patches)
    if (c_aux == 16) return std::max(smem_bytes_c32(B32_WFRAG_BYTES + 2048), smem_bytes(256, BK_WAVES * 2 * 16 * T16_STRIDE * 4));
    return smem_bytes(BK_NS * cd * 4, 16);
}

size_t blocked_backward_workspace(const scn_conv_s* c, int n_slabs, int ns, const int32_t* c_dz, int c_aux) {
    if (!blocked_backward_supported(c, ns, c_dz, c_aux, false) && !blocked_backward_supported(c, ns, c_dz, c_aux, true))
        return 0;
    dim3 grid;
    launch_grid(c, n_slabs, bwd_lds(c_dz[0], c_aux), grid);
    return (size_t)grid.x * grid.y * c_aux * 3 * c_dz[0] * sizeof(float);
}

int blocked_backward(scn_conv_s* c, int n_slabs, int ns, const float* const* dz, const int32_t* c_dz,
                     const float* const* W, const float* aux, int c_aux, int act, float* dx,
                     float* const* dW, void* ws, size_t ws_bytes, const WorkList* wlp, hipStream_t st) {
    PlanDev P = c->plan.dev;
    const WorkList wl = wlp ? *wlp : WorkList{0, nullptr, nullptr, nullptr};
    dim3 grid;
    const int cd = c_dz[0];
    const int nr = c->n_rows, nc = c->g[0].n_cols;
    float* partial = (float*)ws;
    const size_t lds = bwd_lds(cd, c_aux);
    launch_grid(c, n_slabs, lds, grid);
    P.assign = balanced_assignment(c, grid.x);
    if (c_aux == 32) {
        static const bool f32_mfma = getenv("SCN_F32_MFMA") != nullptr;   // A/B switch: fp32-MFMA variant
        if (wl.block) grid.y = 1;
        if (f32_mfma && !wl.block) {
            SCN_ENSURE_LDS(bwd_c32_kernel, lds);
            hipLaunchKernelGGL(bwd_c32_kernel, grid, dim3(BK_THREADS), lds, st, P, dz[0], W[0], W[1], W[2], aux, dx, partial,
                               nr, nc, n_slabs, act);
        } else {
#define SCN_LAUNCH_BWD32(A)                                                                                       \
    do {                                                                                                          \
        SCN_ENSURE_LDS(bwd_c32_bf16_kernel<A>, lds);                                                              \
        hipLaunchKernelGGL(bwd_c32_bf16_kernel<A>, grid, dim3(BK_THREADS), lds, st, P, dz[0], (const float*)nullptr,     \
                           W[0], W[1], W[2], aux, dx, partial, nr, nc, n_slabs, wl);                              \
    } while (0)
            switch (act) {
                case SCN_ACT_TANH: SCN_LAUNCH_BWD32(SCN_ACT_TANH); break;
                case SCN_ACT_RELU: SCN_LAUNCH_BWD32(SCN_ACT_RELU); break;
                case SCN_ACT_LEAKY_RELU: SCN_LAUNCH_BWD32(SCN_ACT_LEAKY_RELU); break;
                default: SCN_LAUNCH_BWD32(SCN_ACT_NONE); break;
            }
        }
    } else if (c_aux == 16) {
        static const bool f32_mfma16 = getenv("SCN_F32_MFMA") != nullptr;   // A/B switch: the fp32-MFMA variant
        if (wl.block) grid.y = 1;
        if (f32_mfma16) {
            SCN_ENSURE_LDS(bwd_c16_kernel, lds);
            hipLaunchKernelGGL(bwd_c16_kernel, grid, dim3(BK_THREADS), lds, st, P, dz[0], W[0], W[1], W[2], aux, dx, partial,
                               nr, nc, n_slabs, act, wl);
        } else {                                                            // default: the C=32 bf16x3 kernel on slab pairs
#define SCN_LAUNCH_BWD16P(A)                                                                                      \
    do {                                                                                                          \
        SCN_ENSURE_LDS((bwd_c32_bf16_kernel<A, false, true>), lds);                                               \
        hipLaunchKernelGGL((bwd_c32_bf16_kernel<A, false, true>), grid, dim3(BK_THREADS), lds, st, P, dz[0],      \
                           (const float*)nullptr, W[0], W[1], W[2], aux, dx, partial, nr, nc, n_slabs, wl);       \
    } while (0)
            switch (act) {
                case SCN_ACT_TANH: SCN_LAUNCH_BWD16P(SCN_ACT_TANH); break;
                case SCN_ACT_RELU: SCN_LAUNCH_BWD16P(SCN_ACT_RELU); break;
                case SCN_ACT_LEAKY_RELU: SCN_LAUNCH_BWD16P(SCN_ACT_LEAKY_RELU); break;
                default: SCN_LAUNCH_BWD16P(SCN_ACT_NONE); break;
            }
        }
    } else if (wl.block) {
        return SCN_ERR_UNSUPPORTED;                                  // first layer with a list: scn_conv_dw_first
    } else if (cd == 32) {
        SCN_ENSURE_LDS(bwd_c1_kernel<32>, lds);
        hipLaunchKernelGGL(bwd_c1_kernel<32>, grid, dim3(SP_THREADS), lds, st, P, dz[0], aux, partial, nr, nc, n_slabs);
    } else {
        SCN_ENSURE_LDS(bwd_c1_kernel<16>, lds);
        hipLaunchKernelGGL(bwd_c1_kernel<16>, grid, dim3(SP_THREADS), lds, st, P, dz[0], aux, partial, nr, nc, n_slabs);
    }
    SCN_LAUNCH_CHECK();
    const int per = c_aux * 3 * cd;
    hipLaunchKernelGGL(blocked_dw_reduce, dim3((per + 255) / 256), dim3(256), 0, st, partial, (int)(grid.x * grid.y), c_aux,
                       cd, dW[0], dW[1], dW[2]);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// Backward of the layer that follows the first one, fused with the first layer's weight gradient (bwd_c32_bf16_kernel<.., FIRST>).
// Workspace: [this layer's dW partials][dW_first partials: 96 floats per workgroup].
bool blocked_backward_first_supported(const scn_conv_s* c, int ns, int ch) {
    return scone_shape(c) && ns == BK_NS && (ch == 32 || ch == 16);          // 16: the slab-pair form
}

size_t blocked_backward_first_workspace(const scn_conv_s* c, int n_slabs, int ns, int ch) {
    if (!blocked_backward_first_supported(c, ns, ch)) return 0;
    dim3 grid;
    launch_grid(c, n_slabs, bwd_lds(ch, ch), grid);
    return (size_t)grid.x * grid.y * (3 * ch * ch + 96) * sizeof(float);
}

// dW_first[slot][cc] of the slab-pair form: partial [b][slot * 32 + 16 s + cc], the two slabs s folded here (fixed order)
__global__ __launch_bounds__(64) void dw_first_reduce_pair_kernel(const float* __restrict__ partial, int n_partials,
                                                                  float* __restrict__ dW0, float* __restrict__ dW1,
                                                                  float* __restrict__ dW2) {
    const int o = blockIdx.x, lane = threadIdx.x;                 // o = slot * 16 + cc
    const int slot = o >> 4, cc = o & 15;
    float s = 0.f;
    for (int b = lane; b < n_partials; b += 64) s += partial[(size_t)b * 96 + slot * 32 + cc] + partial[(size_t)b * 96 + slot * 32 + 16 + cc];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) {
        float* d = slot == 0 ? dW0 : (slot == 1 ? dW1 : dW2);
        d[cc] += s;
    }
}

int blocked_backward_first(scn_conv_s* c, int n_slabs, const float* dz, const float* const* W, const float* aux, int ch, int act,
                           const float* y, float* const* dW, float* const* dW_first, void* ws, const WorkList* wlp,
                           hipStream_t st) {
    PlanDev P = c->plan.dev;
    const WorkList wl = wlp ? *wlp : WorkList{0, nullptr, nullptr, nullptr};
    dim3 grid;
    const size_t lds = bwd_lds(ch, ch);
    launch_grid(c, n_slabs, lds, grid);
    P.assign = balanced_assignment(c, grid.x);
    if (wl.block) grid.y = 1;
    const int nr = c->n_rows, nc = c->g[0].n_cols;
    const int n_wg = (int)(grid.x * grid.y);
    float* partial = (float*)ws;
    float* partial_first = partial + (size_t)n_wg * 3 * ch * ch;
#define SCN_LAUNCH_BWDF(A, PAIRV)                                                                                 \
    do {                                                                                                          \
        SCN_ENSURE_LDS((bwd_c32_bf16_kernel<A, false, PAIRV, true>), lds);                                        \
        hipLaunchKernelGGL((bwd_c32_bf16_kernel<A, false, PAIRV, true>), grid, dim3(BK_THREADS), lds, st, P, dz, y, W[0], \
                           W[1], W[2], aux, (float*)nullptr, partial, nr, nc, n_slabs, wl, partial_first);        \
    } while (0)
    if (ch == 32) {
        switch (act) {
            case SCN_ACT_TANH: SCN_LAUNCH_BWDF(SCN_ACT_TANH, false); break;
            case SCN_ACT_RELU: SCN_LAUNCH_BWDF(SCN_ACT_RELU, false); break;
            case SCN_ACT_LEAKY_RELU: SCN_LAUNCH_BWDF(SCN_ACT_LEAKY_RELU, false); break;
            default: SCN_LAUNCH_BWDF(SCN_ACT_NONE, false); break;
        }
    } else {
        switch (act) {
            case SCN_ACT_TANH: SCN_LAUNCH_BWDF(SCN_ACT_TANH, true); break;
            case SCN_ACT_RELU: SCN_LAUNCH_BWDF(SCN_ACT_RELU, true); break;
            case SCN_ACT_LEAKY_RELU: SCN_LAUNCH_BWDF(SCN_ACT_LEAKY_RELU, true); break;
            default: SCN_LAUNCH_BWDF(SCN_ACT_NONE, true); break;
        }
    }
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(blocked_dw_reduce, dim3((ch * 3 * ch + 255) / 256), dim3(256), 0, st, partial, n_wg, ch, ch, dW[0], dW[1],
                       dW[2]);
    SCN_LAUNCH_CHECK();
    if (ch == 32)
        hipLaunchKernelGGL(dw_first_reduce_kernel, dim3(96), dim3(64), 0, st, partial_first, n_wg, 32, dW_first[0], dW_first[1],
                           dW_first[2]);
    else
        hipLaunchKernelGGL(dw_first_reduce_pair_kernel, dim3(48), dim3(64), 0, st, partial_first, n_wg, dW_first[0], dW_first[1],
                           dW_first[2]);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// (with a y handed in the operator only supplies the row count and, for work lists, the block table)
bool blocked_dw_first_supported(const scn_conv_s* c, int ns, int cd) {
    return c->plan.built && c->n_groups == 1 && ns == BK_NS && (cd == 16 || cd == 32);
}

static size_t dw_first_y_bytes(const scn_conv_s* c, int n_slabs) {
    const size_t b = (size_t)n_slabs * c->n_rows * BK_NS * Y_STRIDE * sizeof(float);
    return (b + 255) / 256 * 256;
}

size_t blocked_dw_first_workspace(const scn_conv_s* c, int n_slabs, int ns, int cd) {
    if (!blocked_dw_first_supported(c, ns, cd)) return 0;
    return dw_first_y_bytes(c, n_slabs) + (size_t)DWS_BLOCKS * 3 * cd * sizeof(float);
}

// y != nullptr: the shifted input saved by the forward (scn_conv_forward_first); otherwise it is computed here into ws.
// wlp: zero-skipping work list (requires y).
int blocked_dw_first(scn_conv_s* c, int n_slabs, const float* x, const float* y, const float* dz, int cd, float* const* dW,
                     void* ws, const WorkList* wlp, hipStream_t st) {
    const PlanDev& P = c->plan.dev;
    float* partial = (float*)((char*)ws + dw_first_y_bytes(c, n_slabs));
    if (wlp && wlp->block) {
        if (!y) return SCN_ERR_BAD_ARG;
        if (cd == 32)
            hipLaunchKernelGGL(dw_first_list_kernel<32>, dim3(DWS_BLOCKS), dim3(DWS_THREADS), 0, st, P, *wlp, y, dz, partial,
                               c->n_rows);
        else
            hipLaunchKernelGGL(dw_first_list_kernel<16>, dim3(DWS_BLOCKS), dim3(DWS_THREADS), 0, st, P, *wlp, y, dz, partial,
                               c->n_rows);
        SCN_LAUNCH_CHECK();
        hipLaunchKernelGGL(dw_first_reduce_kernel, dim3(3 * cd), dim3(64), 0, st, partial, DWS_BLOCKS, cd, dW[0], dW[1], dW[2]);
        SCN_LAUNCH_CHECK();
        return SCN_OK;
    }
    if (!y) {
        if (!scone_shape(c)) return SCN_ERR_UNSUPPORTED;          // recomputing y needs identity + two value arrays
        float* Y = (float*)ws;
        dim3 grid;
        const size_t lds = smem_bytes(16);
        launch_grid(c, n_slabs, lds, grid);
        hipLaunchKernelGGL(gather3_c1_kernel, grid, dim3(BK_THREADS), lds, st, P, x, Y, c->n_rows, c->g[0].n_cols, n_slabs);
        SCN_LAUNCH_CHECK();
        y = Y;
    }
    const int64_t n_points = (int64_t)n_slabs * c->n_rows * BK_NS;
    if (cd == 32)
        hipLaunchKernelGGL(dw_first_stream_kernel<32>, dim3(DWS_BLOCKS), dim3(DWS_THREADS), 0, st, y, dz, partial, n_points);
    else
        hipLaunchKernelGGL(dw_first_stream_kernel<16>, dim3(DWS_BLOCKS), dim3(DWS_THREADS), 0, st, y, dz, partial, n_points);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(dw_first_reduce_kernel, dim3(3 * cd), dim3(64), 0, st, partial, DWS_BLOCKS, cd, dW[0], dW[1], dW[2]);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

int blocked_clear_list(scn_conv_s* c, int ns, int ch, float* t, const WorkList* wl, hipStream_t st) {
    if (!c->plan.built || ns != BK_NS || !wl || !wl->block || (ns * ch) % 4) return SCN_ERR_UNSUPPORTED;
    if (wl->n_work == 0) return SCN_OK;
    hipLaunchKernelGGL(clear_list_kernel, dim3(std::min(wl->n_work, 2048)), dim3(256), 0, st, c->plan.dev, *wl, t, c->n_rows,
                       ns * ch);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// ---- "power" layers: one operator S (identity + one value array), three terms  X0, S-input's own row, S * input ----
static bool power_shape(const scn_conv_s* c) {
    return c->plan.built && c->n_groups == 1 && c->g[0].identity == 1 && c->g[0].n_vals == 1;
}
bool blocked_power_supported(const scn_conv_s* c, int ns, int ch) { return power_shape(c) && ns == BK_NS && ch == 32; }

size_t blocked_power_backward_workspace(const scn_conv_s* c, int n_slabs, int ns, int ch) {
    if (!blocked_power_supported(c, ns, ch)) return 0;
    dim3 grid;
    launch_grid(c, n_slabs, smem_bytes_c32(B32_WFRAG_BYTES + 2048), grid);
    return (size_t)grid.x * grid.y * 32 * 3 * 32 * sizeof(float);
}

int blocked_power_forward(scn_conv_s* c, int n_slabs, const float* x0, const float* x, const float* const* W, int act,
                          float* out, hipStream_t st) {
    PlanDev P = c->plan.dev;
    const WorkList wl{0, nullptr, nullptr, nullptr};
    dim3 grid;
    const size_t lds16 = smem_bytes_c32(W16_WFRAG_BYTES + 16);
    launch_grid(c, n_slabs, lds16, grid);
    P.assign = balanced_assignment(c, grid.x);
    const int nr = c->n_rows, nc = c->g[0].n_cols;
#define SCN_LAUNCH_FWDP(A)                                                                                        \
    do {                                                                                                          \
        SCN_ENSURE_LDS((fwd_c32_w16_kernel<A, true>), lds16);                                                     \
        hipLaunchKernelGGL((fwd_c32_w16_kernel<A, true>), grid, dim3(W16_THREADS), lds16, st, P, x, x0, W[0], W[1], W[2], \
                           out, nr, nc, n_slabs, wl);                                                             \
    } while (0)
    switch (act) {
        case SCN_ACT_TANH: SCN_LAUNCH_FWDP(SCN_ACT_TANH); break;
        case SCN_ACT_RELU: SCN_LAUNCH_FWDP(SCN_ACT_RELU); break;
        case SCN_ACT_LEAKY_RELU: SCN_LAUNCH_FWDP(SCN_ACT_LEAKY_RELU); break;
        default: SCN_LAUNCH_FWDP(SCN_ACT_NONE); break;
    }
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

int blocked_power_backward(scn_conv_s* c, int n_slabs, const float* dz, const float* g1, const float* const* W,
                           const float* aux, int act, float* dx, float* const* dW, void* ws, hipStream_t st) {
    PlanDev P = c->plan.dev;
    const WorkList wl{0, nullptr, nullptr, nullptr};
    dim3 grid;
    const size_t lds = smem_bytes_c32(B32_WFRAG_BYTES + 2048);
    launch_grid(c, n_slabs, lds, grid);
    P.assign = balanced_assignment(c, grid.x);
    const int nr = c->n_rows, nc = c->g[0].n_cols;
    float* partial = (float*)ws;
#define SCN_LAUNCH_BWDP(A)                                                                                        \
    do {                                                                                                          \
        SCN_ENSURE_LDS((bwd_c32_bf16_kernel<A, true>), lds);                                                      \
        hipLaunchKernelGGL((bwd_c32_bf16_kernel<A, true>), grid, dim3(BK_THREADS), lds, st, P, g1, dz, W[0], W[1], W[2], \
                           aux, dx, partial, nr, nc, n_slabs, wl);                                                \
    } while (0)
    switch (act) {
        case SCN_ACT_TANH: SCN_LAUNCH_BWDP(SCN_ACT_TANH); break;
        case SCN_ACT_RELU: SCN_LAUNCH_BWDP(SCN_ACT_RELU); break;
        case SCN_ACT_LEAKY_RELU: SCN_LAUNCH_BWDP(SCN_ACT_LEAKY_RELU); break;
        default: SCN_LAUNCH_BWDP(SCN_ACT_NONE); break;
    }
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(blocked_dw_reduce, dim3((32 * 3 * 32 + 255) / 256), dim3(256), 0, st, partial, (int)(grid.x * grid.y), 32, 32,
                       dW[0], dW[1], dW[2]);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

bool blocked_spmm_supported(const scn_conv_s* c, int k) {
    return c->plan.built && c->n_groups == 1 && k % 4 == 0 && k >= 4 && k <= 128;
}

int blocked_spmm(scn_conv_s* c, int n_slabs, int k, const float* x, float* ya, float* yb, hipStream_t st) {
    PlanDev P = c->plan.dev;
    dim3 grid;
    const size_t lds = smem_bytes(k * 4, 16);
    launch_grid(c, n_slabs, lds, grid);
    P.assign = balanced_assignment(c, grid.x);
    static const bool no_ring = getenv("SCN_SPMM_TWO_BUFFERS") != nullptr;      // A/B switch: the two-buffer kernel for every K
    if (!no_ring && (k == 128 || k == 64)) {
#define SCN_LAUNCH_RING(D, NI)                                                                                    \
    do {                                                                                                          \
        SCN_ENSURE_LDS((spmm_ring_kernel<D, NI>), lds);                                                           \
        hipLaunchKernelGGL((spmm_ring_kernel<D, NI>), grid, dim3(SP_THREADS), lds, st, P, x, ya, yb, c->n_rows,   \
                           c->g[0].n_cols, n_slabs);                                                              \
    } while (0)
        if (yb) { if (k == 128) SCN_LAUNCH_RING(true, 2); else SCN_LAUNCH_RING(true, 1); }
        else    { if (k == 128) SCN_LAUNCH_RING(false, 2); else SCN_LAUNCH_RING(false, 1); }
        SCN_LAUNCH_CHECK();
        return SCN_OK;
    }
    // narrow operands: fold consecutive slabs into one staged piece (see dma_stage_sp) until it is 128 floats wide
    static const bool no_batch = getenv("SCN_SPMM_NO_BATCH") != nullptr;        // A/B switch
    const int batch = (no_batch || k > 32) ? 1 : std::max(1, std::min(n_slabs, 128 / k));
    const size_t ldsb = smem_bytes(k * 4 * batch, 16);
    if (batch > 1) {
        launch_grid(c, (n_slabs + batch - 1) / batch, ldsb, grid);
        P.assign = balanced_assignment(c, grid.x);
    }
    SCN_ENSURE_LDS(spmm_blocked_kernel<true>, ldsb);
    SCN_ENSURE_LDS(spmm_blocked_kernel<false>, ldsb);
    if (yb)
        hipLaunchKernelGGL(spmm_blocked_kernel<true>, grid, dim3(SP_THREADS), ldsb, st, P, x, ya, yb, c->n_rows, c->g[0].n_cols,
                           n_slabs, k, batch);
    else
        hipLaunchKernelGGL(spmm_blocked_kernel<false>, grid, dim3(SP_THREADS), ldsb, st, P, x, ya, yb, c->n_rows,
                           c->g[0].n_cols, n_slabs, k, batch);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

#include "scn_terms.inc"

}  // namespace scn


// Host-only layout helper (no reference counterpart: the reference's dense operators have no storage order).
// All lanes of a wave walk the ELL rows of their 8-row group to the group's widest row, so a block whose rows are sorted
// by entry count wastes fewer gather iterations (1.25x -> 1.09x of nnz at |E| = 1M).  For a SQUARE pattern whose rows and
// columns share one index space: order[new] = old (rows only move inside the blocks the plan would cut) and
// block_start[new] = 1 where those blocks begin -- pass it to scn_conv_create_blocked so the plan keeps exactly these cuts.
extern "C" int scn_plan_refine_order(int32_t n, const int32_t* rowptr, const int32_t* col, int32_t identity, int32_t* order,
                                     uint8_t* block_start) {
    if (n < 0 || !rowptr || !order || !block_start || (n > 0 && !col && rowptr[n] > 0)) return SCN_ERR_BAD_ARG;
    for (int64_t j = 0; j < rowptr[n]; ++j)
        if (col[j] < 0 || col[j] >= n) return SCN_ERR_BAD_SHAPE;
    std::vector<int32_t> mark(n, -1), cur;
    cur.reserve(scn::BK_SRC + 64);
    int bid = 0;
    for (int r0 = 0; r0 < n; ++bid) {
        int rows = 0, w = 0;
        scn::grow_block(rowptr, col, n, identity != 0, r0, bid, mark, cur, rows, w);
        if (rows == 0) rows = 1;                                  // a row the plan cannot hold: leave it where it is
        for (int i = 0; i < rows; ++i) {
            order[r0 + i] = r0 + i;
            block_start[r0 + i] = i == 0;
        }
        std::stable_sort(order + r0, order + r0 + rows, [&](int32_t a, int32_t b) {
            return rowptr[a + 1] - rowptr[a] > rowptr[b + 1] - rowptr[b];
        });
        r0 += rows;
    }
    return SCN_OK;
}

#ifdef SCN_STAMPS
extern "C" int scn_debug_stamps(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(scn::g_stamps), 64) != hipSuccess) return -3;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(scn::g_stamps), z, 64) != hipSuccess) return -3; }
    return 0;
}
#endif
