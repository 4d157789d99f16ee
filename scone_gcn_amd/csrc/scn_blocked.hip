// LDS-blocked kernels for the scone / ebli layer shape: ONE group, identity + two value arrays on a shared
// pattern (slots: identity, S_lower, S_upper), ns = 4, channel widths 16 or 32 (MFMA) and 1 (first layer).
//
// Execution plan (built once per operator on the host, build_block_plan):
//   the output rows are cut into blocks of <= 32 consecutive rows whose gather sources (<= 96 distinct source
//   rows) are staged once per slab into LDS; the block's CSR slice is re-expressed as a padded ELL tile with
//   LOCAL source slots (uint8) so the inner loop touches LDS only.  Rows are expected in a locality order
//   (Hilbert order of the edge midpoints, scone_gcn_amd/complex.py) so a block's sources are ~2.5x its rows.
//
// Kernel structure (per workgroup of 4 waves, grid-strided over blocks with an XCD-contiguous mapping):
//   for block: load ELL tile + source list -> LDS
//     for slab: stage source pieces (16-B coalesced loads -> XOR-swizzled LDS image) | barrier |
//               each wave gathers its 8 rows x 4 trajectories with lane = (point, channel slice) and feeds the
//               gathered [self | lower | upper] vectors straight into f32 MFMA against weights held in registers |
//               activation epilogue + store | barrier
#include <algorithm>
#include <cstring>
#include <numeric>

#include "scn_internal.h"

namespace scn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

int build_block_plan(scn_conv_s* c) {
    // only the scone/ebli shape gets a plan; everything else runs the generic kernels
    if (c->n_groups != 1) return SCN_OK;
    const Group& G = c->g[0];
    if (!(G.identity == 1 && G.n_vals == 2)) return SCN_OK;
    const int n_rows = c->n_rows;
    std::vector<int32_t> blk_row0, src_ptr(1, 0), src_rows, ell_ptr;
    std::vector<uint8_t> blk_rows, width, tile_w, ell_slot, self_slot;
    std::vector<float> ell_v0, ell_v1;
    std::vector<int32_t> mark(G.n_cols, -1), local(G.n_cols, 0), cur;
    cur.reserve(BK_SRC + 64);
    int64_t total_src = 0;
    int wmax = 0;
    int bid = 0;
    for (int r0 = 0; r0 < n_rows;) {
        cur.clear();
        int rows = 0, w = 0;
        while (r0 + rows < n_rows && rows < BK_R) {
            const int r = r0 + rows;
            const int j0 = G.h_rowptr[r], j1 = G.h_rowptr[r + 1];
            int add = 0;
            for (int j = j0; j < j1; ++j)
                if (mark[G.h_col[j]] != bid) ++add;
            bool self_new = mark[r] != bid;
            for (int j = j0; j < j1 && self_new; ++j)
                if (G.h_col[j] == r) self_new = false;
            if (self_new) ++add;
            const int nw = std::max(w, j1 - j0);
            if ((int)cur.size() + add > BK_SRC || nw > BK_MAXW) break;
            for (int j = j0; j < j1; ++j)
                if (mark[G.h_col[j]] != bid) { mark[G.h_col[j]] = bid; cur.push_back(G.h_col[j]); }
            if (mark[r] != bid) { mark[r] = bid; cur.push_back(r); }
            w = nw;
            ++rows;
        }
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
        const size_t base = ell_slot.size();
        ell_slot.resize(base + (size_t)w * BK_R, 0);
        ell_v0.resize(base + (size_t)w * BK_R, 0.f);
        ell_v1.resize(base + (size_t)w * BK_R, 0.f);
        uint8_t tw[4] = {0, 0, 0, 0};
        for (int i = 0; i < BK_R; ++i) {
            uint8_t ss = 0;
            if (i < rows) {
                const int r = r0 + i;
                const int j0 = G.h_rowptr[r], j1 = G.h_rowptr[r + 1];
                for (int j = j0; j < j1; ++j) {
                    const size_t e = base + (size_t)(j - j0) * BK_R + i;
                    ell_slot[e] = (uint8_t)local[G.h_col[j]];
                    ell_v0[e] = G.h_val0[j];
                    ell_v1[e] = G.h_val1[j];
                }
                tw[i >> 3] = std::max<uint8_t>(tw[i >> 3], (uint8_t)(j1 - j0));
                ss = (uint8_t)local[r];
            }
            self_slot.push_back(ss);
        }
        tile_w.insert(tile_w.end(), tw, tw + 4);
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
    if ((st = upload(c, ell_slot, &P.dev.ell_slot)) != SCN_OK) return st;
    if ((st = upload(c, ell_v0, &P.dev.ell_v0)) != SCN_OK) return st;
    if ((st = upload(c, ell_v1, &P.dev.ell_v1)) != SCN_OK) return st;
    if ((st = upload(c, self_slot, &P.dev.self_slot)) != SCN_OK) return st;
    P.mean_src_per_row = (double)total_src / std::max(1, n_rows);
    P.built = true;
    return SCN_OK;
}

// ------------------------------------------------------------------------------------------------
// shared device helpers
// ------------------------------------------------------------------------------------------------
// LDS carve: [stage: BK_SRC * PIECE bytes][v0: BK_R*wmax f32][v1: same][srcrows: BK_SRC i32][slot: BK_R*wmax u8][self: BK_R u8]
struct Smem {
    char* stage;
    float* v0;
    float* v1;
    int32_t* srcrows;
    uint8_t* slot;
    uint8_t* self;
};
__device__ __forceinline__ Smem carve(char* base, int piece, int wmax) {
    Smem s;
    s.stage = base;
    s.v0 = (float*)(base + BK_SRC * piece);
    s.v1 = s.v0 + BK_R * wmax;
    s.srcrows = (int32_t*)(s.v1 + BK_R * wmax);
    s.slot = (uint8_t*)(s.srcrows + BK_SRC);
    s.self = s.slot + BK_R * wmax;
    return s;
}
__host__ __device__ static inline size_t smem_bytes(int piece, int wmax, int extra = 0) {
    size_t b = (size_t)BK_SRC * piece + (size_t)BK_R * wmax * 9 + BK_SRC * 4 + BK_R;
    return ((b + 15) / 16) * 16 + extra;
}

struct BlockMeta { int row0, rows, nsrc, w; };

// loads the block's ELL tile / source list into LDS (caller brackets with barriers)
__device__ __forceinline__ BlockMeta load_block(const PlanDev& P, int b, const Smem& sm) {
    BlockMeta m;
    m.row0 = P.blk_row0[b];
    m.rows = P.blk_rows[b];
    const int sp0 = P.src_ptr[b];
    m.nsrc = P.src_ptr[b + 1] - sp0;
    m.w = P.width[b];
    const int ep = P.ell_ptr[b];
    for (int i = threadIdx.x; i < m.nsrc; i += BK_THREADS) sm.srcrows[i] = P.src_rows[sp0 + i];
    for (int i = threadIdx.x; i < m.w * BK_R; i += BK_THREADS) {
        sm.slot[i] = P.ell_slot[ep + i];
        sm.v0[i] = P.ell_v0[ep + i];
        sm.v1[i] = P.ell_v1[ep + i];
    }
    if (threadIdx.x < BK_R) sm.self[threadIdx.x] = P.self_slot[(size_t)b * BK_R + threadIdx.x];
    return m;
}

// XOR swizzle of the 16-byte chunk index inside a staged piece (keeps the lane=(point, channel slice) gather
// spread over the LDS banks): 512-B pieces (C=32): chunk = n*8 + h*4 + q ; 256-B pieces (C=16): chunk = n*4 + g.
__device__ __forceinline__ int swz32(int slot, int chunk) {
    return chunk ^ ((((slot >> 1) & 1) << 2) | ((slot & 1) << 1) | ((chunk >> 4) & 1));
}
__device__ __forceinline__ int swz16(int slot, int chunk) { return chunk ^ (slot & 3); }

// stage nsrc pieces of PIECE bytes of slab-base Xs into LDS (coalesced 16-B loads)
template <int PIECE, int SWZ>
__device__ __forceinline__ void stage_pieces(const char* Xs, const Smem& sm, int nsrc) {
    constexpr int CPP = PIECE / 16;
    const int total = nsrc * CPP;
    for (int c = threadIdx.x; c < total; c += BK_THREADS) {
        const int slot = c / CPP, ch = c % CPP;
        const f32x4 v = *(const f32x4*)(Xs + (size_t)sm.srcrows[slot] * PIECE + ch * 16);
        int pos = ch;
        if (SWZ == 32) pos = swz32(slot, ch);
        if (SWZ == 16) pos = swz16(slot, ch);
        *(f32x4*)(sm.stage + slot * PIECE + pos * 16) = v;
    }
}

// XCD-contiguous block range of this workgroup: blocks b = first, first+stride, ... < last
__device__ __forceinline__ void block_range(int n_blocks, int& first, int& last, int& stride) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    stride = gridDim.x >> 3;
    const int b0 = (int)((int64_t)n_blocks * xcd / 8);
    last = (int)((int64_t)n_blocks * (xcd + 1) / 8);
    first = b0 + j;
}

// ------------------------------------------------------------------------------------------------
// forward, C_in = C_out = 32   (v_mfma_f32_32x32x2_f32; lane = (point p = lane&31, channel half h = lane>>5))
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BK_THREADS, 2) void fwd_c32_kernel(PlanDev P, const float* __restrict__ X,
                                                                const float* __restrict__ W0,
                                                                const float* __restrict__ W1,
                                                                const float* __restrict__ W2,
                                                                float* __restrict__ out, int n_rows, int n_cols,
                                                                int n_slabs, int act) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PIECE = 512;
    const Smem sm = carve(smem, PIECE, P.ell_w_max);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5, n = p & 3, rt = wave * 8 + (p >> 2);
    // B operands: k-step s of segment g contracts channel 16*h + s
    float Bw[3][16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        Bw[0][s] = W0[(16 * h + s) * 32 + p];
        Bw[1][s] = W1[(16 * h + s) * 32 + p];
        Bw[2][s] = W2[(16 * h + s) * 32 + p];
    }
    int b, b_end, b_stride;
    block_range(P.n_blocks, b, b_end, b_stride);
    for (; b < b_end; b += b_stride) {
        __syncthreads();
        const BlockMeta m = load_block(P, b, sm);
        const int tw = P.tile_w[b * 4 + wave];
        const int cbase = n * 8 + h * 4;
        const int slab0 = blockIdx.y * n_slabs / gridDim.y, slab1 = (blockIdx.y + 1) * n_slabs / gridDim.y;
        for (int slab = slab0; slab < slab1; ++slab) {
            __syncthreads();
            stage_pieces<PIECE, 32>((const char*)X + (size_t)slab * n_cols * PIECE, sm, m.nsrc);
            __syncthreads();
            f32x4 zs[4], zl[4], zu[4];
            {
                const int slot = sm.self[rt];
                const char* base = sm.stage + slot * PIECE;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    zs[q] = *(const f32x4*)(base + swz32(slot, cbase + q) * 16);
                    zl[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    zu[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            for (int t = 0; t < tw; ++t) {
                const int e = t * BK_R + rt;
                const int slot = sm.slot[e];
                const float a0 = sm.v0[e], a1 = sm.v1[e];
                const char* base = sm.stage + slot * PIECE;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 d = *(const f32x4*)(base + swz32(slot, cbase + q) * 16);
                    zl[q] += a0 * d;
                    zu[q] += a1 * d;
                }
            }
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(zs[s >> 2][s & 3], Bw[0][s], acc, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(zl[s >> 2][s & 3], Bw[1][s], acc, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(zu[s >> 2][s & 3], Bw[2][s], acc, 0, 0, 0);
            // D layout: column = lane&31 (output channel), row = (r&3) + 8*(r>>2) + 4*h (point)
            float* o = out + ((size_t)slab * n_rows + m.row0) * (BK_NS * 32) + p;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pt = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int prow = wave * 8 + (pt >> 2);
                if (prow < m.rows) o[(prow * BK_NS + (pt & 3)) * 32] = act_apply(act, acc[r]);
            }
        }
    }
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
                                                                int n_slabs, int act) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PIECE = 256;
    const Smem sm = carve(smem, PIECE, P.ell_w_max);
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
    int b, b_end, b_stride;
    block_range(P.n_blocks, b, b_end, b_stride);
    for (; b < b_end; b += b_stride) {
        __syncthreads();
        const BlockMeta m = load_block(P, b, sm);
        const int tw = P.tile_w[b * 4 + wave];
        const int chunk = n * 4 + g;
        const int slab0 = blockIdx.y * n_slabs / gridDim.y, slab1 = (blockIdx.y + 1) * n_slabs / gridDim.y;
        for (int slab = slab0; slab < slab1; ++slab) {
            __syncthreads();
            stage_pieces<PIECE, 16>((const char*)X + (size_t)slab * n_cols * PIECE, sm, m.nsrc);
            __syncthreads();
            f32x4 zsA, zsB, zlA = {0.f, 0.f, 0.f, 0.f}, zuA = zlA, zlB = zlA, zuB = zlA;
            {
                const int sA = sm.self[rtA], sB = sm.self[rtB];
                zsA = *(const f32x4*)(sm.stage + sA * PIECE + swz16(sA, chunk) * 16);
                zsB = *(const f32x4*)(sm.stage + sB * PIECE + swz16(sB, chunk) * 16);
            }
            for (int t = 0; t < tw; ++t) {
                const int eA = t * BK_R + rtA, eB = eA + 4;
                const int sA = sm.slot[eA], sB = sm.slot[eB];
                const f32x4 dA = *(const f32x4*)(sm.stage + sA * PIECE + swz16(sA, chunk) * 16);
                const f32x4 dB = *(const f32x4*)(sm.stage + sB * PIECE + swz16(sB, chunk) * 16);
                zlA += sm.v0[eA] * dA;
                zuA += sm.v1[eA] * dA;
                zlB += sm.v0[eB] * dB;
                zuB += sm.v1[eB] * dB;
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
            // D layout: column = lane&15 (output channel), row = 4*g + r (point): row-in-quad = g, trajectory = r
            float* o = out + ((size_t)slab * n_rows + m.row0) * (BK_NS * 16) + p;
            const int prA = wave * 8 + g, prB = prA + 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (prA < m.rows) o[(prA * BK_NS + r) * 16] = act_apply(act, accA[r]);
                if (prB < m.rows) o[(prB * BK_NS + r) * 16] = act_apply(act, accB[r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward, C_in = 1 -> C_out = C (first layer, TE:143-147 with flow (E,1)): three gathered scalars per point,
// then out = act(x*w0 + lo*w1 + up*w2) written as coalesced float4.
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(BK_THREADS, 2) void fwd_c1_kernel(PlanDev P, const float* __restrict__ X,
                                                               const float* __restrict__ W0,
                                                               const float* __restrict__ W1,
                                                               const float* __restrict__ W2,
                                                               float* __restrict__ out, int n_rows, int n_cols,
                                                               int n_slabs, int act) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PIECE = 16, CQ = C / 4;
    const Smem sm = carve(smem, PIECE, P.ell_w_max);
    float* Z = (float*)(smem + smem_bytes(PIECE, P.ell_w_max));      // [BK_R*BK_NS][3]
    const int tid = threadIdx.x;
    const int cq = tid % CQ;                                          // constant per thread: 256 % CQ == 0
    f32x4 w0 = *(const f32x4*)(W0 + cq * 4), w1 = *(const f32x4*)(W1 + cq * 4), w2 = *(const f32x4*)(W2 + cq * 4);
    int b, b_end, b_stride;
    block_range(P.n_blocks, b, b_end, b_stride);
    const int slab0 = blockIdx.y * n_slabs / gridDim.y, slab1 = (blockIdx.y + 1) * n_slabs / gridDim.y;
    for (; b < b_end; b += b_stride) {
        __syncthreads();
        const BlockMeta m = load_block(P, b, sm);
        for (int slab = slab0; slab < slab1; ++slab) {
            __syncthreads();
            stage_pieces<PIECE, 0>((const char*)X + (size_t)slab * n_cols * PIECE, sm, m.nsrc);
            __syncthreads();
            if (tid < BK_R * BK_NS) {
                const int rt = tid >> 2, n = tid & 3;
                const float* st = (const float*)sm.stage;
                float zs = st[sm.self[rt] * 4 + n], zl = 0.f, zu = 0.f;
                const int tw = P.tile_w[b * 4 + (rt >> 3)];
                for (int t = 0; t < tw; ++t) {
                    const int e = t * BK_R + rt;
                    const float d = st[sm.slot[e] * 4 + n];
                    zl = fmaf(sm.v0[e], d, zl);
                    zu = fmaf(sm.v1[e], d, zu);
                }
                Z[tid * 3] = zs; Z[tid * 3 + 1] = zl; Z[tid * 3 + 2] = zu;
            }
            __syncthreads();
            float* o = out + ((size_t)slab * n_rows + m.row0) * (BK_NS * C);
            const int total = m.rows * BK_NS * CQ;
            for (int idx = tid; idx < total; idx += BK_THREADS) {
                const int pt = idx / CQ;
                const float zs = Z[pt * 3], zl = Z[pt * 3 + 1], zu = Z[pt * 3 + 2];
                f32x4 v = zs * w0 + zl * w1 + zu * w2;
                v[0] = act_apply(act, v[0]); v[1] = act_apply(act, v[1]);
                v[2] = act_apply(act, v[2]); v[3] = act_apply(act, v[3]);
                *(f32x4*)(o + (size_t)idx * 4) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dual SpMM on K-float pieces (K % 4 == 0, K <= 128): ya = val0-operator * x, yb = val1-operator * x
// ------------------------------------------------------------------------------------------------
template <bool DUAL>
__global__ __launch_bounds__(BK_THREADS, 2) void spmm_blocked_kernel(PlanDev P, const float* __restrict__ X,
                                                                     float* __restrict__ ya, float* __restrict__ yb,
                                                                     int n_rows, int n_cols, int n_slabs, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int piece = K * 4, cpp = K / 4;
    const Smem sm = carve(smem, piece, P.ell_w_max);
    int b, b_end, b_stride;
    block_range(P.n_blocks, b, b_end, b_stride);
    const int slab0 = blockIdx.y * n_slabs / gridDim.y, slab1 = (blockIdx.y + 1) * n_slabs / gridDim.y;
    for (; b < b_end; b += b_stride) {
        __syncthreads();
        const BlockMeta m = load_block(P, b, sm);
        for (int slab = slab0; slab < slab1; ++slab) {
            __syncthreads();
            {
                const char* Xs = (const char*)X + (size_t)slab * n_cols * piece;
                const int total = m.nsrc * cpp;
                for (int c = threadIdx.x; c < total; c += BK_THREADS) {
                    const int slot = c / cpp, ch = c - slot * cpp;
                    *(f32x4*)(sm.stage + slot * piece + ch * 16) =
                        *(const f32x4*)(Xs + (size_t)sm.srcrows[slot] * piece + ch * 16);
                }
            }
            __syncthreads();
            const size_t obase = ((size_t)slab * n_rows + m.row0) * K;
            const int total = m.rows * cpp;
            for (int idx = threadIdx.x; idx < total; idx += BK_THREADS) {
                const int r = idx / cpp, ch = idx - r * cpp;
                const int tw = P.tile_w[b * 4 + (r >> 3)];
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
                for (int t = 0; t < tw; ++t) {
                    const int e = t * BK_R + r;
                    const f32x4 d = *(const f32x4*)(sm.stage + sm.slot[e] * piece + ch * 16);
                    a0 += sm.v0[e] * d;
                    if (DUAL) a1 += sm.v1[e] * d;
                }
                *(f32x4*)(ya + obase + (size_t)idx * 4) = a0;
                if (DUAL) *(f32x4*)(yb + obase + (size_t)idx * 4) = a1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, c_dz = c_aux = 32
//   G = [dz | S_lower^T dz | S_upper^T dz] gathered like the forward; dx = (G @ [W0;W1;W2]^T) * act'(aux);
//   dW_g += aux^T G_g via MFMA with the points on the contraction axis (G transposed through a per-wave LDS patch).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
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
    constexpr int PIECE = 512;
    const Smem sm = carve(smem, PIECE, P.ell_w_max);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* patch = (float*)(smem + smem_bytes(PIECE, P.ell_w_max)) + wave * (32 * T32_STRIDE);
    const int p = lane & 31, h = lane >> 5, n = p & 3, rt = wave * 8 + (p >> 2);
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
    const int slab0 = blockIdx.y * n_slabs / gridDim.y, slab1 = (blockIdx.y + 1) * n_slabs / gridDim.y;
    for (; b < b_end; b += b_stride) {
        __syncthreads();
        const BlockMeta m = load_block(P, b, sm);
        const int tw = P.tile_w[b * 4 + wave];
        const int cbase = n * 8 + h * 4;
        for (int slab = slab0; slab < slab1; ++slab) {
            __syncthreads();
            stage_pieces<PIECE, 32>((const char*)DZ + (size_t)slab * n_cols * PIECE, sm, m.nsrc);
            // this lane's aux values: point pt(r) = (r&3) + 8*(r>>2) + 4*h, channel ca = p  (also the dW A operand)
            const size_t tbase = ((size_t)slab * n_rows + m.row0) * (BK_NS * 32) + p;
            float a[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pt = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int prow = wave * 8 + (pt >> 2);
                a[r] = prow < m.rows ? aux[tbase + (prow * BK_NS + (pt & 3)) * 32] : 0.f;
            }
            __syncthreads();
            f32x4 gs[4], gl[4], gu[4];
            {
                const int slot = sm.self[rt];
                const char* base = sm.stage + slot * PIECE;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    gs[q] = *(const f32x4*)(base + swz32(slot, cbase + q) * 16);
                    gl[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    gu[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            for (int t = 0; t < tw; ++t) {
                const int e = t * BK_R + rt;
                const int slot = sm.slot[e];
                const float a0 = sm.v0[e], a1 = sm.v1[e];
                const char* base = sm.stage + slot * PIECE;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 d = *(const f32x4*)(base + swz32(slot, cbase + q) * 16);
                    gl[q] += a0 * d;
                    gu[q] += a1 * d;
                }
            }
            if (dx) {
                f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gs[s >> 2][s & 3], Bt[0][s], acc, 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gl[s >> 2][s & 3], Bt[1][s], acc, 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gu[s >> 2][s & 3], Bt[2][s], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pt = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int prow = wave * 8 + (pt >> 2);
                    if (prow < m.rows)
                        dx[tbase + (prow * BK_NS + (pt & 3)) * 32] = acc[r] * act_grad_from_output(act, a[r]);
                }
            }
            // dW_g += aux^T G_g : A[i = ca][k = point] = a[s], B[k = point][j = c] = G_g[pt(s,h)][c = p]
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const f32x4* src = g == 0 ? gs : (g == 1 ? gl : gu);
#pragma unroll
                for (int q = 0; q < 4; ++q) *(f32x4*)(patch + p * T32_STRIDE + 16 * h + 4 * q) = src[q];
                wave_lds_fence();
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int pt = (s & 3) + 8 * (s >> 2) + 4 * h;
                    dWacc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], patch[pt * T32_STRIDE + p], dWacc[g], 0, 0, 0);
                }
                wave_lds_fence();
            }
        }
    }
    // reduce the four waves' dW tiles in a fixed order and emit this workgroup's partial [ca][slot*32 + c]
    __syncthreads();
    float* red = (float*)sm.stage;                              // 4 waves * 3 * 1024 floats = 48 KB
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ca = (r & 3) + 8 * (r >> 2) + 4 * h;
            red[wave * 3072 + ca * 96 + g * 32 + p] = dWacc[g][r];
        }
    __syncthreads();
    float* out = partial + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 3072;
    for (int i = threadIdx.x; i < 3072; i += BK_THREADS)
        out[i] = (red[i] + red[3072 + i]) + (red[2 * 3072 + i] + red[3 * 3072 + i]);
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
                                                                int n_slabs, int act) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PIECE = 256;
    const Smem sm = carve(smem, PIECE, P.ell_w_max);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* patch = (float*)(smem + smem_bytes(PIECE, P.ell_w_max)) + wave * (2 * 16 * T16_STRIDE);
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
    int b, b_end, b_stride;
    block_range(P.n_blocks, b, b_end, b_stride);
    const int slab0 = blockIdx.y * n_slabs / gridDim.y, slab1 = (blockIdx.y + 1) * n_slabs / gridDim.y;
    for (; b < b_end; b += b_stride) {
        __syncthreads();
        const BlockMeta m = load_block(P, b, sm);
        const int tw = P.tile_w[b * 4 + wave];
        const int chunk = n * 4 + g4;
        const int prA = wave * 8 + g4, prB = prA + 4;       // D-layout rows: point 4*g4 + r -> row-in-quad g4, trajectory r
        for (int slab = slab0; slab < slab1; ++slab) {
            __syncthreads();
            stage_pieces<PIECE, 16>((const char*)DZ + (size_t)slab * n_cols * PIECE, sm, m.nsrc);
            const size_t tbase = ((size_t)slab * n_rows + m.row0) * (BK_NS * 16) + p;
            float aA[4], aB[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                aA[r] = prA < m.rows ? aux[tbase + (prA * BK_NS + r) * 16] : 0.f;
                aB[r] = prB < m.rows ? aux[tbase + (prB * BK_NS + r) * 16] : 0.f;
            }
            __syncthreads();
            f32x4 G[3][2];
            {
                const int sA = sm.self[rtA], sB = sm.self[rtB];
                G[0][0] = *(const f32x4*)(sm.stage + sA * PIECE + swz16(sA, chunk) * 16);
                G[0][1] = *(const f32x4*)(sm.stage + sB * PIECE + swz16(sB, chunk) * 16);
                G[1][0] = G[1][1] = G[2][0] = G[2][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            for (int t = 0; t < tw; ++t) {
                const int eA = t * BK_R + rtA, eB = eA + 4;
                const int sA = sm.slot[eA], sB = sm.slot[eB];
                const f32x4 dA = *(const f32x4*)(sm.stage + sA * PIECE + swz16(sA, chunk) * 16);
                const f32x4 dB = *(const f32x4*)(sm.stage + sB * PIECE + swz16(sB, chunk) * 16);
                G[1][0] += sm.v0[eA] * dA;
                G[2][0] += sm.v1[eA] * dA;
                G[1][1] += sm.v0[eB] * dB;
                G[2][1] += sm.v1[eB] * dB;
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
                wave_lds_fence();
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int pt = 4 * g4 + s;
                    dWacc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(aA[s], patch[pt * T16_STRIDE + p], dWacc[g], 0, 0, 0);
                    dWacc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(aB[s], patch[16 * T16_STRIDE + pt * T16_STRIDE + p], dWacc[g], 0, 0, 0);
                }
                wave_lds_fence();
            }
        }
    }
    __syncthreads();
    float* red = (float*)sm.stage;                              // 4 waves * 768 floats
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * 768 + (4 * g4 + r) * 48 + g * 16 + p] = dWacc[g][r];
    __syncthreads();
    float* out = partial + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 768;
    for (int i = threadIdx.x; i < 768; i += BK_THREADS)
        out[i] = (red[i] + red[768 + i]) + (red[2 * 768 + i] + red[3 * 768 + i]);
}

// ------------------------------------------------------------------------------------------------
// backward of the first layer: c_dz = C, c_aux = 1, dW only:  dW_g[0][c] += sum_p x[p] * G_g[p][c]
// thread = (row, trajectory, channel quad); per-thread accumulators reduced once at the end.
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(BK_THREADS, 2) void bwd_c1_kernel(PlanDev P, const float* __restrict__ DZ,
                                                               const float* __restrict__ aux,
                                                               float* __restrict__ partial, int n_rows, int n_cols,
                                                               int n_slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PIECE = BK_NS * C * 4, CPP = PIECE / 16, CQ = C / 4;
    const Smem sm = carve(smem, PIECE, P.ell_w_max);
    const int tid = threadIdx.x;
    f32x4 acc[3];
    acc[0] = acc[1] = acc[2] = f32x4{0.f, 0.f, 0.f, 0.f};
    int b, b_end, b_stride;
    block_range(P.n_blocks, b, b_end, b_stride);
    const int slab0 = blockIdx.y * n_slabs / gridDim.y, slab1 = (blockIdx.y + 1) * n_slabs / gridDim.y;
    for (; b < b_end; b += b_stride) {
        __syncthreads();
        const BlockMeta m = load_block(P, b, sm);
        for (int slab = slab0; slab < slab1; ++slab) {
            __syncthreads();
            stage_pieces<PIECE, 0>((const char*)DZ + (size_t)slab * n_cols * PIECE, sm, m.nsrc);
            __syncthreads();
            const float* xs = aux + ((size_t)slab * n_rows + m.row0) * BK_NS;
            const int total = m.rows * CPP;                       // (row, chunk) pairs; chunk = n*CQ + cq
            for (int idx = tid; idx < total; idx += BK_THREADS) {  // 256 % CPP == 0 -> cq constant per thread
                const int r = idx / CPP, ch = idx - r * CPP;
                const float x = xs[r * BK_NS + ch / CQ];
                const int tw = P.tile_w[b * 4 + (r >> 3)];
                f32x4 gl = {0.f, 0.f, 0.f, 0.f}, gu = gl;
                const f32x4 gsv = *(const f32x4*)(sm.stage + sm.self[r] * PIECE + ch * 16);
                for (int t = 0; t < tw; ++t) {
                    const int e = t * BK_R + r;
                    const f32x4 d = *(const f32x4*)(sm.stage + sm.slot[e] * PIECE + ch * 16);
                    gl += sm.v0[e] * d;
                    gu += sm.v1[e] * d;
                }
                acc[0] += x * gsv;
                acc[1] += x * gl;
                acc[2] += x * gu;
            }
        }
    }
    // threads with equal cq = tid % CQ hold the same channels: reduce over them in a fixed order
    __syncthreads();
    f32x4* red = (f32x4*)sm.stage;                                // [3][256]
#pragma unroll
    for (int g = 0; g < 3; ++g) red[g * BK_THREADS + tid] = acc[g];
    __syncthreads();
    if (tid < 3 * CQ) {
        const int g = tid / CQ, cq = tid - g * CQ;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int t = cq; t < BK_THREADS; t += CQ) s += red[g * BK_THREADS + t];
        float* out = partial + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (3 * C) + g * C + cq * 4;
        out[0] = s[0]; out[1] = s[1]; out[2] = s[2]; out[3] = s[3];
    }
}

// dW_slot[i] += sum over partials (fixed order).  layout: partial[b][ca*ncols3 + slot*c + cc]
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

static void launch_grid(const scn_conv_s* c, int n_slabs, dim3& grid) {
    const int nb = c->plan.dev.n_blocks;
    int gx = std::min(512, ((nb + 7) / 8) * 8);
    gx = std::max(gx, 8);
    int gy = 1;
    if (gx < 512) gy = std::max(1, std::min(n_slabs, 512 / gx));
    grid = dim3(gx, gy);
}

bool blocked_forward_supported(const scn_conv_s* c, int ns, const int32_t* c_in, int c_out) {
    if (!scone_shape(c) || ns != BK_NS) return false;
    const int ci = c_in[0];
    return (ci == 32 && c_out == 32) || (ci == 16 && c_out == 16) || (ci == 1 && (c_out == 16 || c_out == 32));
}

int blocked_forward(scn_conv_s* c, int n_slabs, int ns, const float* const* src, const int32_t* c_in,
                    const float* const* W, int c_out, int act, float* out, hipStream_t st) {
    const PlanDev& P = c->plan.dev;
    dim3 grid;
    launch_grid(c, n_slabs, grid);
    const int ci = c_in[0];
    const int nr = c->n_rows, nc = c->g[0].n_cols;
    if (ci == 32) {
        const size_t lds = smem_bytes(512, P.ell_w_max);
        SCN_ENSURE_LDS(fwd_c32_kernel, lds);
        hipLaunchKernelGGL(fwd_c32_kernel, grid, dim3(BK_THREADS), lds, st, P, src[0], W[0], W[1], W[2], out, nr, nc,
                           n_slabs, act);
    } else if (ci == 16) {
        const size_t lds = smem_bytes(256, P.ell_w_max);
        SCN_ENSURE_LDS(fwd_c16_kernel, lds);
        hipLaunchKernelGGL(fwd_c16_kernel, grid, dim3(BK_THREADS), lds, st, P, src[0], W[0], W[1], W[2], out, nr, nc,
                           n_slabs, act);
    } else if (c_out == 32) {
        hipLaunchKernelGGL(fwd_c1_kernel<32>, grid, dim3(BK_THREADS), smem_bytes(16, P.ell_w_max, BK_R * BK_NS * 12), st, P,
                           src[0], W[0], W[1], W[2], out, nr, nc, n_slabs, act);
    } else {
        hipLaunchKernelGGL(fwd_c1_kernel<16>, grid, dim3(BK_THREADS), smem_bytes(16, P.ell_w_max, BK_R * BK_NS * 12), st, P,
                           src[0], W[0], W[1], W[2], out, nr, nc, n_slabs, act);
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

size_t blocked_backward_workspace(const scn_conv_s* c, int n_slabs, int ns, const int32_t* c_dz, int c_aux) {
    if (!blocked_backward_supported(c, ns, c_dz, c_aux, false) && !blocked_backward_supported(c, ns, c_dz, c_aux, true))
        return 0;
    dim3 grid;
    launch_grid(c, n_slabs, grid);
    return (size_t)grid.x * grid.y * c_aux * 3 * c_dz[0] * sizeof(float);
}

int blocked_backward(scn_conv_s* c, int n_slabs, int ns, const float* const* dz, const int32_t* c_dz,
                     const float* const* W, const float* aux, int c_aux, int act, float* dx,
                     float* const* dW, void* ws, size_t ws_bytes, hipStream_t st) {
    const PlanDev& P = c->plan.dev;
    dim3 grid;
    launch_grid(c, n_slabs, grid);
    const int cd = c_dz[0];
    const int nr = c->n_rows, nc = c->g[0].n_cols;
    float* partial = (float*)ws;
    if (c_aux == 32) {
        const size_t lds = smem_bytes(512, P.ell_w_max, 4 * 32 * T32_STRIDE * 4);
        SCN_ENSURE_LDS(bwd_c32_kernel, lds);
        hipLaunchKernelGGL(bwd_c32_kernel, grid, dim3(BK_THREADS), lds, st, P, dz[0], W[0], W[1], W[2], aux, dx, partial, nr,
                           nc, n_slabs, act);
    } else if (c_aux == 16) {
        const size_t lds = smem_bytes(256, P.ell_w_max, 4 * 2 * 16 * T16_STRIDE * 4);
        SCN_ENSURE_LDS(bwd_c16_kernel, lds);
        hipLaunchKernelGGL(bwd_c16_kernel, grid, dim3(BK_THREADS), lds, st, P, dz[0], W[0], W[1], W[2], aux, dx, partial, nr,
                           nc, n_slabs, act);
    } else if (cd == 32) {
        SCN_ENSURE_LDS(bwd_c1_kernel<32>, smem_bytes(512, P.ell_w_max));
        hipLaunchKernelGGL(bwd_c1_kernel<32>, grid, dim3(BK_THREADS), smem_bytes(512, P.ell_w_max), st, P, dz[0], aux, partial,
                           nr, nc, n_slabs);
    } else {
        hipLaunchKernelGGL(bwd_c1_kernel<16>, grid, dim3(BK_THREADS), smem_bytes(256, P.ell_w_max), st, P, dz[0], aux, partial,
                           nr, nc, n_slabs);
    }
    SCN_LAUNCH_CHECK();
    const int per = c_aux * 3 * cd;
    hipLaunchKernelGGL(blocked_dw_reduce, dim3((per + 255) / 256), dim3(256), 0, st, partial, (int)(grid.x * grid.y), c_aux,
                       cd, dW[0], dW[1], dW[2]);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

bool blocked_spmm_supported(const scn_conv_s* c, int k) {
    return scone_shape(c) && k % 4 == 0 && k >= 4 && k <= 128;
}

int blocked_spmm(scn_conv_s* c, int n_slabs, int k, const float* x, float* ya, float* yb, hipStream_t st) {
    const PlanDev& P = c->plan.dev;
    dim3 grid;
    launch_grid(c, n_slabs, grid);
    const size_t lds = smem_bytes(k * 4, P.ell_w_max);
    SCN_ENSURE_LDS(spmm_blocked_kernel<true>, lds);
    SCN_ENSURE_LDS(spmm_blocked_kernel<false>, lds);
    if (yb)
        hipLaunchKernelGGL(spmm_blocked_kernel<true>, grid, dim3(BK_THREADS), lds, st, P, x, ya, yb, c->n_rows, c->g[0].n_cols,
                           n_slabs, k);
    else
        hipLaunchKernelGGL(spmm_blocked_kernel<false>, grid, dim3(BK_THREADS), lds, st, P, x, ya, yb, c->n_rows,
                           c->g[0].n_cols, n_slabs, k);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

}  // namespace scn
