// The whole gradient step of a SCoNe model on a SMALL complex in one launch (scn_small_step).
//
// The reference's own problem sizes (TE:86-90: |E| = 1001, batch 100; the drifter data set: |E| = 320) are launch-bound on the
// layer-by-layer kernels: five fused conv launches, readout, cross-entropy and their reductions are ~17 launches of 4-18 us
// each per optimiser step (profiles/HISTORY.md section 3.4), and what every launch does is a chain of a few dependent L2 round trips.
// Trajectories do not interact (TE:256 vmap) -- only the weight gradient sums over them -- and on a complex this small ONE
// trajectory's activation tensor (|E| x 16 floats) fits the LDS of a CU.  So: one workgroup per trajectory runs every layer,
// the readout, the cross-entropy and the whole backward on LDS-resident activations, with workgroup barriers between layers and
// no halo, no staging and no inter-workgroup dependency; the per-trajectory weight-gradient partials are then summed in a fixed
// order by a second, tiny launch.  Where half the device would idle (|E| > 384 and 2 x trajectories <= CUs) a trajectory gets TWO
// workgroups that hand each other their rows after every layer (PAIRED, see small_step_kernel).
//
//   lane (r = lane & 15, q = lane >> 4) of a wave owns row 16 t + r of tile t and the channel quad 4 q .. 4 q + 3: it gathers
//   sum_j S[row][col_j] * H[col_j][quad] with 16-byte LDS reads, and that register layout IS the B operand of
//   v_mfma_f32_16x16x4_f32 (K = channel, N = row), whose D tile comes back in the same (row, quad) layout: out^T = W^T z^T for
//   the forward, dX^T = W G^T for the backward.  The weight gradient dW = aux^T G needs its operands with the rows along K: aux and
//   segment 0 (dz itself) are read from LDS that way, the two gathered segments are transposed in registers by an MFMA against the
//   identity (exact: products with 1.0 and 0.0).  Everything is fp32 FMA arithmetic -- no split.
//   The operator's rows stay in REGISTERS for the whole launch (SmOp below): a lane serves the same rows in every layer.
//
// Served: hidden width 16, one input channel, 2 .. 6 layers, |E| small enough for two activation buffers in 160 KB of LDS (~1100).
// Measured against the layer-by-layer kernels: profiles/r04_small_step_ab.txt, profiles/r05_small_pair_ab.txt; when the trainer
// takes it: ops.small_step_pays.
#include "scn_internal.h"

#include <atomic>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

namespace scn {

constexpr int SM_C = 16;
// (SM_WAVES_MAX = 12)                          // waves per workgroup: 8 (256 registers per lane: up to nine row tiles per wave resident) or, for
                                                // complexes of at most 24 tiles, 12 (170 registers, one or two tiles per wave, three waves per SIMD)
constexpr int SM_MAX_LAYERS = 6;
constexpr int SM_ITEMS = 512, SM_MAXD = 64;     // readout item list / neighbourhood width one wave handles (as scn_readout.hip)
constexpr int SM_LAYER_W = 3 * SM_C * SM_C;     // 768 weights per layer
constexpr int SM_CH = 12;                       // operator entries of a row kept in registers (three per lane of the row's four); longer rows
                                                // take their tail from the overflow list in LDS (from memory where it has no room)
constexpr int SM_RO_PRE = 6;                    // readout items per neighbour slot requested at kernel start
constexpr int SM_MAXT = 9;                      // 16-row tiles per wave at most: |E| <= 16 * 8 * 9 = 1152

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- diagnostic build only (-DSCN_STAMPS, tools/build_stamps.sh): wall-clock stamps (100 MHz) of workgroup 0 at the phase boundaries
#ifdef SCN_STAMPS
__device__ unsigned long long g_small_stamps[16];
#define SM_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_small_stamps[k] = wall_clock64(); } while (0)
__device__ unsigned long long g_small_cycles[16];              // shader-clock stamps inside one forward tile (tile 1 of wave 0, layer 2)
#define SM_CYC(k, cond) do { if (cond) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        if (threadIdx.x == 0 && blockIdx.x == 0) g_small_cycles[k] = clock64(); } } while (0)
#else
#define SM_STAMP(k)
#define SM_CYC(k, cond)
#endif

struct SmallArgs {
    int32_t n_edges, ns, n_layers, act, max_deg, same_t;
    const int32_t* rowptr;   const float4* ent;      // operator rows: (col, val_lower, val_upper, 0) per entry
    const int32_t* rowptr_t; const float4* ent_t;    // its transpose (the same arrays for symmetric shifts: same_t)
    const float4* ell; const float4* ell_t;           // [E][12] the rows' first twelve entries at a fixed stride, columns as LDS offsets
    const float4* ovf;   const int32_t* ovf_ptr;     // the entries past a row's twelfth in the same form, a row's range among them ([E + 1]);
    const float4* ovf_t; const int32_t* ovf_ptr_t;   // n_ovf / n_ovf_t entries in all
    int32_t n_ovf, n_ovf_t;
    const float* x;                                   // [S][E][ns]
    const int32_t* last_nodes;                        // [S * ns]
    const float* y;                                   // [S * ns][max_deg]
    float scale;
    const int32_t* nbr; const int32_t* inc_ptr; const int32_t* inc_edge; const float* inc_sign;
    const float* W[3 * SM_MAX_LAYERS + 1];
    float* hs;                                        // [n_layers - 1][N][E][16]  saved activations H_1 .. H_{L-1}
    float* ys;                                        // [N][E][4]                 (x, S_lo x, S_up x, 0): written for non-symmetric shifts only
    float* partial;                                   // [P][PW]                   per-workgroup weight-gradient partials (P = N, paired: 2 N)
    double* loss_part;                                // [P]
    // paired form (two workgroups per trajectory, see small_step_kernel): hand-over buffers and flags
    int32_t n_traj, tag;
    float* xg;                                        // [n_layers - 1][N][E][16]  H_L, then the dx of layers L .. 3
    int* flags;                                       // [N][2 SM_MAX_LAYERS][2]   phase p of half h has been stored (= tag)
    int32_t* step;                                    // scn_small_step_adam: [0] the next optimiser step's index, [1] the one in flight (or null)
};

__host__ __device__ static inline int small_pw(int n_layers) { return 3 * SM_C + (n_layers - 1) * SM_LAYER_W + SM_C; }

// LDS: two activation buffers, one layer's weights, readout scratch, the input flow, reduction scratch
struct SmallLds {
    int epad;
    bool y_lds;                                       // (S_lo x, S_up x) of every row stay in LDS for the first layer's weight gradient
    bool ovf_lds;                                     // the operator's entries past a row's twelfth stay in LDS (else: read from memory)
    size_t off_w, off_misc, off_x, off_red, off_y, off_ovf, total;
};
__host__ __device__ static inline int small_waves(int n_edges) { return ((n_edges + 15) >> 4) <= 24 ? 12 : 8; }
__host__ __device__ static inline SmallLds small_lds(int n_edges, int n_ovf) {
    const int waves = small_waves(n_edges);
    SmallLds L;
    L.epad = (n_edges + 15) & ~15;
    const size_t buf = (size_t)L.epad * SM_C * 4;
    L.off_w = 2 * buf;
    L.off_misc = L.off_w + SM_LAYER_W * 4;
    // misc: lgs[64] dl[64] wlast[16] d_ptr[80] it_e[512] it_s[512] it_d[512] bh[64*16] dwf[waves*48]
    const size_t misc = (64 + 64 + 16 + 80 + 3 * SM_ITEMS + 64 * SM_C + waves * 48) * 4;
    L.off_x = L.off_misc + misc;
    L.off_red = L.off_x + (size_t)L.epad * 4;
    const size_t red = (size_t)waves * SM_LAYER_W * 4;
    L.total = L.off_red + (buf >= red ? 0 : red);            // big buffers: the reduction overlays the dead input buffer
    const size_t ovf = ((size_t)n_ovf * 12 + 15) & ~(size_t)15;    // three floats per entry
    L.off_ovf = L.total;
    L.ovf_lds = n_ovf > 0 && waves == 8 && L.total + ovf <= 160 * 1024;    // (every layer needs it, both directions: before the first layer's y
                                                             //  below; the twelve-wave kernels of |E| <= 384 measured no gain: 0.053 / 0.054 ms)
    if (L.ovf_lds) L.total += ovf;
    L.off_y = L.total;
    L.y_lds = L.total + (size_t)L.epad * 8 <= 160 * 1024;    // (the largest complexes recompute them from the flow instead)
    if (L.y_lds) L.total += (size_t)L.epad * 8;
    return L;
}

__device__ __forceinline__ float sm_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float sm_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int sm_excl_scan(int v, int lane) {
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    return x - v;
}

// act on a channel quad: one wave-uniform switch per quad; tanh is scn::fast_tanh's two forms evaluated side by side and selected (the
// compiler's version of that function branches per element on whether ANY lane is small: four divergent regions per tile)
__device__ __forceinline__ f32x4 sm_act4(int act, f32x4 z) {
    f32x4 o;
    if (act == SCN_ACT_TANH) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = z[j], ax = fabsf(x), x2 = x * x;
            const float poly = x * fmaf(x2, fmaf(x2, fmaf(x2, -17.f / 315.f, 2.f / 15.f), -1.f / 3.f), 1.f);
            const float t = __builtin_amdgcn_exp2f(ax * 2.885390081777927f);
            const float r = 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
            float big = __builtin_copysignf(r, x);
            asm volatile("" : "+v"(big));                        // (computed unconditionally: the select below stays a v_cndmask)
            o[j] = ax < 0.125f ? poly : big;
        }
    } else if (act == SCN_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaxf(z[j], 0.f);
    } else if (act == SCN_ACT_LEAKY_RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = z[j] >= 0.f ? z[j] : 0.01f * z[j];
    } else {
        o = z;
    }
    return o;
}

// Activation buffers hold row r's channel quad q at position q ^ ((r >> 2) & 3) of the row's 64 bytes: the sixteen lanes of a gather
// read sixteen unrelated rows at the SAME quad, which without the rotation is four bank groups for sixteen lanes.
__device__ __forceinline__ int sm_at(int r, int q) { return r * SM_C + 4 * (q ^ ((r >> 2) & 3)); }
__device__ __forceinline__ int sm_at1(int r, int c) { return r * SM_C + 4 * ((c >> 2) ^ ((r >> 2) & 3)) + (c & 3); }

// byte offset of row c's quad q in an activation buffer = sm_enc(c) ^ (16 * q): one v_xor per read instead of the five of sm_at
__device__ __forceinline__ uint32_t sm_enc(int c) { return (uint32_t)(c * SM_C + 4 * ((c >> 2) & 3)) * 4u; }

// The operator's rows of this lane, resident in registers for the whole kernel: tile k of the wave is rows 16 (wave + 8 k) .. + 15,
// lane (r16, q) keeps entries u = q, q + 4, q + 8 of row 16 t + r16; the four lanes of a row hand each other their entries with
// ds_bpermute (no memory traffic).  Every layer and, for symmetric shifts, the backward reuse them: the operator is read from memory
// ONCE per launch instead of once per layer and tile -- those reads, a round trip to L2 each, were what the first form of this kernel
// spent its time on (22 us per layer at |E| = 1001).
template <int MAXT>
struct SmOp {
    float enc[MAXT][3], v0[MAXT][3], v1[MAXT][3];     // enc: the column as a ready byte offset into an activation buffer (sm_enc; >> 6 = the column)
    int cnt[MAXT], cmax[MAXT];                        // entries of the row, most entries of any row of the tile (wave-uniform)
    int ovs[MAXT];                                    // where the row's entries past the twelfth start in the overflow list
};
template <int MAXT, int WAVES, bool OVF>
__device__ __forceinline__ void sm_load_op(SmOp<MAXT>& op, const int32_t* __restrict__ rowptr, const float4* __restrict__ ell,
                                           const int32_t* __restrict__ ovf_ptr, int E, int nt, int t0, int r16, int q) {
    // The entries come from the fixed-stride copy (small_prepare): their addresses do not wait for the row pointers -- one round trip
    // to the L2 at the head of the kernel instead of two.  Entries past a row's end are stored as column 0 with zero values, so the
    // gather needs no per-entry predicate (straight-line code: all LDS operations of a tile issue back to back).
#pragma unroll
    for (int k = 0; k < MAXT; ++k) {
        const int t = t0 + WAVES * k, r = 16 * t + r16;
        const bool valid = t < nt && r < E;
        op.cnt[k] = valid ? rowptr[r + 1] - rowptr[r] : 0;
        if constexpr (OVF) op.ovs[k] = (ovf_ptr && valid) ? ovf_ptr[r] : 0;      // (null: the list is not in LDS, the tails are read from memory)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const float4 e = ell[(size_t)(valid ? r : 0) * SM_CH + q + 4 * s];
            op.enc[k][s] = valid ? e.x : 0.f;
            op.v0[k][s] = valid ? e.y : 0.f;
            op.v1[k][s] = valid ? e.z : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < MAXT; ++k) {
        int m = op.cnt[k] < SM_CH ? op.cnt[k] : SM_CH;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(m, o, 64); m = y > m ? y : m; }
        op.cmax[k] = __builtin_amdgcn_readfirstlane(m);
    }
}
// lo += sum_u val_lower[u] * H[col_u][quad q], up likewise, over the entries of tile k's row
template <int MAXT, int K, bool OVF>
__device__ __forceinline__ void sm_gather(const SmOp<MAXT>& op, const int32_t* __restrict__ rowptr, const float4* __restrict__ ent,
                                          const float* ovf, const float* in, int r, int r16, int q, f32x4& lo, f32x4& up) {
    // Two halves of six entries, each: all eighteen exchanges, then all six reads, then the FMAs -- two LDS latencies per half instead of
    // two per entry (a wave's tile is a latency chain: 2400 of its 4900 cycles were this gather while it went entry by entry).
#ifndef SM_GB
#define SM_GB 6
#endif
#pragma unroll
    for (int u0 = 0; u0 < SM_CH; u0 += SM_GB) {
        if (u0 < op.cmax[K]) {                                  // wave-uniform: every lane takes part in the exchange
            int c[SM_GB];
            float av[SM_GB], bv[SM_GB];
            f32x4 d[SM_GB];
#pragma unroll
            for (int u = 0; u < SM_GB; ++u) {
                const int owner = r16 + 16 * ((u0 + u) & 3), sl = (u0 + u) >> 2;
                c[u] = __float_as_int(__shfl(op.enc[K][sl], owner, 64));
                av[u] = __shfl(op.v0[K][sl], owner, 64);
                bv[u] = __shfl(op.v1[K][sl], owner, 64);
            }
#pragma unroll
            for (int u = 0; u < SM_GB; ++u) d[u] = *(const f32x4*)((const char*)in + ((uint32_t)c[u] ^ (uint32_t)(16 * q)));
#pragma unroll
            for (int u = 0; u < SM_GB; ++u) {
                lo += av[u] * d[u];
                up += bv[u] * d[u];
            }
        }
    }
    // Rows longer than the register copy: a quarter of the rows of the reference's own complexes (|E| = 1001: mean 11.2 entries, 236
    // rows of 13 .. 20; the Ebli pair L1, L1^2 has many more) and nearly every tile has one.  Their tails come from the overflow list
    // in LDS.  Read from memory -- the first form, kept for lists the LDS has no room for -- they are loads in the middle of the
    // tile's chain whose s_waitcnt also waits for every OLDER store of the wave (one counter): harmless while those are plain stores,
    // but in the paired form the previous tile's agent-scope stores are acknowledged by the memory side, ~2 us later.
    if (op.cnt[K] > SM_CH) {
        if (OVF && ovf) {
            const float* e = ovf + 3 * op.ovs[K];
            for (int j = SM_CH; j < op.cnt[K]; ++j, e += 3) {
                const f32x4 d = *(const f32x4*)((const char*)in + ((uint32_t)__float_as_int(e[0]) ^ (uint32_t)(16 * q)));
                lo += e[1] * d;
                up += e[2] * d;
            }
        } else {
            const int j0 = rowptr[r];
            for (int j = j0 + SM_CH; j < j0 + op.cnt[K]; ++j) {
                const float4 e = ent[j];
                const f32x4 d = *(const f32x4*)(in + sm_at(__float_as_int(e.x), q));
                lo += e.y * d;
                up += e.z * d;
            }
        }
    }
}
// (S_lo x)[row], (S_up x)[row] of tile k's row from the staged flow: each lane its own entries, the row's four lanes add up
template <int MAXT, int K, bool OVF>
__device__ __forceinline__ void sm_shift_x(const SmOp<MAXT>& op, const int32_t* __restrict__ rowptr, const float4* __restrict__ ent,
                                           const float* ovf, const float* xs, int r, int q, float& lo, float& up) {
    lo = 0.f; up = 0.f;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const float xv = xs[__float_as_int(op.enc[K][s]) >> 6];
        lo = fmaf(op.v0[K][s], xv, lo);
        up = fmaf(op.v1[K][s], xv, up);
    }
    if (q == 0 && op.cnt[K] > SM_CH) {
        if (OVF && ovf) {
            const float* e = ovf + 3 * op.ovs[K];
            for (int j = SM_CH; j < op.cnt[K]; ++j, e += 3) {
                const float xv = xs[__float_as_int(e[0]) >> 6];
                lo = fmaf(e[1], xv, lo);
                up = fmaf(e[2], xv, up);
            }
        } else {
            const int j0 = rowptr[r];
            for (int j = j0 + SM_CH; j < j0 + op.cnt[K]; ++j) {
                const float4 e = ent[j];
                const float xv = xs[__float_as_int(e.x)];
                lo = fmaf(e.y, xv, lo);
                up = fmaf(e.z, xv, up);
            }
        }
    }
    lo += __shfl_xor(lo, 16, 64); lo += __shfl_xor(lo, 32, 64);
    up += __shfl_xor(up, 16, 64); up += __shfl_xor(up, 32, 64);
}

template <typename F, int... Ks>
__device__ __forceinline__ void sm_tiles(F&& f, std::integer_sequence<int, Ks...>) { (f(std::integral_constant<int, Ks>{}), ...); }

// Hand-over between the two workgroups of a trajectory (paired form): agent-scope accesses (sc1), which neither hit a stale line of
// this CU's L1 nor stay behind in it.  Sixteen bytes per instruction (as four relaxed atomics of a dword each -- what the language
// offers -- the hand-over cost 22 us of a 100 us step instead of ~10), so inline assembly, which the compiler's counters do not see:
// the stores are waited for explicitly before the barrier that precedes the flag, and a load's result is not touched (nor its
// register copied: the load is unconditional, its use goes through sm_landed) before the s_waitcnt of sm_landed.
__device__ __forceinline__ void sm_store_sc1(float* p, f32x4 v) {
#ifdef SM_AB_PLAIN_STORE
    *(f32x4*)p = v;
    return;
#endif
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 sm_load_sc1(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void sm_landed(f32x4 (&v)[N]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]) : : "memory");
#pragma unroll
    for (int k = 1; k < N; ++k) asm volatile("" : "+v"(v[k]) : : "memory");
}
#ifndef SM_SPIN_LIMIT
#define SM_SPIN_LIMIT 400000
#endif

// PAIRED: two workgroups per trajectory.  The tiles go to them in alternating blocks of WAVES tiles (workgroup `half` takes the tiles
// wave + WAVES (2 k + half)), each keeps the FULL activation buffers in its LDS and the operator rows of its own tiles in registers,
// and after every layer (forward: H_l, backward: dx_l) the two hand each other their rows through memory: the rows are stored with
// agent scope, a flag per (phase, half) is raised once the workgroup's stores have been acknowledged, the partner waits for it and
// loads the rows into its own LDS.  The readout (a few hundred items) is done by both.  The two workgroups of a pair are
// blockIdx b and b + 8 -- the same XCD, one L2 -- and the host takes this form only when all workgroups of the launch are resident
// together (2 N <= CUs; a workgroup's LDS leaves room for one per CU), so the wait below always ends; it is bounded all the same
// (SM_SPIN_LIMIT polls, ~0.3 s): a partner that never arrives turns the trajectory's loss into NaN instead of hanging the device.
template <int MAXT, int WAVES, bool PAIRED>
__global__ __launch_bounds__(64 * WAVES) void small_step_kernel(SmallArgs a) {
    constexpr int SM_THREADS = 64 * WAVES, SM_WAVES = WAVES;
    constexpr int TS = PAIRED ? 2 * WAVES : WAVES;              // tile stride of a wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int E = a.n_edges, L = a.n_layers, act = a.act;
    const SmallLds lay = small_lds(E, a.n_ovf > a.n_ovf_t ? a.n_ovf : a.n_ovf_t);
    const int epad = lay.epad, nt = epad >> 4;
    float* const lds = (float*)smem;                            // (buffers are addressed as lds + offset: the compiler keeps LDS instructions)
    const int bufsz = epad * SM_C;                              // floats per activation buffer: A at 0, B at bufsz
    float* Wl = (float*)(smem + lay.off_w);
    float* misc = (float*)(smem + lay.off_misc);
    float* lgs = misc;                       // [64]
    float* dls = lgs + 64;                   // [64]
    float* wlast = dls + 64;                 // [16]
    int* d_ptr = (int*)(wlast + 16);         // [80]
    int* it_e = d_ptr + 80;                  // [512]
    float* it_s = (float*)(it_e + SM_ITEMS); // [512]
    int* it_d = (int*)(it_s + SM_ITEMS);     // [512]
    float* bh = (float*)(it_d + SM_ITEMS);   // [64][16]
    float* dwf_red = bh + 64 * SM_C;         // [waves][48]
    float* xs = (float*)(smem + lay.off_x);  // [epad] the input flow of this trajectory
    float2* ysl = (float2*)(smem + lay.off_y);  // [epad] (S_lo x, S_up x), when there is room (lay.y_lds)
    float* const ovf_l = (float*)(smem + lay.off_ovf);          // [n_ovf][3] the operator's overflow list, when there is room (lay.ovf_lds)
    // (the twelve-wave kernels of |E| <= 384 have no register to spare for it and measured no gain, small_lds; the nine-tile instance
    //  of |E| > 1024 has neither the registers nor the LDS)
    constexpr bool OVF = WAVES == 8 && (PAIRED || MAXT <= 8);
    const float* const ovf = OVF && lay.ovf_lds ? ovf_l : nullptr;
    const int half = PAIRED ? (blockIdx.x >> 3) & 1 : 0;
    const int n = PAIRED ? (blockIdx.x >> 4) * 8 + (blockIdx.x & 7) : blockIdx.x;
    if (n >= a.n_traj) return;                                  // (paired: the grid is padded to whole groups of sixteen)
    const int s = n / a.ns, i = n - s * a.ns;
    const int N = a.n_traj;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int t0 = wave + (PAIRED ? WAVES * half : 0);          // first tile of this wave
    const int PW = small_pw(L);
    const int my_p = PAIRED ? 2 * n + half : n;
    float* my_partial = a.partial + (size_t)my_p * PW;
    int* const fail = d_ptr + 79;                               // (the list of slots ends at d_ptr[max_deg <= 64])
    int* const flags_n = PAIRED ? a.flags + (size_t)n * (4 * SM_MAX_LAYERS) : nullptr;
    float* const xg_n = PAIRED ? a.xg + (size_t)n * E * SM_C : nullptr;
    if (PAIRED && tid == 0) *fail = 0;
    if (a.step && blockIdx.x == 0 && tid == 0) a.step[1] = a.step[0];      // (small_reduce_kernel applies update step[1] and advances step[0])
    // phase p of this half is in memory: called by every thread after a barrier that followed the wave's s_waitcnt vmcnt(0)
    auto post = [&](int phase) {
#ifdef SM_AB_DROP_POST                                           // (diagnostic build, tools/small_pair_timeout.sh: a partner that never arrives)
        if (half == 1 && phase == 1) return;
#endif
        if (PAIRED && tid == 0) __hip_atomic_store(flags_n + 2 * phase + half, a.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // wait for the partner's phase p and bring its rows (blocks 2 k + 1 - half of 16 WAVES rows) from `src` into `out`
    auto collect = [&](float* out, const float* src, int phase) {
        if constexpr (PAIRED) {
#ifdef SM_AB_NO_COLLECT
            return;
#endif
            if (tid == 0) {
                const int* f = flags_n + 2 * phase + (1 - half);
                int spins = 0;
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.tag) {
                    if (++spins > SM_SPIN_LIMIT) { *fail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            f32x4 v[MAXT];
            const int qq = tid & 3;
#pragma unroll
            for (int k = 0; k < MAXT; ++k) {
                const int r = (2 * k + 1 - half) * (16 * WAVES) + (tid >> 2);
                v[k] = sm_load_sc1(src + (size_t)(r < E ? r : 0) * SM_C + 4 * qq);
            }
            sm_landed(v);
#pragma unroll
            for (int k = 0; k < MAXT; ++k) {
                const int r = (2 * k + 1 - half) * (16 * WAVES) + (tid >> 2);
                if (r < epad) *(f32x4*)(out + sm_at(r, qq)) = r < E ? v[k] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __syncthreads();
        }
    };
    SM_STAMP(0);
#ifdef SCN_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0) g_small_stamps[14] = clock64();
#endif

    // ---------------- the readout's tables of this trajectory (wave 0) are a chain of four dependent loads: last node -> neighbour ->
    // incidence range -> incident edges.  One link per phase, each issued where the previous one has long landed, so that no barrier of
    // the forward waits for a round trip of the chain (done in one go at the top it held the first barrier back by ~5 us).
    int ro_v = -1, ro_start = 0, ro_cnt = 0, ro_e[SM_RO_PRE];
    float ro_s[SM_RO_PRE], ro_y = 0.f;
#pragma unroll
    for (int j = 0; j < SM_RO_PRE; ++j) { ro_e[j] = 0; ro_s[j] = 0.f; }
    int ro_vlast = 0;
    if (wave == 0 && lane < a.max_deg) {
        ro_vlast = a.last_nodes[n];
        ro_y = a.y[(size_t)n * a.max_deg + lane];
    }
    SmOp<MAXT> op;
    sm_load_op<MAXT, TS, OVF>(op, a.rowptr, a.ell, lay.ovf_lds ? a.ovf_ptr : nullptr, E, nt, t0, r16, q);
    if (lay.ovf_lds)
        for (int j = tid; j < a.n_ovf; j += SM_THREADS) {
            const float4 e = a.ovf[j];
            ovf_l[3 * j] = e.x; ovf_l[3 * j + 1] = e.y; ovf_l[3 * j + 2] = e.z;
        }

    // A layer's three weight matrices are requested one layer AHEAD into registers and written to LDS when their layer starts: read
    // where they are needed they were a round trip to the L2 between two barriers, ~2 us of every layer in both directions.
    // (The twelve-wave kernels of |E| <= 384 run at 168 of their 170 registers: they read the weights where they need them.)
    constexpr bool W_AHEAD = WAVES == 8 && (PAIRED || MAXT <= 8);     // (nor has the nine-tile instance: profiles/r05_small_pair_ab.txt)
    constexpr int WPT = (SM_LAYER_W + SM_THREADS - 1) / SM_THREADS;
    float wpre[WPT];
    int w_layer = 1;
    auto request_w = [&](int li) {
        w_layer = li;
        if constexpr (W_AHEAD) {
#pragma unroll
            for (int j = 0; j < WPT; ++j) {
                const int o = tid + j * SM_THREADS;
                wpre[j] = o < SM_LAYER_W ? a.W[3 * li + o / (SM_C * SM_C)][o % (SM_C * SM_C)] : 0.f;
            }
        }
    };
    auto place_w = [&]() {
        if constexpr (W_AHEAD) {
#pragma unroll
            for (int j = 0; j < WPT; ++j) {
                const int o = tid + j * SM_THREADS;
                if (o < SM_LAYER_W) Wl[o] = wpre[j];
            }
        } else {
            for (int o = tid; o < SM_LAYER_W; o += SM_THREADS) Wl[o] = a.W[3 * w_layer + o / (SM_C * SM_C)][o % (SM_C * SM_C)];
        }
    };
    request_w(1);

    // ---------------- layer 1 (one input channel): y = (x, S_lo x, S_up x), H_1 = act(y . W_1)
    for (int e = tid; e < epad; e += SM_THREADS) xs[e] = e < E ? a.x[((size_t)s * E + e) * a.ns + i] : 0.f;
    if (tid < 3 * SM_C) Wl[tid] = a.W[tid / SM_C][tid % SM_C];
    if (tid < SM_C) wlast[tid] = a.W[3 * L][tid];
    __syncthreads();
    if (wave == 0 && lane < a.max_deg) ro_v = a.nbr[(size_t)ro_vlast * a.max_deg + lane];
    SM_STAMP(1);
    float4* ysn = (float4*)a.ys + (size_t)n * E;
    float* hs_n = a.hs + (size_t)n * E * SM_C;                  // + layer * N * E * 16
    const size_t hs_layer = (size_t)N * E * SM_C;
    {
        float* out = lds;                                        // buffer A
        f32x4 w0, w1, w2;
#pragma unroll
        for (int j = 0; j < 4; ++j) { w0[j] = Wl[4 * q + j]; w1[j] = Wl[SM_C + 4 * q + j]; w2[j] = Wl[2 * SM_C + 4 * q + j]; }
        auto tile = [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            int t = t0 + TS * K;
            asm volatile("" : "+s"(t));                          // (per-tile addresses are recomputed, not kept in registers across the layers)
            if (t >= nt) return;
            const int r = 16 * t + r16;
            const bool valid = r < E;
            float lo, up;
            sm_shift_x<MAXT, K, OVF>(op, a.rowptr, a.ent, ovf, xs, r, q, lo, up);
            const float x0 = xs[r];
            f32x4 o;
            o = sm_act4(act, x0 * w0 + lo * w1 + up * w2);
            if (!valid) o = f32x4{0.f, 0.f, 0.f, 0.f};
            *(f32x4*)(out + sm_at(r, q)) = o;
            if (lay.y_lds && q == 0) ysl[r] = make_float2(lo, up);
            if (valid) {
                if constexpr (PAIRED) sm_store_sc1(hs_n + (size_t)r * SM_C + 4 * q, o);
                else *(f32x4*)(hs_n + (size_t)r * SM_C + 4 * q) = o;
                if (!lay.y_lds && !a.same_t && q == 0) ysn[r] = make_float4(x0, lo, up, 0.f);
            }
        };
        sm_tiles(tile, std::make_integer_sequence<int, MAXT>{});
    }
    if constexpr (PAIRED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    post(0);
    collect(lds, hs_n, 0);
    if (wave == 0 && ro_v >= 0) {
        ro_start = a.inc_ptr[ro_v];
        ro_cnt = a.inc_ptr[ro_v + 1] - ro_start;
    }
    SM_STAMP(3);

    // ---------------- layers 2 .. L: out = act(H W_0 + (S_lo H) W_1 + (S_up H) W_2)
    int in_o = 0, out_o = bufsz;
    for (int li = 1; li < L; ++li) {
        const float* in = lds + in_o;
        float* out = lds + out_o;
        place_w();
        __syncthreads();
        if (li + 1 < L) request_w(li + 1);                       // (the last layer's stay in LDS for the backward's first)
        float wa[3][4];                                          // A[m = c_out = r16][k = (s, q)] = W_g[4 q + s][r16]
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) wa[g][u] = Wl[g * 256 + (4 * q + u) * SM_C + r16];
        auto tile = [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            int t = t0 + TS * K;
            asm volatile("" : "+s"(t));                          // (per-tile addresses are recomputed, not kept in registers across the layers)
            if (t >= nt) return;
            const int r = 16 * t + r16;
            const bool valid = r < E;
            SM_CYC(0, K == 1 && li == 1);
            f32x4 zs = *(const f32x4*)(in + sm_at(r, q)), zl = {0.f, 0.f, 0.f, 0.f}, zu = zl;     // (rows past the end hold zeros)
            sm_gather<MAXT, K, OVF>(op, a.rowptr, a.ent, ovf, in, r, r16, q, zl, zu);
            SM_CYC(1, K == 1 && li == 1);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acl = acc, acu = acc;   // three independent chains (a dependent MFMA waits ~40 cycles)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0][u], zs[u], acc, 0, 0, 0);
                acl = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1][u], zl[u], acl, 0, 0, 0);
                acu = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[2][u], zu[u], acu, 0, 0, 0);
            }
            acc += acl + acu;
            f32x4 o;
            SM_CYC(2, K == 1 && li == 1);
            o = sm_act4(act, acc);
            if (!valid) o = f32x4{0.f, 0.f, 0.f, 0.f};
            SM_CYC(3, K == 1 && li == 1);
            *(f32x4*)(out + sm_at(r, q)) = o;
            if constexpr (PAIRED) {
                if (valid) sm_store_sc1((li < L - 1 ? hs_n + li * hs_layer : xg_n) + (size_t)r * SM_C + 4 * q, o);
            } else {
                if (valid && li < L - 1) *(f32x4*)(hs_n + li * hs_layer + (size_t)r * SM_C + 4 * q) = o;
            }
            SM_CYC(4, K == 1 && li == 1);
            SM_CYC(5, K == 2 && li == 1);
        };
        sm_tiles(tile, std::make_integer_sequence<int, MAXT>{});
        if constexpr (PAIRED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        post(li);
        collect(out, li < L - 1 ? hs_n + li * hs_layer : xg_n, li);
        if (li == 1 && wave == 0) {
#pragma unroll
            for (int j = 0; j < SM_RO_PRE; ++j)
                if (j < ro_cnt) {
                    ro_e[j] = a.inc_edge[ro_start + j];
                    ro_s[j] = a.inc_sign[ro_start + j];
                }
        }
        SM_STAMP(3 + li);
        const int tmp = in_o; in_o = out_o; out_o = tmp;
    }
    const float* H = lds + in_o;                                 // H_L
    float* dz = lds + out_o;                                     // free buffer -> readout gradient

    // ---------------- readout (TE:151-152, 298-303), cross-entropy (STM:54) and their gradient.  Wave 0 lists the trajectory's items
    // (neighbour slot, incident edge, sign), every 16-lane group of the workgroup then takes slots (logits) and later items (scatter);
    // the softmax in between is one wave's work.
    for (int idx = tid; idx < epad * 4; idx += SM_THREADS) *(f32x4*)(dz + (size_t)idx * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    bool overflow = false;
    if (wave == 0) {
        const int off = sm_excl_scan(ro_cnt, lane);
        int total = __shfl(off + ro_cnt, 63, 64);
        overflow = total > SM_ITEMS;                             // (the host checks the bound; never index past the list)
        if (overflow) total = 0;
        if (lane < a.max_deg) d_ptr[lane] = overflow ? 0 : off;
        if (lane == 0) d_ptr[a.max_deg] = total;
        if (!overflow) {
#pragma unroll
            for (int j = 0; j < SM_RO_PRE; ++j)
                if (j < ro_cnt) {
                    it_e[off + j] = ro_e[j];
                    it_s[off + j] = ro_s[j];
                    it_d[off + j] = lane;
                }
            for (int j = SM_RO_PRE; j < ro_cnt; ++j) {
                it_e[off + j] = a.inc_edge[ro_start + j];
                it_s[off + j] = a.inc_sign[ro_start + j];
                it_d[off + j] = lane;
            }
        }
    }
    __syncthreads();
    const int grp = wave * 4 + (lane >> 4), n_grp = 4 * SM_WAVES, cc = lane & 15;      // 16-lane groups of the workgroup, lane = channel
    const float wc = wlast[cc];
    for (int d = grp; d < a.max_deg; d += n_grp) {
        const int t0 = d_ptr[d], t1 = d_ptr[d + 1];
        float acc = 0.f;
        for (int t = t0; t < t1; ++t) acc = fmaf(it_s[t], H[sm_at1(it_e[t], cc)], acc);
        bh[d * SM_C + cc] = acc;
        float lg = acc * wc;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) lg += __shfl_xor(lg, o, 64);
        if (cc == 0) lgs[d] = lg;
    }
    __syncthreads();
    if (wave == 0) {
        const int g = lane >> 4;
        const bool live = lane < a.max_deg;
        const float logit = live ? lgs[lane] : 0.f;
        const float xm = live ? logit : -INFINITY;
        const float m = sm_wave_max(xm);
        const float se = sm_wave_sum(live ? expf(xm - m) : 0.f);
        const float lp = logit - (m + logf(se));
        const float gy = live ? ro_y * a.scale : 0.f;            // d loss / d logp
        double lpart = live ? (double)lp * (double)gy : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lpart += __shfl_xor(lpart, o, 64);
        if (lane == 0) a.loss_part[my_p] = overflow ? (double)NAN : (half == 0 ? lpart : 0.0);    // (paired: both halves do the readout)
        const float gs = sm_wave_sum(gy);
        if (live) dls[lane] = gy - expf(lp) * gs;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // one wave: its LDS operations complete in order
        // d w_last[c] = sum_d dl[d] * bh[d][c]: lane (g, c) sums the slots d = g, g + 4, ..; the four groups combine in a fixed order
        float dwl = 0.f;
        for (int d = g; d < a.max_deg; d += 4) dwl = fmaf(dls[d], bh[d * SM_C + cc], dwl);
        dwl += __shfl_xor(dwl, 16, 64);
        dwl += __shfl_xor(dwl, 32, 64);
        if (lane < SM_C) my_partial[PW - SM_C + lane] = half == 0 ? dwl : 0.f;
    }
    __syncthreads();
    SM_STAMP(9);
    {                                                            // dz_L[e][c] += sign * dl[d] * w_last[c] * act'(H_L[e][c]): an edge has two
        const int total = d_ptr[a.max_deg];                      // endpoints, so at most two addends meet in an entry -- the sum does not depend
        const uint32_t dz_addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)dz;     // on their order.  (The factor
        for (int t = grp; t < total; t += n_grp) {               // act' rides on the addends: a pass over the whole buffer for the few
            const int at = sm_at1(it_e[t], cc);                  // hundred entries that are not zero was 1.5 us of the kernel.)
            const float v = it_s[t] * dls[it_d[t]] * wc * act_grad_from_output(act, H[at]);
            asm volatile("ds_add_f32 %0, %1" : : "v"(dz_addr + (uint32_t)at * 4u), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // (the saved activations hs / ys are read back from here on: every store to them has been acknowledged by the L2 at a barrier since --
    //  __syncthreads() waits for the wave's vector-memory counter -- and none of their lines has been loaded into this CU's L1 before)
    if (!a.same_t) {                                            // the backward gathers through the transpose (the forward's last use of the
        sm_load_op<MAXT, TS, OVF>(op, a.rowptr_t, a.ell_t, lay.ovf_lds ? a.ovf_ptr_t : nullptr, E, nt, t0, r16, q);          // overflow list is barriers back)
        if (lay.ovf_lds)
            for (int j = tid; j < a.n_ovf_t; j += SM_THREADS) {
                const float4 e = a.ovf_t[j];
                ovf_l[3 * j] = e.x; ovf_l[3 * j + 1] = e.y; ovf_l[3 * j + 2] = e.z;
            }
    }
    const float4* ent_b = a.same_t ? a.ent : a.ent_t;
    const int32_t* rowptr_b = a.same_t ? a.rowptr : a.rowptr_t;
    __syncthreads();
    SM_STAMP(10);

    // ---------------- backward of layers L .. 2: G = (dz, S_lo^T dz, S_up^T dz), dx = (sum_g G_g W_g^T) act'(H_{l-1}), dW_g = H_{l-1}^T G_g
    { const int tmp = in_o; in_o = out_o; out_o = tmp; }       // in = dz_L, out = the buffer H_L is done with
    const bool red_overlay = (size_t)epad * SM_C >= (size_t)SM_WAVES * SM_LAYER_W;
    float idm[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) idm[u] = (4 * q + u == r16) ? 1.f : 0.f;
    for (int li = L - 1; li >= 1; --li) {
        const float* in = lds + in_o;
        float* out = lds + out_o;
        // H_li, the input of layer li + 1, comes back into the OUTPUT buffer in one sweep (every load independent): a tile reads its rows'
        // values there -- as the act' factor and, rows along K, as the A operand of the weight gradient -- before it writes dx over them
        {
            const float* aux = hs_n + (size_t)(li - 1) * hs_layer;
            for (int idx = tid; idx < epad * 4; idx += SM_THREADS) {
                const int r = idx >> 2, qq = idx & 3;
                if (PAIRED && ((idx / SM_THREADS) & 1) != half) continue;       // (the tiles read aux at their own rows only)
                const f32x4 v = r < E ? *(const f32x4*)(aux + (size_t)r * SM_C + 4 * qq) : f32x4{0.f, 0.f, 0.f, 0.f};
                *(f32x4*)(out + sm_at(r, qq)) = v;
            }
        }
        if (li < L - 1) place_w();
        __syncthreads();
        if (li > 1) request_w(li - 1);
        f32x4 wb[3];                                             // A[m = c = r16][k = (s, q)] = W_g[r16][4 q + s]
#pragma unroll
        for (int g = 0; g < 3; ++g) wb[g] = *(const f32x4*)(Wl + g * 256 + r16 * SM_C + 4 * q);
        f32x4 dWacc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) dWacc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        float dwf[3][4];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) dwf[g][j] = 0.f;
        auto tile = [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            int t = t0 + TS * K;
            asm volatile("" : "+s"(t));                          // (per-tile addresses are recomputed, not kept in registers across the layers)
            if (t >= nt) return;
            const int r = 16 * t + r16;
            const bool valid = r < E;
            const f32x4 auxq = *(const f32x4*)(out + sm_at(r, q));
            float auxT[4], g0T[4];                               // rows 16 t + 4 q + u of channel r16: aux, and dz itself (segment 0)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                auxT[u] = out[sm_at1(16 * t + 4 * q + u, r16)];
                g0T[u] = in[sm_at1(16 * t + 4 * q + u, r16)];
            }
            f32x4 G[3];
            G[0] = *(const f32x4*)(in + sm_at(r, q));
            G[1] = G[2] = f32x4{0.f, 0.f, 0.f, 0.f};
            sm_gather<MAXT, K, OVF>(op, rowptr_b, ent_b, ovf, in, r, r16, q, G[1], G[2]);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acl = acc, acu = acc;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[0][u], G[0][u], acc, 0, 0, 0);
                acl = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[1][u], G[1][u], acl, 0, 0, 0);
                acu = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[2][u], G[2][u], acu, 0, 0, 0);
            }
            acc += acl + acu;
            f32x4 dx;
#pragma unroll
            for (int j = 0; j < 4; ++j) dx[j] = valid ? acc[j] * act_grad_from_output(act, auxq[j]) : 0.f;
            if (li > 1) {
                asm volatile("" ::: "memory");                   // (the reads of this tile's aux rows above stay above)
                *(f32x4*)(out + sm_at(r, q)) = dx;
                if constexpr (PAIRED)
                    if (valid) sm_store_sc1(xg_n + (size_t)(L - li) * hs_layer + (size_t)r * SM_C + 4 * q, dx);
            } else {                                             // dW_1[g][c] += y[row][g] * dx[row][c]
                float y0, y1, y2;
                if (lay.y_lds) {
                    const float2 lu = ysl[r];
                    y0 = xs[r]; y1 = lu.x; y2 = lu.y;
                } else if (a.same_t) {
                    sm_shift_x<MAXT, K, OVF>(op, rowptr_b, ent_b, ovf, xs, r, q, y1, y2);
                    y0 = xs[r];
                } else {
                    const float4 yv = valid ? ysn[r] : make_float4(0.f, 0.f, 0.f, 0.f);
                    y0 = yv.x; y1 = yv.y; y2 = yv.z;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dwf[0][j] = fmaf(y0, dx[j], dwf[0][j]);
                    dwf[1][j] = fmaf(y1, dx[j], dwf[1][j]);
                    dwf[2][j] = fmaf(y2, dx[j], dwf[2][j]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) dWacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(auxT[u], g0T[u], dWacc[0], 0, 0, 0);
            f32x4 Gt1 = {0.f, 0.f, 0.f, 0.f}, Gt2 = Gt1;         // G_g with the rows along K: lane holds rows 4 q + j of channel r16
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                Gt1 = __builtin_amdgcn_mfma_f32_16x16x4f32(G[1][u], idm[u], Gt1, 0, 0, 0);
                Gt2 = __builtin_amdgcn_mfma_f32_16x16x4f32(G[2][u], idm[u], Gt2, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                dWacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(auxT[u], Gt1[u], dWacc[1], 0, 0, 0);
                dWacc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(auxT[u], Gt2[u], dWacc[2], 0, 0, 0);
            }
        };
        sm_tiles(tile, std::make_integer_sequence<int, MAXT>{});
        if constexpr (PAIRED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                         // every wave is done with `in`
        if (li > 1) post(2 * L - 1 - li);                        // (the partner's wait overlaps the reduction below)
        float* red = red_overlay ? lds + in_o : (float*)(smem + lay.off_red);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[wave * SM_LAYER_W + g * 256 + (4 * q + j) * SM_C + r16] = dWacc[g][j];
        if (li == 1) {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = dwf[g][j];
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
                    if (r16 == 0) dwf_red[wave * 48 + g * SM_C + 4 * q + j] = v;
                }
        }
        __syncthreads();
        for (int o = tid; o < SM_LAYER_W; o += SM_THREADS) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < SM_WAVES; ++w) sum += red[w * SM_LAYER_W + o];
            my_partial[3 * SM_C + (li - 1) * SM_LAYER_W + o] = sum;
        }
        if (li == 1 && tid < 48) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < SM_WAVES; ++w) sum += dwf_red[w * 48 + tid];
            my_partial[tid] = sum;
        }
        if (li > 1) collect(out, xg_n + (size_t)(L - li) * hs_layer, 2 * L - 1 - li);
        __syncthreads();
        SM_STAMP(10 + (L - li));
        const int tmp = in_o; in_o = out_o; out_o = tmp;
    }
    if (PAIRED && *fail) {                                       // the partner never arrived (every hand-over is barriers back): neither the
        if (tid == 0) a.loss_part[my_p] = (double)NAN;           // loss nor the gradients of this launch may pass for results
        for (int o = tid; o < PW; o += SM_THREADS) my_partial[o] = NAN;
    }
#ifdef SCN_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0) g_small_stamps[15] = clock64();
#endif
}

// dW[k] (+)= sum_n partial[n][...] (trajectories in a fixed order), loss (+)= sum_n loss_part[n].  One block per 16 outputs: thread
// (output, group g of 16) sums trajectories g, g + 16, .. with every load independent, the groups combine in order.
struct SmallReduce {
    int32_t n_traj, pw, n_w, overwrite;
    int32_t off[3 * SM_MAX_LAYERS + 2];
    float* dW[3 * SM_MAX_LAYERS + 1];
    const float* partial;
    const double* loss_part;
    double* loss;
    int* flags;                                       // paired form: the hand-over flags go back to zero for the next launch / replay
    int32_t n_flags;
    // scn_small_step_adam: the optimiser step on the summed gradient by the thread that has it (w null: none)
    float* w; float* m; float* v;                     // flat, in the order of the outputs o
    float lr, b1, b2, eps, wd2, g_scale;
    int32_t* step;
};
__global__ __launch_bounds__(256) void small_reduce_kernel(SmallReduce a) {
    __shared__ float part[256];
    for (int f = blockIdx.x * 256 + threadIdx.x; f < a.n_flags; f += gridDim.x * 256) a.flags[f] = 0;
    const int oo = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + oo;
    float acc = 0.f;
    if (o < a.pw)
        for (int n = g; n < a.n_traj; n += 16) acc += a.partial[(size_t)n * a.pw + o];
    part[g * 16 + oo] = acc;
    __shared__ float cc[2];
    if (a.w && threadIdx.x == 0) adam_corrections(a.b1, a.b2, a.step[1], cc[0], cc[1]);
    __syncthreads();
    if (g == 0 && o < a.pw) {
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += part[k * 16 + oo];
        int k = 0;
        while (k + 1 < a.n_w && o >= a.off[k + 1]) ++k;
        float* d = a.dW[k] + (o - a.off[k]);
        const float grad = a.overwrite ? sum : *d + sum;
        *d = grad;
        if (a.w) adam_update(a.w + o, grad, a.m + o, a.v + o, a.lr, a.b1, a.b2, a.eps, cc[0], cc[1], a.wd2, a.g_scale);
    }
    if (a.w && blockIdx.x == 0 && threadIdx.x == 0) a.step[0] = a.step[1] + 1;      // (step[1] was set by small_step_kernel; nobody reads [0] here)
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        double accd = 0.0;
        for (int n = threadIdx.x; n < a.n_traj; n += 64) accd += a.loss_part[n];
#pragma unroll
        for (int w = 32; w > 0; w >>= 1) accd += __shfl_xor(accd, w, 64);
        if (threadIdx.x == 0) a.loss[0] = a.overwrite ? accd : a.loss[0] + accd;
    }
}

static bool small_shape(const scn_conv_s* c) {
    return c && c->n_groups == 1 && c->g[0].identity == 1 && c->g[0].n_vals == 2 && c->g[0].n_cols == c->n_rows;
}

template <int T, int WV, bool PAIRED = false>
static int small_raise_lds(size_t bytes) {
    SCN_HIP_TRY(hipFuncSetAttribute((const void*)small_step_kernel<T, WV, PAIRED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return SCN_OK;
}

// The paired form: blocks of eight tiles alternate between the two workgroups, the first takes ceil(blocks / 2) of them.
static inline int small_blocks(int n_edges) { return (((n_edges + 15) >> 4) + 7) / 8; }
static std::atomic<int> g_small_pairing{0};          // 0: when it applies, 1: never (scn_small_step_pairing)
static std::atomic<int> g_small_tag{0};
static int small_cus() {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) cus = v;
        else cus = 0;
    }
    return cus;
}
static bool small_paired(int n_edges, int n_traj) {
    return g_small_pairing.load() == 0 && small_waves(n_edges) == 8 && small_blocks(n_edges) >= 2 && 2 * n_traj <= small_cus();
}

// Called by scn_conv_create* (scn_conv.hip) on every new handle: an operator scn_small_step can serve gets its entry pack
// (col, val_lower, val_upper, 0) on the device HERE, at create time, and the kernel instance of its size gets its LDS limit raised --
// so that scn_small_step itself neither allocates nor copies nor synchronises, like every other launch of the library (its first call
// on a fresh handle can be captured into a HIP graph).  Other operators: nothing to do.
int small_prepare(scn_conv_s* c) {
    if (!small_shape(c) || c->n_rows > 16 * 8 * SM_MAXT) return SCN_OK;
    const SmallLds lay = small_lds(c->n_rows, 0);
    if (lay.total > 160 * 1024) return SCN_OK;
    const Group& G = c->g[0];
    std::vector<float4> h((size_t)std::max<int64_t>(G.nnz, 1));
    for (int64_t j = 0; j < G.nnz; ++j) {
        int32_t col = G.h_col[j];
        float cf;
        std::memcpy(&cf, &col, 4);
        h[j] = make_float4(cf, G.h_val0[j], G.h_val1[j], 0.f);
    }
    void* d = nullptr;
    SCN_HIP_TRY(hipMalloc(&d, h.size() * sizeof(float4)));
    c->plan.allocs.push_back(d);
    SCN_HIP_TRY(hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
    c->small_pack = d;
    std::vector<float4> ell((size_t)c->n_rows * SM_CH, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int32_t r = 0; r < c->n_rows; ++r)
        for (int64_t j = G.h_rowptr[r]; j < G.h_rowptr[r + 1] && j < (int64_t)G.h_rowptr[r] + SM_CH; ++j) {
            const int32_t col = G.h_col[j];
            const uint32_t enc = (uint32_t)(col * SM_C + 4 * ((col >> 2) & 3)) * 4u;      // = sm_enc(col)
            float ef;
            std::memcpy(&ef, &enc, 4);
            ell[(size_t)r * SM_CH + (j - G.h_rowptr[r])] = make_float4(ef, G.h_val0[j], G.h_val1[j], 0.f);
        }
    void* de = nullptr;
    SCN_HIP_TRY(hipMalloc(&de, std::max<size_t>(ell.size(), 1) * sizeof(float4)));
    c->plan.allocs.push_back(de);
    if (!ell.empty()) SCN_HIP_TRY(hipMemcpy(de, ell.data(), ell.size() * sizeof(float4), hipMemcpyHostToDevice));
    c->small_ell = de;
    std::vector<int32_t> op_ptr((size_t)c->n_rows + 1, 0);
    std::vector<float4> ov;
    for (int32_t r = 0; r < c->n_rows; ++r) {
        op_ptr[r] = (int32_t)ov.size();
        for (int64_t j = (int64_t)G.h_rowptr[r] + SM_CH; j < G.h_rowptr[r + 1]; ++j) {
            const int32_t col = G.h_col[j];
            const uint32_t enc = (uint32_t)(col * SM_C + 4 * ((col >> 2) & 3)) * 4u;
            float ef;
            std::memcpy(&ef, &enc, 4);
            ov.push_back(make_float4(ef, G.h_val0[j], G.h_val1[j], 0.f));
        }
    }
    op_ptr[c->n_rows] = (int32_t)ov.size();
    c->small_n_ovf = (int32_t)ov.size();
    void* dp = nullptr;
    SCN_HIP_TRY(hipMalloc(&dp, op_ptr.size() * 4));
    c->plan.allocs.push_back(dp);
    SCN_HIP_TRY(hipMemcpy(dp, op_ptr.data(), op_ptr.size() * 4, hipMemcpyHostToDevice));
    c->small_ovf_ptr = (int32_t*)dp;
    void* dv = nullptr;
    SCN_HIP_TRY(hipMalloc(&dv, std::max<size_t>(ov.size(), 1) * sizeof(float4)));
    c->plan.allocs.push_back(dv);
    if (!ov.empty()) SCN_HIP_TRY(hipMemcpy(dv, ov.data(), ov.size() * sizeof(float4), hipMemcpyHostToDevice));
    c->small_ovf = dv;
    {                                                           // (the kernels' LDS limit: whatever a launch's layout comes to, overflow list included)
        const int waves = small_waves(c->n_rows);
        const int tiles_per_wave = ((lay.epad >> 4) + waves - 1) / waves;
        int st;
        if (waves == 12) st = tiles_per_wave <= 1 ? small_raise_lds<1, 12>((size_t)160 * 1024) : small_raise_lds<2, 12>((size_t)160 * 1024);
        else if (tiles_per_wave <= 3) st = small_raise_lds<3, 8>((size_t)160 * 1024);
        else if (tiles_per_wave <= 6) st = small_raise_lds<6, 8>((size_t)160 * 1024);
        else if (tiles_per_wave <= 8) st = small_raise_lds<8, 8>((size_t)160 * 1024);
        else st = small_raise_lds<SM_MAXT, 8>((size_t)160 * 1024);
        if (st != SCN_OK) return st;
        if (waves == 8 && small_blocks(c->n_rows) >= 2) {
            switch ((small_blocks(c->n_rows) + 1) / 2) {
                case 2: st = small_raise_lds<2, 8, true>((size_t)160 * 1024); break;
                case 3: st = small_raise_lds<3, 8, true>((size_t)160 * 1024); break;
                case 4: st = small_raise_lds<4, 8, true>((size_t)160 * 1024); break;
                default: st = small_raise_lds<5, 8, true>((size_t)160 * 1024); break;
            }
            if (st != SCN_OK) return st;
        }
    }
    return SCN_OK;
}

}  // namespace scn

using namespace scn;

#ifdef SCN_STAMPS
extern "C" int scn_debug_small_stamps(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(scn::g_small_stamps), 128) == hipSuccess ? 0 : -3;
}
extern "C" int scn_debug_small_cycles(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(scn::g_small_cycles), 128) == hipSuccess ? 0 : -3;
}
#endif

extern "C" {

int scn_small_step_supported(scn_conv_t conv, int32_t n_layers, int32_t hidden, int32_t max_deg, int32_t max_items) {
    if (!small_shape(conv)) return 0;
    if (n_layers < 2 || n_layers > SM_MAX_LAYERS || hidden != SM_C) return 0;
    if (max_deg <= 0 || max_deg > SM_MAXD || max_items > SM_ITEMS) return 0;
    if (conv->n_rows > 16 * 8 * SM_MAXT || !conv->small_pack) return 0;     // the pack is built by scn_conv_create* (small_prepare)
    return small_lds(conv->n_rows, 0).total <= 160 * 1024 ? 1 : 0;
}

size_t scn_small_step_workspace(int32_t n_edges, int32_t n_traj, int32_t n_layers) {
    if (n_edges <= 0 || n_traj <= 0 || n_layers < 2 || n_layers > SM_MAX_LAYERS) return 0;
    const size_t hs = (size_t)(n_layers - 1) * n_traj * n_edges * SM_C * 4;
    const size_t ys = (size_t)n_traj * n_edges * 16;
    const size_t part = ((size_t)2 * n_traj * small_pw(n_layers) * 4 + 15) / 16 * 16;     // (sized for the paired form: 2 N partials,
    const size_t flags = (size_t)n_traj * 4 * SM_MAX_LAYERS * 4;                          //  hand-over buffers of the size of hs, flags)
    return hs + ys + part + (size_t)2 * n_traj * 8 + hs + flags + 256;
}

int scn_small_step_pairing(int32_t mode) {
    if (mode < 0 || mode > 1) return SCN_ERR_BAD_ARG;
    g_small_pairing.store(mode);
    return SCN_OK;
}

}  // extern "C"

struct SmallAdam {
    float* w; float* m; float* v;
    float lr, b1, b2, eps, weight_decay, g_scale;
    int32_t* step;
};

static int small_step_impl(scn_conv_t conv, scn_conv_t conv_t, int32_t n_slabs, int32_t ns, int32_t n_layers, int32_t hidden,
                           const float* x, const int32_t* last_nodes, const float* y, float scale, const int32_t* nbr, int32_t n_nodes,
                           int32_t max_deg, int32_t max_items, const int32_t* inc_ptr, const int32_t* inc_edge, const float* inc_sign,
                           const float* const* W, int32_t act, float* const* dW, double* loss, int32_t overwrite, void* workspace,
                           size_t workspace_bytes, void* stream, const SmallAdam* adam) {
    if (!conv || !conv_t || !x || !last_nodes || !y || !nbr || !inc_ptr || !inc_edge || !inc_sign || !W || !dW || !loss || !workspace)
        return SCN_ERR_BAD_ARG;
    if (n_slabs <= 0 || ns <= 0 || n_nodes <= 0 || act < 0 || act > 3) return SCN_ERR_BAD_SHAPE;
    if (!small_shape(conv_t) || conv_t->n_rows != conv->n_rows) return SCN_ERR_BAD_SHAPE;
    if (!scn_small_step_supported(conv, n_layers, hidden, max_deg, max_items)) return SCN_ERR_UNSUPPORTED;
    for (int k = 0; k < 3 * n_layers + 1; ++k)
        if (!W[k] || !dW[k]) return SCN_ERR_BAD_ARG;
    const int E = conv->n_rows, N = n_slabs * ns;
    if (workspace_bytes < scn_small_step_workspace(E, N, n_layers)) return SCN_ERR_WORKSPACE;
    SmallArgs a{};
    a.n_edges = E; a.ns = ns; a.n_layers = n_layers; a.act = act; a.max_deg = max_deg;
    a.same_t = conv_t == conv ? 1 : 0;
    if (!conv_t->small_pack) return SCN_ERR_UNSUPPORTED;
    a.ent = (const float4*)conv->small_pack;
    a.ent_t = (const float4*)conv_t->small_pack;
    a.ell = (const float4*)conv->small_ell;
    a.ell_t = (const float4*)conv_t->small_ell;
    a.ovf = (const float4*)conv->small_ovf; a.ovf_ptr = conv->small_ovf_ptr; a.n_ovf = conv->small_n_ovf;
    a.ovf_t = (const float4*)conv_t->small_ovf; a.ovf_ptr_t = conv_t->small_ovf_ptr; a.n_ovf_t = conv_t->small_n_ovf;
    a.rowptr = conv->g[0].d_rowptr;
    a.rowptr_t = conv_t->g[0].d_rowptr;
    a.x = x; a.last_nodes = last_nodes; a.y = y; a.scale = scale;
    a.nbr = nbr; a.inc_ptr = inc_ptr; a.inc_edge = inc_edge; a.inc_sign = inc_sign;
    for (int k = 0; k < 3 * n_layers + 1; ++k) a.W[k] = W[k];
    char* ws = (char*)(((uintptr_t)workspace + 255) / 256 * 256);
    a.hs = (float*)ws;
    ws += (size_t)(n_layers - 1) * N * E * SM_C * 4;
    a.ys = (float*)ws;
    ws += (size_t)N * E * 16;
    a.partial = (float*)ws;
    const int pw = small_pw(n_layers);
    ws += ((size_t)2 * N * pw * 4 + 15) / 16 * 16;
    a.loss_part = (double*)ws;
    ws += (size_t)2 * N * 8;
    a.xg = (float*)ws;
    ws += (size_t)(n_layers - 1) * N * E * SM_C * 4;
    a.flags = (int*)ws;
    a.n_traj = N;
    a.step = adam ? adam->step : nullptr;
    const bool paired = small_paired(E, N);
    if (paired) {
        // A fresh word per launch, never zero (the flags' resting value: small_reduce_kernel puts them back, so a graph replay -- same
        // tag -- starts from zeros as well).  A workspace used for the first time holds arbitrary words; one of them equal to this
        // launch's tag in the very cell a workgroup polls is a 2^-32 event per cell.
        a.tag = (int32_t)(((uint32_t)(g_small_tag.fetch_add(1) + 1) * 2654435761u) | 1u);
    }
    const SmallLds lay = small_lds(E, std::max(a.n_ovf, a.n_ovf_t));
    hipStream_t s = (hipStream_t)stream;
    const int waves = small_waves(E);
    const int tiles_per_wave = ((lay.epad >> 4) + waves - 1) / waves;
#define SCN_LAUNCH_SMALL(T, WV) hipLaunchKernelGGL((small_step_kernel<T, WV, false>), dim3(N), dim3(64 * WV), lay.total, s, a)
#define SCN_LAUNCH_PAIRED(T) hipLaunchKernelGGL((small_step_kernel<T, 8, true>), dim3(16 * ((N + 7) / 8)), dim3(512), lay.total, s, a)
    if (paired) {
        switch ((small_blocks(E) + 1) / 2) {
            case 2: SCN_LAUNCH_PAIRED(2); break;
            case 3: SCN_LAUNCH_PAIRED(3); break;
            case 4: SCN_LAUNCH_PAIRED(4); break;
            default: SCN_LAUNCH_PAIRED(5); break;
        }
    } else if (waves == 12) { if (tiles_per_wave <= 1) SCN_LAUNCH_SMALL(1, 12); else SCN_LAUNCH_SMALL(2, 12); }
    else if (tiles_per_wave <= 3) SCN_LAUNCH_SMALL(3, 8);
    else if (tiles_per_wave <= 6) SCN_LAUNCH_SMALL(6, 8);
    else if (tiles_per_wave <= 8) SCN_LAUNCH_SMALL(8, 8);
    else SCN_LAUNCH_SMALL(SM_MAXT, 8);
    SCN_LAUNCH_CHECK();
    SmallReduce r{};
    r.n_traj = paired ? 2 * N : N; r.pw = pw;
    r.flags = paired ? a.flags : nullptr;
    r.n_flags = paired ? N * 4 * SM_MAX_LAYERS : 0; r.n_w = 3 * n_layers + 1; r.overwrite = overwrite ? 1 : 0;
    int off = 0;
    for (int k = 0; k < r.n_w; ++k) {
        r.off[k] = off;
        r.dW[k] = dW[k];
        off += k < 3 ? SM_C : (k < 3 * n_layers ? SM_C * SM_C : SM_C);
    }
    r.off[r.n_w] = off;
    r.partial = a.partial; r.loss_part = a.loss_part; r.loss = loss;
    if (adam) {
        r.w = adam->w; r.m = adam->m; r.v = adam->v;
        r.lr = adam->lr; r.b1 = adam->b1; r.b2 = adam->b2; r.eps = adam->eps;
        r.wd2 = 2.f * adam->weight_decay; r.g_scale = adam->g_scale;
        r.step = adam->step;
    }
    hipLaunchKernelGGL(small_reduce_kernel, dim3((pw + 15) / 16), dim3(256), 0, s, r);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" {

int scn_small_step(scn_conv_t conv, scn_conv_t conv_t, int32_t n_slabs, int32_t ns, int32_t n_layers, int32_t hidden,
                   const float* x, const int32_t* last_nodes, const float* y, float scale, const int32_t* nbr, int32_t n_nodes,
                   int32_t max_deg, int32_t max_items, const int32_t* inc_ptr, const int32_t* inc_edge, const float* inc_sign,
                   const float* const* W, int32_t act, float* const* dW, double* loss, int32_t overwrite, void* workspace,
                   size_t workspace_bytes, void* stream) {
    return small_step_impl(conv, conv_t, n_slabs, ns, n_layers, hidden, x, last_nodes, y, scale, nbr, n_nodes, max_deg, max_items,
                           inc_ptr, inc_edge, inc_sign, W, act, dW, loss, overwrite, workspace, workspace_bytes, stream, nullptr);
}

int scn_small_step_adam(scn_conv_t conv, scn_conv_t conv_t, int32_t n_slabs, int32_t ns, int32_t n_layers, int32_t hidden,
                        const float* x, const int32_t* last_nodes, const float* y, float scale, const int32_t* nbr, int32_t n_nodes,
                        int32_t max_deg, int32_t max_items, const int32_t* inc_ptr, const int32_t* inc_edge, const float* inc_sign,
                        const float* const* W, int32_t act, float* const* dW, double* loss, void* workspace, size_t workspace_bytes,
                        float* w_flat, float* m_flat, float* v_flat, float lr, float b1, float b2, float eps, int32_t* step_dev,
                        float weight_decay, void* stream) {
    if (!w_flat || !m_flat || !v_flat || !step_dev || !W) return SCN_ERR_BAD_ARG;
    if (n_layers < 2 || n_layers > SM_MAX_LAYERS) return SCN_ERR_UNSUPPORTED;
    int off = 0;
    for (int k = 0; k < 3 * n_layers + 1; ++k) {                // the weights as ONE flat buffer in the order of the list
        if (W[k] != w_flat + off) return SCN_ERR_BAD_ARG;
        off += k < 3 ? SM_C : (k < 3 * n_layers ? SM_C * SM_C : SM_C);
    }
    const SmallAdam ad{w_flat, m_flat, v_flat, lr, b1, b2, eps, weight_decay, 1.f, step_dev};
    return small_step_impl(conv, conv_t, n_slabs, ns, n_layers, hidden, x, last_nodes, y, scale, nbr, n_nodes, max_deg, max_items,
                           inc_ptr, inc_edge, inc_sign, W, act, dW, loss, 1, workspace, workspace_bytes, stream, &ad);
}

}  // extern "C"
