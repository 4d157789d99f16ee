// Dense per-point contractions over already-gathered terms (Bunch / SCCONV layers at scale, TE:181-195):
//   forward : out[p,:]  = act( sum_k G_k[p,:] @ W_k )                       G_k = S_k X_src(k)  (blocked SpMM)
//   backward: dx[p,:]   = ( sum_k G'_k[p,:] @ W_k^T ) * act'(aux[p,:])      G'_k = S_k^T dZ_dst(k)
//             dW_k     += sum_p aux[p,:]^T G'_k[p,:]                        (per-block partials, fixed-order reduce)
// p runs over every (slab, row, trajectory) point of one level; all tensors are [points][channels] fp32.
// Generic widths run on VALU kernels; the width that matters at scale (every term and the output 32 channels wide) runs on
// v_mfma_f32_32x32x2_f32 with the operands streamed straight from HBM (no gather here, so no LDS staging either).
#include <algorithm>
#include <cstring>

#include "scn_internal.h"

namespace scn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int DN_MAX_TERMS = 3;
constexpr int DN_MAX_TERMS1 = 6;      // the rank-one streaming kernels (every term ONE channel wide): three terms x (positive, negative part)
constexpr int DN_THREADS = 256;

struct DenseFwdArgs {
    int64_t n_points;
    int32_t n_terms, c_out, act;
    int32_t c_in[DN_MAX_TERMS];
    const float* G[DN_MAX_TERMS];
    const float* W[DN_MAX_TERMS];
    float* out;
};

// thread = (point, group of 4 output channels) ; weights of all terms in LDS
__global__ __launch_bounds__(DN_THREADS) void dense_fwd_kernel(DenseFwdArgs a) {
    extern __shared__ float wl[];                 // [term][c_in][c_out]
    int woff[DN_MAX_TERMS + 1];
    woff[0] = 0;
    for (int k = 0; k < a.n_terms; ++k) woff[k + 1] = woff[k] + a.c_in[k] * a.c_out;
    for (int k = 0; k < a.n_terms; ++k)
        for (int i = threadIdx.x; i < a.c_in[k] * a.c_out; i += DN_THREADS) wl[woff[k] + i] = a.W[k][i];
    __syncthreads();
    const int cg = (a.c_out + 3) / 4;             // channel groups per point
    const int64_t total = a.n_points * cg;
    for (int64_t idx = (int64_t)blockIdx.x * DN_THREADS + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * DN_THREADS) {
        const int64_t pnt = idx / cg;
        const int c0 = (int)(idx - pnt * cg) * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < a.n_terms; ++k) {
            const int ci = a.c_in[k];
            const float* g = a.G[k] + pnt * ci;
            const float* w = wl + woff[k];
            for (int c = 0; c < ci; ++c) {
                const float x = g[c];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c0 + j < a.c_out) acc[j] = fmaf(x, w[c * a.c_out + c0 + j], acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c0 + j < a.c_out) a.out[pnt * a.c_out + c0 + j] = act_apply_fast(a.act, acc[j]);
    }
}

struct DenseBwdArgs {
    int64_t n_points;
    int32_t n_terms, c_aux, act;
    int32_t c[DN_MAX_TERMS];
    const float* G[DN_MAX_TERMS];
    const float* W[DN_MAX_TERMS];                // [c_aux][c_k] forward weights
    const float* aux;
    float* dx;                                   // may be null
    float* partial;                              // [gridDim.x][sum_k c_aux*c_k]
};

constexpr int DN_TILE = 64;                      // points per LDS tile

// one block: loop over tiles of 64 points; aux and G tiles staged in LDS; thread-owned dW entries in registers
__global__ __launch_bounds__(DN_THREADS) void dense_bwd_kernel(DenseBwdArgs a) {
    extern __shared__ float sm[];
    // LDS: W [sum c_aux*c_k] | AUX [DN_TILE][c_aux] | G_k [DN_TILE][c_k] ...
    int woff[DN_MAX_TERMS + 1], goff[DN_MAX_TERMS + 1];
    woff[0] = 0;
    for (int k = 0; k < a.n_terms; ++k) woff[k + 1] = woff[k] + a.c_aux * a.c[k];
    float* W = sm;
    float* AUX = W + woff[a.n_terms];
    goff[0] = 0;
    for (int k = 0; k < a.n_terms; ++k) goff[k + 1] = goff[k] + DN_TILE * a.c[k];
    float* G = AUX + DN_TILE * a.c_aux;
    for (int k = 0; k < a.n_terms; ++k)
        for (int i = threadIdx.x; i < a.c_aux * a.c[k]; i += DN_THREADS) W[woff[k] + i] = a.W[k][i];
    // dW ownership: entry e of term k (e = ca*c_k + cc) belongs to thread e % 256, slot e / 256 (<= 4 per term)
    float dw[DN_MAX_TERMS][4];
#pragma unroll
    for (int k = 0; k < DN_MAX_TERMS; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) dw[k][i] = 0.f;
    const int64_t n_tiles = (a.n_points + DN_TILE - 1) / DN_TILE;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t p0 = t * DN_TILE;
        const int np = (int)std::min<int64_t>(DN_TILE, a.n_points - p0);
        __syncthreads();
        for (int i = threadIdx.x; i < np * a.c_aux; i += DN_THREADS) AUX[i] = a.aux[p0 * a.c_aux + i];
        for (int k = 0; k < a.n_terms; ++k)
            for (int i = threadIdx.x; i < np * a.c[k]; i += DN_THREADS) G[goff[k] + i] = a.G[k][p0 * a.c[k] + i];
        __syncthreads();
        if (a.dx) {
            for (int o = threadIdx.x; o < np * a.c_aux; o += DN_THREADS) {
                const int pl = o / a.c_aux, ca = o - pl * a.c_aux;
                float acc = 0.f;
                for (int k = 0; k < a.n_terms; ++k) {
                    const float* g = G + goff[k] + pl * a.c[k];
                    const float* w = W + woff[k] + ca * a.c[k];
                    for (int cc = 0; cc < a.c[k]; ++cc) acc = fmaf(g[cc], w[cc], acc);
                }
                a.dx[p0 * a.c_aux + o] = acc * act_grad_from_output(a.act, AUX[o]);
            }
        }
#pragma unroll
        for (int k = 0; k < DN_MAX_TERMS; ++k) {
            if (k < a.n_terms) {
                const int ck = a.c[k], ne = a.c_aux * ck;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = threadIdx.x + i * DN_THREADS;
                    if (e < ne) {
                        const int ca = e / ck, cc = e - ca * ck;
                        float acc = dw[k][i];
                        for (int pl = 0; pl < np; ++pl) acc = fmaf(AUX[pl * a.c_aux + ca], G[goff[k] + pl * ck + cc], acc);
                        dw[k][i] = acc;
                    }
                }
            }
        }
    }
    float* outp = a.partial + (size_t)blockIdx.x * woff[a.n_terms];
#pragma unroll
    for (int k = 0; k < DN_MAX_TERMS; ++k)
        if (k < a.n_terms) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = threadIdx.x + i * DN_THREADS;
                if (e < a.c_aux * a.c[k]) outp[woff[k] + e] = dw[k][i];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// MFMA path: every term and the output are 32 channels wide.  One wave = tiles of 32 points:
//   lane = (point p = lane & 31, half h = lane >> 5) loads its point's channels 16h .. 16h+15 of every term (4 x 16 B),
//   A[m = point][k] = G_k[point][16h + s], B[k][n] = W_k[16h + s][n] (registers), D[m = pt(r, h)][n = lane & 31].
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int DM_WAVES = 4, DM_THREADS = 64 * DM_WAVES;
constexpr int DM_PSTRIDE = 36;                   // floats per point row of the transpose patch

template <int ACT>
__global__ __launch_bounds__(DM_THREADS, 2) void dense_fwd_mfma_kernel(DenseFwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    float Bt[DN_MAX_TERMS][16];
#pragma unroll
    for (int k = 0; k < DN_MAX_TERMS; ++k)
#pragma unroll
        for (int s = 0; s < 16; ++s) Bt[k][s] = k < a.n_terms ? a.W[k][(16 * h + s) * 32 + p] : 0.f;
    const int64_t n_tiles = (a.n_points + 31) / 32;
    for (int64_t t = (int64_t)blockIdx.x * DM_WAVES + wave; t < n_tiles; t += (int64_t)gridDim.x * DM_WAVES) {
        const int64_t p0 = t * 32;
        const int64_t pt = std::min<int64_t>(p0 + p, a.n_points - 1);          // clamp: tail lanes re-read the last point
        f32x4 g[DN_MAX_TERMS][4];
#pragma unroll
        for (int k = 0; k < DN_MAX_TERMS; ++k)
            if (k < a.n_terms) {
#pragma unroll
                for (int q = 0; q < 4; ++q) g[k][q] = *(const f32x4*)(a.G[k] + pt * 32 + 16 * h + 4 * q);
            }
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < DN_MAX_TERMS; ++k)
            if (k < a.n_terms) {
#pragma unroll
                for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g[k][s >> 2][s & 3], Bt[k][s], acc, 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t q = p0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (q < a.n_points) a.out[q * 32 + p] = act_apply_fast(ACT, acc[r]);
        }
    }
}

// backward: dX tile by MFMA against W_k^T, dW_k += aux^T G_k by a second MFMA whose B operand (a channel's 32 points) comes
// through a per-wave LDS transpose patch; per-block partials in the layout dense_dw_reduce expects.
template <int ACT>
__global__ __launch_bounds__(DM_THREADS, 2) void dense_bwd_mfma_kernel(DenseBwdArgs a) {
    __shared__ float lds[DM_WAVES * 32 * DM_PSTRIDE];               // patches; reused for the final cross-wave reduction
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    float* patch = lds + wave * 32 * DM_PSTRIDE;
    float Bt[DN_MAX_TERMS][16];                                       // B[k = c][n = ca] = W_k[ca = p][c = 16h + s]
#pragma unroll
    for (int k = 0; k < DN_MAX_TERMS; ++k)
#pragma unroll
        for (int s = 0; s < 16; ++s) Bt[k][s] = k < a.n_terms ? a.W[k][p * 32 + 16 * h + s] : 0.f;
    f32x16 dWacc[DN_MAX_TERMS];
#pragma unroll
    for (int k = 0; k < DN_MAX_TERMS; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) dWacc[k][r] = 0.f;
    const int64_t n_tiles = (a.n_points + 31) / 32;
    for (int64_t t = (int64_t)blockIdx.x * DM_WAVES + wave; t < n_tiles; t += (int64_t)gridDim.x * DM_WAVES) {
        const int64_t p0 = t * 32;
        const bool mine = p0 + p < a.n_points;
        const int64_t pt = mine ? p0 + p : a.n_points - 1;
        f32x4 g[DN_MAX_TERMS][4];
#pragma unroll
        for (int k = 0; k < DN_MAX_TERMS; ++k)
            if (k < a.n_terms) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *(const f32x4*)(a.G[k] + pt * 32 + 16 * h + 4 * q);
                    g[k][q] = mine ? v : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        float ax[16];                                                  // aux in the D layout: point pt(r, h), channel p
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t q = p0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            ax[r] = q < a.n_points ? a.aux[q * 32 + p] : 0.f;
        }
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < DN_MAX_TERMS; ++k)
            if (k < a.n_terms) {
#pragma unroll
                for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g[k][s >> 2][s & 3], Bt[k][s], acc, 0, 0, 0);
            }
        if (a.dx) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t q = p0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (q < a.n_points) a.dx[q * 32 + p] = acc[r] * act_grad_from_output(ACT, ax[r]);
            }
        }
#pragma unroll
        for (int k = 0; k < DN_MAX_TERMS; ++k)
            if (k < a.n_terms) {
#pragma unroll
                for (int q = 0; q < 4; ++q) *(f32x4*)(patch + p * DM_PSTRIDE + 16 * h + 4 * q) = g[k][q];
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int s = 0; s < 16; ++s) {                         // k-step s contracts points pt(s, h)
                    const int ps = (s & 3) + 8 * (s >> 2) + 4 * h;
                    dWacc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[s], patch[ps * DM_PSTRIDE + p], dWacc[k], 0, 0, 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
    }
    // dWacc[k][r] = dW_k[ca = pt(r, h)][c = p]: add the four waves in a fixed order, emit this block's partial
    int woff[DN_MAX_TERMS + 1];
    woff[0] = 0;
    for (int k = 0; k < a.n_terms; ++k) woff[k + 1] = woff[k] + 1024;
    float* outp = a.partial + (size_t)blockIdx.x * woff[a.n_terms];
    for (int k = 0; k < a.n_terms; ++k) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ca = (r & 3) + 8 * (r >> 2) + 4 * h;
            lds[wave * 1024 + ca * 32 + p] = dWacc[k][r];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 1024; i += DM_THREADS)
            outp[woff[k] + i] = (lds[i] + lds[1024 + i]) + (lds[2048 + i] + lds[3072 + i]);
    }
}

// one output channel (the last Bunch layer, TE:198) with terms of width % 4 == 0: thread = (point, 16-byte chunk of a term
// row) so that a wave reads whole rows coalesced; the chunk dot products are folded over the CQ lanes of a point.
template <int CQ>      // chunks per point row (c_in / 4): 4 or 8
__global__ __launch_bounds__(256) void dense_fwd_out1_kernel(DenseFwdArgs a) {
    const int cq = threadIdx.x % CQ;
    f32x4 w[DN_MAX_TERMS];
#pragma unroll
    for (int k = 0; k < DN_MAX_TERMS; ++k) w[k] = k < a.n_terms ? *(const f32x4*)(a.W[k] + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t total = a.n_points * CQ;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total + (CQ - 1); i += (int64_t)gridDim.x * 256) {
        const bool ok = i < total;                                      // keep whole lane groups inside the shuffles
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < DN_MAX_TERMS; ++k)
            if (k < a.n_terms && ok) {
                const f32x4 g = *(const f32x4*)(a.G[k] + i * 4);
                s += g[0] * w[k][0] + g[1] * w[k][1] + g[2] * w[k][2] + g[3] * w[k][3];
            }
#pragma unroll
        for (int m = 1; m < CQ; m <<= 1) s += __shfl_xor(s, m, 64);
        if (ok && cq == 0) a.out[i / CQ] = act_apply_fast(a.act, s);
    }
}

// every term one channel wide (first layer of the composed Ebli / Bunch models): out[p][:] = act(sum_k g_k[p] * W_k[0][:]);
// thread = (point, 4 output channels), weights in registers, coalesced 16-byte stores -- a pure write stream.
// (CG = c_out / 4 and the activation are template parameters: no run-time 64-bit division or activation switch per element,
// -5 % on the composed Ebli first layer, -7 % on the Bunch one at |E| = 1M)
struct DenseIn1Args {
    int64_t n_points;
    int32_t n_terms;
    const float* G[DN_MAX_TERMS1];
    const float* W[DN_MAX_TERMS1];
    float* out;
};
template <int CG, int ACT>
__global__ __launch_bounds__(256) void dense_fwd_in1_kernel(DenseIn1Args a) {
    static_assert(256 % CG == 0, "a thread keeps its channel group over the grid-stride loop");
    const int cq = threadIdx.x % CG;
    f32x4 w[DN_MAX_TERMS1];
#pragma unroll
    for (int k = 0; k < DN_MAX_TERMS1; ++k) w[k] = k < a.n_terms ? *(const f32x4*)(a.W[k] + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
    // IN1_U items per thread and trip, every scalar load of a trip issued before the first use: with up to six 4-byte loads per
    // 16-byte store the kernel is bound by load latency, not by bytes (x6 -> 32 at |E| = 1M: 2.3 TB/s with one item per trip)
    constexpr int IN1_U = 4;
    const int64_t total = a.n_points * CG, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += IN1_U * stride) {
        float g[IN1_U][DN_MAX_TERMS1];
#pragma unroll
        for (int u = 0; u < IN1_U; ++u) {
            const int64_t iu = i + u * stride;
            const int64_t pnt = (iu < total ? iu : i) / CG;
#pragma unroll
            for (int k = 0; k < DN_MAX_TERMS1; ++k) g[u][k] = k < a.n_terms ? a.G[k][pnt] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < IN1_U; ++u) {
            const int64_t iu = i + u * stride;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < DN_MAX_TERMS1; ++k)
                if (k < a.n_terms) v += g[u][k] * w[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = act_apply_fast(ACT, v[j]);
            if (iu < total) __builtin_nontemporal_store(v, (f32x4*)(a.out + iu * 4));
        }
    }
}

// every term ONE channel wide in the backward (the last Bunch layer, 32 -> 1, TE:184-192: G'_k = S_k^T dz is a scalar per point):
//   dx[p][ca] = (sum_k g_k[p] W_k[ca][0]) * act'(aux[p][ca]),   dW_k[ca][0] += sum_p aux[p][ca] g_k[p]
// thread = (point, 4 aux channels): aux read once, dx written once, coalesced 16-byte accesses -- a pure stream; per-thread
// partial sums reduced per block in a fixed order.  partial layout: [block][k * c_aux + ca] (what dense_dw_reduce expects).
struct DenseBwdG1Args {
    int64_t n_points;
    int32_t n_terms, c_aux, act;
    const float* G[DN_MAX_TERMS1];
    const float* W[DN_MAX_TERMS1];
    const float* aux;
    float* dx;
    float* partial;
};
template <int CG, int ACT>                                        // CG = c_aux / 4 in {4, 8, 16}: 256 % CG == 0
__global__ __launch_bounds__(256) void dense_bwd_g1_kernel(DenseBwdG1Args a) {
    __shared__ f32x4 red[DN_MAX_TERMS1 * 256];
    constexpr int cg = CG;
    const int cq = threadIdx.x % cg;
    f32x4 w[DN_MAX_TERMS1], acc[DN_MAX_TERMS1];
#pragma unroll
    for (int k = 0; k < DN_MAX_TERMS1; ++k) {
        w[k] = k < a.n_terms ? *(const f32x4*)(a.W[k] + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
        acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // two items per thread and trip, their loads issued together (see dense_fwd_in1_kernel: load latency, not bytes)
    constexpr int G1_U = 2;
    const int64_t total = a.n_points * cg, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += G1_U * stride) {
        f32x4 x[G1_U];
        float g[G1_U][DN_MAX_TERMS1];
#pragma unroll
        for (int u = 0; u < G1_U; ++u) {
            const int64_t iu = i + u * stride;
            const bool ok = iu < total;
            const int64_t is = ok ? iu : i;
            x[u] = *(const f32x4*)(a.aux + is * 4);
            if (!ok) x[u] = f32x4{0.f, 0.f, 0.f, 0.f};                   // (a repeated item adds nothing to the sums)
#pragma unroll
            for (int k = 0; k < DN_MAX_TERMS1; ++k) g[u][k] = k < a.n_terms ? a.G[k][is / cg] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < G1_U; ++u) {
            const int64_t iu = i + u * stride;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < DN_MAX_TERMS1; ++k)
                if (k < a.n_terms) {
                    v += g[u][k] * w[k];
                    acc[k] += g[u][k] * x[u];
                }
            if (a.dx && iu < total) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= act_grad_from_output(ACT, x[u][j]);
                __builtin_nontemporal_store(v, (f32x4*)(a.dx + iu * 4));  // (a plain store: +5 %)
            }
        }
    }
#pragma unroll
    for (int k = 0; k < DN_MAX_TERMS1; ++k) red[k * 256 + threadIdx.x] = acc[k];
    __syncthreads();
    if ((int)threadIdx.x < a.n_terms * cg) {                        // threads with equal tid % cg hold the same channels
        const int k = threadIdx.x / cg, q = threadIdx.x - k * cg;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int t = q; t < 256; t += cg) sum += red[k * 256 + t];
        float* o = a.partial + (size_t)blockIdx.x * (a.n_terms * a.c_aux) + k * a.c_aux + q * 4;
        o[0] = sum[0]; o[1] = sum[1]; o[2] = sum[2]; o[3] = sum[3];
    }
}

struct DenseReduceArgs {
    const float* partial;
    int32_t n_partials, total, n_terms;
    int32_t off[DN_MAX_TERMS1 + 1];
    float* dW[DN_MAX_TERMS1];
};
__global__ void dense_dw_reduce(DenseReduceArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    float s = 0.f;
    for (int b = 0; b < a.n_partials; ++b) s += a.partial[(size_t)b * a.total + i];
    for (int k = 0; k < a.n_terms; ++k)
        if (i < a.off[k + 1]) {
            if (a.dW[k]) a.dW[k][i - a.off[k]] += s;
            return;
        }
}

// ------------------------------------------------------------------------------------------------
// The first TWO Bunch layers without a 32-channel gather (scn_split_sign / scn_fold1_*; derivation in profiles/HISTORY.md section 3.1):
// bunch_func starts from [0, flow, 0] (TE:179), so the first layer's output of every level is relu of ONE rank-one term,
//   H1_j[p][:] = relu(g_j[p] w_j) = max(g_j[p], 0) relu(w_j) + min(g_j[p], 0) min(w_j, 0),        g_j = S x  (one channel)
// and the second layer's pre-activation is a sum of rank-one terms of SHIFTED SCALARS:
//   (S_k H1_j) W_k = (S_k g_j^+)[p] (relu(w_j) W_k) + (S_k g_j^-)[p] (min(w_j, 0) W_k).
// The shifts run on one-channel tensors, the expansion is dense_fwd_in1_kernel, and both layers' weight gradients follow from
// u_k^(+/-) = sum_p (S_k g_j^(+/-))[p] dZ2[p][:] (dense_bwd_g1_kernel: one stream over dZ2) by the two folds below.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_sign_kernel(int64_t n, const float* __restrict__ g, float* __restrict__ gp,
                                                         float* __restrict__ gm) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = g[i];
        gp[i] = fmaxf(v, 0.f);
        gm[i] = fminf(v, 0.f);
    }
}

// out = act(t0 + t1 [+ t2 + t3]) elementwise, 16 bytes per thread and trip (hidden widths above 32: the partial pre-activations /
// partial input gradients of a layer's 32-channel blocks, summed in term order; out may be t0)
struct SumTerms { const float* t[4]; int n; };
template <int ACT>
__global__ __launch_bounds__(256) void sum_act_kernel(int64_t n4, SumTerms T, float* out) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 v = ((const f32x4*)T.t[0])[i];
        for (int k = 1; k < T.n; ++k) v += ((const f32x4*)T.t[k])[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_apply_fast(ACT, v[j]);
        ((f32x4*)out)[i] = v;
    }
}

// Ap[c] = sum_a relu(w1[a]) W2[a][c],  Am[c] = sum_a min(w1[a], 0) W2[a][c]        (one workgroup, thread = output channel)
__global__ void fold1_forward_kernel(const float* __restrict__ w1, const float* __restrict__ W2, int c1, int c2,
                                     float* __restrict__ Ap, float* __restrict__ Am) {
    const int c = threadIdx.x;
    if (c >= c2) return;
    float sp = 0.f, sm = 0.f;
    for (int a = 0; a < c1; ++a) {
        const float w = w1[a], v = W2[a * c2 + c];
        sp = fmaf(fmaxf(w, 0.f), v, sp);
        sm = fmaf(fminf(w, 0.f), v, sm);
    }
    Ap[c] = sp;
    Am[c] = sm;
}

// dW2[a][c] += relu(w1[a]) up[c] + min(w1[a], 0) um[c]
// dw1[a]    += [w1[a] > 0] sum_c up[c] W2[a][c] + [w1[a] < 0] sum_c um[c] W2[a][c]          (thread = row a of W2)
__global__ void fold1_backward_kernel(const float* __restrict__ w1, const float* __restrict__ W2, const float* __restrict__ up,
                                      const float* __restrict__ um, int c1, int c2, float* __restrict__ dW2,
                                      float* __restrict__ dw1) {
    const int a = threadIdx.x;
    if (a >= c1) return;
    const float w = w1[a], wp = fmaxf(w, 0.f), wm = fminf(w, 0.f);
    float sp = 0.f, sm = 0.f;
    for (int c = 0; c < c2; ++c) {
        const float v = W2[a * c2 + c], p = up[c], m = um[c];
        dW2[a * c2 + c] += wp * p + wm * m;
        sp = fmaf(p, v, sp);
        sm = fmaf(m, v, sm);
    }
    dw1[a] += (w > 0.f ? sp : 0.f) + (w < 0.f ? sm : 0.f);
}

static int dense_blocks(int64_t n_points) { return (int)std::min<int64_t>(2048, (n_points + DN_TILE - 1) / DN_TILE); }

}  // namespace scn

using namespace scn;

extern "C" {

int scn_dense_terms_forward(int64_t n_points, int32_t n_terms, const float* const* G, const int32_t* c_in,
                            const float* const* W, int32_t c_out, int32_t act, float* out, void* stream) {
    if (!G || !c_in || !W || !out) return SCN_ERR_BAD_ARG;
    if (n_points <= 0 || n_terms <= 0 || n_terms > DN_MAX_TERMS1 || c_out <= 0 || act < 0 || act > 3) return SCN_ERR_BAD_SHAPE;
    {
        bool in1 = c_out == 16 || c_out == 32 || c_out == 64;
        for (int k = 0; k < n_terms; ++k) in1 = in1 && c_in[k] == 1;
        if (in1) {                                                  // one-channel terms (up to DN_MAX_TERMS1 of them)
            DenseIn1Args a1;
            std::memset(&a1, 0, sizeof(a1));
            a1.n_points = n_points; a1.n_terms = n_terms; a1.out = out;
            for (int k = 0; k < n_terms; ++k) {
                if (!G[k] || !W[k]) return SCN_ERR_BAD_ARG;
                a1.G[k] = G[k]; a1.W[k] = W[k];
            }
            const int blocks = (int)std::min<int64_t>(8192, (n_points * (c_out / 4) + 255) / 256);
#define SCN_LAUNCH_IN1(CG)                                                                                                   \
    switch (act) {                                                                                                           \
        case SCN_ACT_TANH: hipLaunchKernelGGL((dense_fwd_in1_kernel<CG, SCN_ACT_TANH>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a1); break; \
        case SCN_ACT_RELU: hipLaunchKernelGGL((dense_fwd_in1_kernel<CG, SCN_ACT_RELU>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a1); break; \
        case SCN_ACT_LEAKY_RELU: hipLaunchKernelGGL((dense_fwd_in1_kernel<CG, SCN_ACT_LEAKY_RELU>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a1); break; \
        default: hipLaunchKernelGGL((dense_fwd_in1_kernel<CG, SCN_ACT_NONE>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a1); break; \
    }
            if (c_out == 16) { SCN_LAUNCH_IN1(4) } else if (c_out == 32) { SCN_LAUNCH_IN1(8) } else { SCN_LAUNCH_IN1(16) }
            SCN_LAUNCH_CHECK();
            return SCN_OK;
        }
    }
    if (n_terms > DN_MAX_TERMS) return SCN_ERR_BAD_SHAPE;
    DenseFwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_points = n_points; a.n_terms = n_terms; a.c_out = c_out; a.act = act; a.out = out;
    size_t lds = 0;
    for (int k = 0; k < n_terms; ++k) {
        if (!G[k] || !W[k] || c_in[k] <= 0) return SCN_ERR_BAD_ARG;
        a.c_in[k] = c_in[k]; a.G[k] = G[k]; a.W[k] = W[k];
        lds += (size_t)c_in[k] * c_out * sizeof(float);
    }
    bool all32 = c_out == 32;
    for (int k = 0; k < n_terms; ++k) all32 = all32 && c_in[k] == 32;
    if (all32) {
        const int blocks = (int)std::min<int64_t>(2048, ((n_points + 31) / 32 + DM_WAVES - 1) / DM_WAVES);
        hipStream_t st = (hipStream_t)stream;
        switch (act) {
            case SCN_ACT_TANH: hipLaunchKernelGGL(dense_fwd_mfma_kernel<SCN_ACT_TANH>, dim3(blocks), dim3(DM_THREADS), 0, st, a); break;
            case SCN_ACT_RELU: hipLaunchKernelGGL(dense_fwd_mfma_kernel<SCN_ACT_RELU>, dim3(blocks), dim3(DM_THREADS), 0, st, a); break;
            case SCN_ACT_LEAKY_RELU: hipLaunchKernelGGL(dense_fwd_mfma_kernel<SCN_ACT_LEAKY_RELU>, dim3(blocks), dim3(DM_THREADS), 0, st, a); break;
            default: hipLaunchKernelGGL(dense_fwd_mfma_kernel<SCN_ACT_NONE>, dim3(blocks), dim3(DM_THREADS), 0, st, a); break;
        }
        SCN_LAUNCH_CHECK();
        return SCN_OK;
    }
    bool same = c_out == 1 && (c_in[0] == 16 || c_in[0] == 32);
    for (int k = 1; k < n_terms; ++k) same = same && c_in[k] == c_in[0];
    if (same) {                                                     // one output channel, equal term widths 16 / 32
        const int cqn = c_in[0] / 4;
        const int blocks = (int)std::min<int64_t>(8192, (n_points * cqn + 255) / 256);
        if (cqn == 8) hipLaunchKernelGGL(dense_fwd_out1_kernel<8>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(dense_fwd_out1_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
        SCN_LAUNCH_CHECK();
        return SCN_OK;
    }
    if (lds > 64 * 1024) return SCN_ERR_UNSUPPORTED;
    const int64_t total = n_points * ((c_out + 3) / 4);
    const int blocks = (int)std::min<int64_t>(8192, (total + DN_THREADS - 1) / DN_THREADS);
    hipLaunchKernelGGL(dense_fwd_kernel, dim3(blocks), dim3(DN_THREADS), lds, (hipStream_t)stream, a);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

size_t scn_dense_terms_backward_workspace(int64_t n_points, int32_t n_terms, const int32_t* c, int32_t c_aux) {
    if (!c || n_points <= 0 || n_terms <= 0 || n_terms > DN_MAX_TERMS1) return 0;
    size_t tot = 0;
    for (int k = 0; k < n_terms; ++k) tot += (size_t)c_aux * c[k];
    return (size_t)std::max(dense_blocks(n_points), 1024) * tot * sizeof(float) + 256;   // (1024: the rank-one streaming kernel's grid)
}

int scn_dense_terms_backward(int64_t n_points, int32_t n_terms, const float* const* G, const int32_t* c,
                             const float* const* W, const float* aux, int32_t c_aux, int32_t act, float* dx,
                             float* const* dW, void* workspace, size_t workspace_bytes, void* stream) {
    if (!G || !c || !W || !aux || !dW || !workspace) return SCN_ERR_BAD_ARG;
    if (n_points <= 0 || n_terms <= 0 || n_terms > DN_MAX_TERMS1 || c_aux <= 0 || act < 0 || act > 3) return SCN_ERR_BAD_SHAPE;
    if (workspace_bytes < scn_dense_terms_backward_workspace(n_points, n_terms, c, c_aux)) return SCN_ERR_WORKSPACE;
    bool all1 = c_aux == 16 || c_aux == 32 || c_aux == 64;
    for (int k = 0; k < n_terms; ++k) all1 = all1 && c[k] == 1;
    if (n_terms > DN_MAX_TERMS && !all1) return SCN_ERR_BAD_SHAPE;    // more than three terms: the rank-one stream only
    DenseBwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_points = n_points; a.n_terms = n_terms; a.c_aux = c_aux; a.act = act; a.aux = aux; a.dx = dx;
    a.partial = (float*)workspace;
    size_t lds = (size_t)DN_TILE * c_aux * sizeof(float);
    DenseReduceArgs r;
    std::memset(&r, 0, sizeof(r));
    r.off[0] = 0;
    for (int k = 0; k < n_terms; ++k) {
        if (!G[k] || !W[k] || c[k] <= 0) return SCN_ERR_BAD_ARG;
        if (c_aux * c[k] > 4 * DN_THREADS) return SCN_ERR_UNSUPPORTED;
        if (k < DN_MAX_TERMS) { a.c[k] = c[k]; a.G[k] = G[k]; a.W[k] = W[k]; }
        lds += (size_t)(c_aux * c[k] + DN_TILE * c[k]) * sizeof(float);
        r.off[k + 1] = r.off[k] + c_aux * c[k];
        r.dW[k] = dW[k];
    }
    hipStream_t st = (hipStream_t)stream;
    if (all1) {                                                     // rank-one terms: the streaming kernel
        DenseBwdG1Args g;
        std::memset(&g, 0, sizeof(g));
        g.n_points = n_points; g.n_terms = n_terms; g.c_aux = c_aux; g.act = act; g.aux = aux; g.dx = dx;
        g.partial = (float*)workspace;
        for (int k = 0; k < n_terms; ++k) { g.G[k] = G[k]; g.W[k] = W[k]; }
        const int nbs = (int)std::min<int64_t>(1024, (n_points * (c_aux / 4) + 255) / 256);
#define SCN_LAUNCH_G1(CG)                                                                                         \
    switch (act) {                                                                                                \
        case SCN_ACT_TANH: hipLaunchKernelGGL((dense_bwd_g1_kernel<CG, SCN_ACT_TANH>), dim3(nbs), dim3(256), 0, st, g); break; \
        case SCN_ACT_RELU: hipLaunchKernelGGL((dense_bwd_g1_kernel<CG, SCN_ACT_RELU>), dim3(nbs), dim3(256), 0, st, g); break; \
        case SCN_ACT_LEAKY_RELU: hipLaunchKernelGGL((dense_bwd_g1_kernel<CG, SCN_ACT_LEAKY_RELU>), dim3(nbs), dim3(256), 0, st, g); break; \
        default: hipLaunchKernelGGL((dense_bwd_g1_kernel<CG, SCN_ACT_NONE>), dim3(nbs), dim3(256), 0, st, g); break; \
    }
        if (c_aux == 16) { SCN_LAUNCH_G1(4) } else if (c_aux == 32) { SCN_LAUNCH_G1(8) } else { SCN_LAUNCH_G1(16) }
        SCN_LAUNCH_CHECK();
        r.partial = g.partial; r.n_partials = nbs; r.total = r.off[n_terms]; r.n_terms = n_terms;
        hipLaunchKernelGGL(dense_dw_reduce, dim3((r.total + 255) / 256), dim3(256), 0, st, r);
        SCN_LAUNCH_CHECK();
        return SCN_OK;
    }
    const int nb = dense_blocks(n_points);
    bool all32 = c_aux == 32;
    for (int k = 0; k < n_terms; ++k) all32 = all32 && c[k] == 32;
    if (all32) {
        switch (act) {
            case SCN_ACT_TANH: hipLaunchKernelGGL(dense_bwd_mfma_kernel<SCN_ACT_TANH>, dim3(nb), dim3(DM_THREADS), 0, st, a); break;
            case SCN_ACT_RELU: hipLaunchKernelGGL(dense_bwd_mfma_kernel<SCN_ACT_RELU>, dim3(nb), dim3(DM_THREADS), 0, st, a); break;
            case SCN_ACT_LEAKY_RELU: hipLaunchKernelGGL(dense_bwd_mfma_kernel<SCN_ACT_LEAKY_RELU>, dim3(nb), dim3(DM_THREADS), 0, st, a); break;
            default: hipLaunchKernelGGL(dense_bwd_mfma_kernel<SCN_ACT_NONE>, dim3(nb), dim3(DM_THREADS), 0, st, a); break;
        }
    } else {
        if (lds > 64 * 1024) return SCN_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(dense_bwd_kernel, dim3(nb), dim3(DN_THREADS), lds, st, a);
    }
    SCN_LAUNCH_CHECK();
    r.partial = a.partial; r.n_partials = nb; r.total = r.off[n_terms]; r.n_terms = n_terms;
    hipLaunchKernelGGL(dense_dw_reduce, dim3((r.total + 255) / 256), dim3(256), 0, st, r);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

int scn_sum_act(int64_t n, int32_t n_terms, const float* const* terms, int32_t act, float* out, void* stream) {
    if (!terms || !out) return SCN_ERR_BAD_ARG;
    if (n <= 0 || (n & 3) || n_terms < 1 || n_terms > 4) return SCN_ERR_BAD_SHAPE;
    SumTerms T{};
    T.n = n_terms;
    for (int k = 0; k < n_terms; ++k) {
        if (!terms[k] || ((uintptr_t)terms[k] & 15)) return SCN_ERR_BAD_ARG;
        T.t[k] = terms[k];
    }
    if ((uintptr_t)out & 15) return SCN_ERR_BAD_ARG;
    const int64_t n4 = n / 4;
    const dim3 grid((int)std::min<int64_t>(8192, (n4 + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    switch (act) {
        case SCN_ACT_TANH: hipLaunchKernelGGL(sum_act_kernel<SCN_ACT_TANH>, grid, dim3(256), 0, st, n4, T, out); break;
        case SCN_ACT_RELU: hipLaunchKernelGGL(sum_act_kernel<SCN_ACT_RELU>, grid, dim3(256), 0, st, n4, T, out); break;
        case SCN_ACT_LEAKY_RELU: hipLaunchKernelGGL(sum_act_kernel<SCN_ACT_LEAKY_RELU>, grid, dim3(256), 0, st, n4, T, out); break;
        case SCN_ACT_NONE: hipLaunchKernelGGL(sum_act_kernel<SCN_ACT_NONE>, grid, dim3(256), 0, st, n4, T, out); break;
        default: return SCN_ERR_BAD_ARG;
    }
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

int scn_split_sign(int64_t n, const float* g, float* g_pos, float* g_neg, void* stream) {
    if (!g || !g_pos || !g_neg) return SCN_ERR_BAD_ARG;
    if (n <= 0) return SCN_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(split_sign_kernel, dim3((int)std::min<int64_t>(4096, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, g,
                       g_pos, g_neg);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

int scn_fold1_forward(const float* w1, const float* W2, int32_t c1, int32_t c2, float* a_pos, float* a_neg, void* stream) {
    if (!w1 || !W2 || !a_pos || !a_neg) return SCN_ERR_BAD_ARG;
    if (c1 <= 0 || c2 <= 0 || c2 > 1024) return SCN_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(fold1_forward_kernel, dim3(1), dim3((c2 + 63) / 64 * 64), 0, (hipStream_t)stream, w1, W2, c1, c2, a_pos, a_neg);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

int scn_fold1_backward(const float* w1, const float* W2, const float* u_pos, const float* u_neg, int32_t c1, int32_t c2,
                       float* dW2, float* dw1, void* stream) {
    if (!w1 || !W2 || !u_pos || !u_neg || !dW2 || !dw1) return SCN_ERR_BAD_ARG;
    if (c1 <= 0 || c2 <= 0 || c1 > 1024) return SCN_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(fold1_backward_kernel, dim3(1), dim3((c1 + 63) / 64 * 64), 0, (hipStream_t)stream, w1, W2, u_pos, u_neg, c1, c2,
                       dW2, dw1);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

}  // extern "C"
