// Dense per-point contractions over already-gathered terms (Bunch / SCCONV layers at scale, TE:181-195):
//   forward : out[p,:]  = act( sum_k G_k[p,:] @ W_k )                       G_k = S_k X_src(k)  (blocked SpMM)
//   backward: dx[p,:]   = ( sum_k G'_k[p,:] @ W_k^T ) * act'(aux[p,:])      G'_k = S_k^T dZ_dst(k)
//             dW_k     += sum_p aux[p,:]^T G'_k[p,:]                        (per-block partials, fixed-order reduce)
// p runs over every (slab, row, trajectory) point of one level; all tensors are [points][channels] fp32.
// VALU kernels (the three shifts of a Bunch level have different sources, so there is no shared gather to fuse with).
#include <algorithm>
#include <cstring>

#include "scn_internal.h"

namespace scn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int DN_MAX_TERMS = 3;
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

struct DenseReduceArgs {
    const float* partial;
    int32_t n_partials, total, n_terms;
    int32_t off[DN_MAX_TERMS + 1];
    float* dW[DN_MAX_TERMS];
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

static int dense_blocks(int64_t n_points) { return (int)std::min<int64_t>(2048, (n_points + DN_TILE - 1) / DN_TILE); }

}  // namespace scn

using namespace scn;

extern "C" {

int scn_dense_terms_forward(int64_t n_points, int32_t n_terms, const float* const* G, const int32_t* c_in,
                            const float* const* W, int32_t c_out, int32_t act, float* out, void* stream) {
    if (!G || !c_in || !W || !out) return SCN_ERR_BAD_ARG;
    if (n_points <= 0 || n_terms <= 0 || n_terms > DN_MAX_TERMS || c_out <= 0 || act < 0 || act > 3) return SCN_ERR_BAD_SHAPE;
    DenseFwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_points = n_points; a.n_terms = n_terms; a.c_out = c_out; a.act = act; a.out = out;
    size_t lds = 0;
    for (int k = 0; k < n_terms; ++k) {
        if (!G[k] || !W[k] || c_in[k] <= 0) return SCN_ERR_BAD_ARG;
        a.c_in[k] = c_in[k]; a.G[k] = G[k]; a.W[k] = W[k];
        lds += (size_t)c_in[k] * c_out * sizeof(float);
    }
    if (lds > 64 * 1024) return SCN_ERR_UNSUPPORTED;
    const int64_t total = n_points * ((c_out + 3) / 4);
    const int blocks = (int)std::min<int64_t>(8192, (total + DN_THREADS - 1) / DN_THREADS);
    hipLaunchKernelGGL(dense_fwd_kernel, dim3(blocks), dim3(DN_THREADS), lds, (hipStream_t)stream, a);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

size_t scn_dense_terms_backward_workspace(int64_t n_points, int32_t n_terms, const int32_t* c, int32_t c_aux) {
    if (!c || n_points <= 0 || n_terms <= 0 || n_terms > DN_MAX_TERMS) return 0;
    size_t tot = 0;
    for (int k = 0; k < n_terms; ++k) tot += (size_t)c_aux * c[k];
    return (size_t)dense_blocks(n_points) * tot * sizeof(float) + 256;
}

int scn_dense_terms_backward(int64_t n_points, int32_t n_terms, const float* const* G, const int32_t* c,
                             const float* const* W, const float* aux, int32_t c_aux, int32_t act, float* dx,
                             float* const* dW, void* workspace, size_t workspace_bytes, void* stream) {
    if (!G || !c || !W || !aux || !dW || !workspace) return SCN_ERR_BAD_ARG;
    if (n_points <= 0 || n_terms <= 0 || n_terms > DN_MAX_TERMS || c_aux <= 0 || act < 0 || act > 3) return SCN_ERR_BAD_SHAPE;
    if (workspace_bytes < scn_dense_terms_backward_workspace(n_points, n_terms, c, c_aux)) return SCN_ERR_WORKSPACE;
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
        a.c[k] = c[k]; a.G[k] = G[k]; a.W[k] = W[k];
        lds += (size_t)(c_aux * c[k] + DN_TILE * c[k]) * sizeof(float);
        r.off[k + 1] = r.off[k] + c_aux * c[k];
        r.dW[k] = dW[k];
    }
    if (lds > 64 * 1024) return SCN_ERR_UNSUPPORTED;
    const int nb = dense_blocks(n_points);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dense_bwd_kernel, dim3(nb), dim3(DN_THREADS), lds, st, a);
    SCN_LAUNCH_CHECK();
    r.partial = a.partial; r.n_partials = nb; r.total = r.off[n_terms]; r.n_terms = n_terms;
    hipLaunchKernelGGL(dense_dw_reduce, dim3((r.total + 255) / 256), dim3(256), 0, st, r);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

}  // extern "C"
