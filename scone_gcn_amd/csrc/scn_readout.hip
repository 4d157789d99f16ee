// Readout (+ log-softmax), its backward, flow scatter and the fused Adam/ridge step.
#include <cstring>

#include "scn_internal.h"

namespace scn {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

constexpr int RO_MAXD = 64;   // max neighbourhood width handled by one wave (lane d)

// ------------------------------------------------------------------------------------------------
// Readout (TE:151-152 with Bconds_func TE:298-303) and its gradient, one wave per trajectory.
// The work of a trajectory is a short ITEM list: (neighbour slot d, incident edge e of that neighbour, incidence sign).  It is
// built first, with the lanes over slots / items (a chain of three dependent loads in all: nbr -> inc_ptr -> inc_edge), and
// then consumed with the lanes over (item group, channel), every load independent of the others.  Walking slots and incident
// edges serially with all lanes on one item -- the first form of these kernels, kept below for item lists that do not fit --
// is a chain of ~4 dependent loads per item: 58 us per launch at |E| = 1001, which made the readout the dearest part of a
// small-complex optimiser step.
// ------------------------------------------------------------------------------------------------
constexpr int RO_ITEMS = 512;   // items a trajectory's list can hold (max_deg x incident edges; 13 x ~7 on the reference's complexes)

__device__ __forceinline__ int wave_excl_scan(int v, int lane) {
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    return x - v;
}

__device__ void readout_fwd_serial(int n, int lane, int ns, int n_edges, int c, const float* __restrict__ H,
                                   const float* __restrict__ w, const int32_t* __restrict__ nbr, int max_deg, int vlast,
                                   const int32_t* __restrict__ inc_ptr, const int32_t* __restrict__ inc_edge,
                                   const float* __restrict__ inc_sign, float* __restrict__ bh, float* __restrict__ logits,
                                   float* __restrict__ logp) {
    const int s = n / ns, i = n - s * ns;
    float my_logit = 0.f;   // lane d keeps logit d
    for (int d = 0; d < max_deg; ++d) {
        const int v = nbr[(size_t)vlast * max_deg + d];
        float lg = 0.f;
        for (int c0 = 0; c0 < c; c0 += 64) {
            const int cc = c0 + lane;
            float acc = 0.f;
            if (v >= 0 && cc < c) {
                for (int j = inc_ptr[v]; j < inc_ptr[v + 1]; ++j) {
                    const int e = inc_edge[j];
                    acc = fmaf(inc_sign[j], H[(((size_t)s * n_edges + e) * ns + i) * c + cc], acc);
                }
            }
            if (cc < c) {
                bh[((size_t)n * max_deg + d) * c + cc] = acc;
                lg = fmaf(acc, w[cc], lg);
            }
        }
        lg = wave_sum(lg);
        if (lane == d) my_logit = lg;
    }
    const float x = lane < max_deg ? my_logit : -INFINITY;
    const float m = wave_max(x);
    const float se = wave_sum(lane < max_deg ? expf(x - m) : 0.f);
    const float lse = m + logf(se);
    if (lane < max_deg) {
        logits[(size_t)n * max_deg + lane] = my_logit;
        logp[(size_t)n * max_deg + lane] = my_logit - lse;
    }
}

__global__ __launch_bounds__(64) void readout_fwd_kernel(int ns, int n_edges, int c, const float* __restrict__ H,
                                                         const float* __restrict__ w, const int32_t* __restrict__ nbr,
                                                         int max_deg, const int32_t* __restrict__ last_nodes,
                                                         const int32_t* __restrict__ inc_ptr,
                                                         const int32_t* __restrict__ inc_edge,
                                                         const float* __restrict__ inc_sign,
                                                         float* __restrict__ bh, float* __restrict__ logits,
                                                         float* __restrict__ logp) {
    __shared__ int it_e[RO_ITEMS];
    __shared__ float it_s[RO_ITEMS];
    __shared__ int d_ptr[RO_MAXD + 1];
    __shared__ float lgs[RO_MAXD];
    const int n = blockIdx.x, lane = threadIdx.x;
    const int s = n / ns, i = n - s * ns;
    const int vlast = last_nodes[n];
    int start = 0, cnt = 0;
    if (lane < max_deg) {
        const int v = nbr[(size_t)vlast * max_deg + lane];
        if (v >= 0) {
            start = inc_ptr[v];
            cnt = inc_ptr[v + 1] - start;
        }
    }
    const int off = wave_excl_scan(cnt, lane);
    const int total = __shfl(off + cnt, 63, 64);
    if (total > RO_ITEMS || c > 64) {                       // (wave-uniform)
        readout_fwd_serial(n, lane, ns, n_edges, c, H, w, nbr, max_deg, vlast, inc_ptr, inc_edge, inc_sign, bh, logits, logp);
        return;
    }
    if (lane < max_deg) d_ptr[lane] = off;
    if (lane == 0) d_ptr[max_deg] = total;
    for (int j = 0; j < cnt; ++j) {
        it_e[off + j] = inc_edge[start + j];
        it_s[off + j] = inc_sign[start + j];
    }
    __syncthreads();
    const int cpad = c <= 16 ? 16 : (c <= 32 ? 32 : 64), G = 64 / cpad;
    const int g = lane / cpad, cc = lane - g * cpad;
    const float wc = cc < c ? w[cc] : 0.f;
    const float* Hn = H + ((size_t)s * n_edges * ns + i) * c + cc;               // + e * ns * c
    const size_t estride = (size_t)ns * c;
    for (int d0 = 0; d0 < max_deg; d0 += G) {
        const int d = d0 + g;
        const bool live = d < max_deg && cc < c;
        float acc = 0.f;
        if (live) {
            const int t0 = d_ptr[d], t1 = d_ptr[d + 1];
            for (int t = t0; t < t1; t += 4) {             // four independent loads per trip, summed in list order
                float h[4], sg[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int tu = t + u < t1 ? t + u : t;
                    h[u] = Hn[(size_t)it_e[tu] * estride];
                    sg[u] = t + u < t1 ? it_s[tu] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (t + u < t1) acc = fmaf(sg[u], h[u], acc);
            }
            bh[((size_t)n * max_deg + d) * c + cc] = acc;
        }
        float lg = live ? acc * wc : 0.f;
        for (int o = cpad >> 1; o > 0; o >>= 1) lg += __shfl_xor(lg, o, 64);      // inside the aligned cpad-lane group
        if (cc == 0 && d < max_deg) lgs[d] = lg;
    }
    __syncthreads();
    const float my_logit = lane < max_deg ? lgs[lane] : 0.f;
    const float x = lane < max_deg ? my_logit : -INFINITY;
    const float m = wave_max(x);
    const float se = wave_sum(lane < max_deg ? expf(x - m) : 0.f);
    const float lse = m + logf(se);
    if (lane < max_deg) {
        logits[(size_t)n * max_deg + lane] = my_logit;
        logp[(size_t)n * max_deg + lane] = my_logit - lse;
    }
}

__device__ void readout_bwd_serial(int n, int lane, int ns, int n_edges, int c, const float* __restrict__ H,
                                   const float* __restrict__ w, int max_deg, const int32_t* __restrict__ inc_ptr,
                                   const int32_t* __restrict__ inc_edge, const float* __restrict__ inc_sign,
                                   const int32_t* __restrict__ edge_nodes, int act, float* __restrict__ dz, int clear,
                                   const float* dl, const int* nb) {
    const int s = n / ns, i = n - s * ns;
    for (int d = 0; d < max_deg; ++d) {
        const int v = nb[d];
        if (v < 0) continue;
        for (int j = inc_ptr[v]; j < inc_ptr[v + 1]; ++j) {
            const int e = inc_edge[j];
            const int t = edge_nodes[2 * e], h = edge_nodes[2 * e + 1];
            const int other = (t == v) ? h : t;
            int dk = -1;
            for (int q = 0; q < max_deg; ++q)
                if (nb[q] == other) dk = q;
            if (dk >= 0 && other < v) continue;          // handled from the other endpoint's side
            // the two endpoints carry opposite incidence signs (B1[tail]=-1, B1[head]=+1; a flip scales both)
            const size_t base = (((size_t)s * n_edges + e) * ns + i) * c;
            if (clear) {
                for (int cc = lane; cc < c; cc += 64) dz[base + cc] = 0.f;
                continue;
            }
            const float coef = inc_sign[j] * (dl[d] - (dk >= 0 ? dl[dk] : 0.f));
            for (int cc = lane; cc < c; cc += 64)
                dz[base + cc] = coef * w[cc] * act_grad_from_output(act, H[base + cc]);
        }
    }
}

// zero dz[s][r][i][0 .. c) for every row r (the whole workgroup; 16-byte stores when the width allows), and wait for the stores:
// wave 0 writes some of these entries right afterwards (the caller puts a barrier between)
__device__ __forceinline__ void readout_zero_column(float* __restrict__ dz, int s, int i, int ns, int n_edges, int c, int lane, int nt) {
    float* base = dz + ((size_t)s * n_edges * ns + i) * c;
    const size_t estride = (size_t)ns * c;
    if ((c & 3) == 0) {
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        const int q = c >> 2;
        for (int t = lane; t < n_edges * q; t += nt) {
            const int r = t / q, k = t - r * q;
            *(f32x4_*)(base + (size_t)r * estride + 4 * k) = f32x4_{0.f, 0.f, 0.f, 0.f};
        }
    } else {
        for (int t = lane; t < n_edges * c; t += nt) {
            const int r = t / c, k = t - r * c;
            base[(size_t)r * estride + k] = 0.f;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);                   // vmcnt(0) expcnt(0) lgkmcnt(0)
    __threadfence_block();
}

constexpr int RO_ZERO_THREADS = 512;           // clear == 2: eight waves zero the column, wave 0 goes on alone
__global__ __launch_bounds__(RO_ZERO_THREADS) void readout_bwd_kernel(int ns, int n_edges, int c, const float* __restrict__ H,
                                                         const float* __restrict__ w, const int32_t* __restrict__ nbr,
                                                         int max_deg, const int32_t* __restrict__ last_nodes,
                                                         const int32_t* __restrict__ inc_ptr,
                                                         const int32_t* __restrict__ inc_edge,
                                                         const float* __restrict__ inc_sign,
                                                         const int32_t* __restrict__ edge_nodes,
                                                         const float* __restrict__ d_logp,
                                                         const float* __restrict__ logp, int act,
                                                         float* __restrict__ dz, float* __restrict__ dl_out, int clear) {
    // clear == 1: write zeros to exactly the dz entries the normal pass writes (scn_readout_clear_dz)
    // clear == 2: the workgroup (RO_ZERO_THREADS threads in this form, 64 otherwise) first zeroes its trajectory's whole column of dz
    //             ([s][all rows][i][:]: nobody else writes there), then wave 0 runs the normal pass alone -- small complexes: no fill
    //             launch over the gradient buffer, which may hold anything
    __shared__ float dl[RO_MAXD];
    __shared__ int nb[RO_MAXD];
    __shared__ int d_ptr[RO_MAXD + 1], d_start[RO_MAXD];
    __shared__ int it_e[RO_ITEMS];
    __shared__ float it_coef[RO_ITEMS];
    __shared__ unsigned char it_d[RO_ITEMS];
    const int n = blockIdx.x, lane = threadIdx.x;
    const int s = n / ns, i = n - s * ns;
    if (clear == 2) {
        readout_zero_column(dz, s, i, ns, n_edges, c, threadIdx.x, blockDim.x);
        __syncthreads();
        clear = 0;
        if (threadIdx.x >= 64) return;                 // (a finished wave no longer takes part in the barriers below)
    }
    const int vlast = last_nodes[n];
    const float g = (!clear && lane < max_deg) ? d_logp[(size_t)n * max_deg + lane] : 0.f;
    const float gs = wave_sum(g);
    int start = 0, cnt = 0;
    if (lane < max_deg) {
        if (!clear) {
            const float v = g - expf(logp[(size_t)n * max_deg + lane]) * gs;
            dl[lane] = v;
            dl_out[(size_t)n * max_deg + lane] = v;
        }
        const int v = nbr[(size_t)vlast * max_deg + lane];
        nb[lane] = v;
        if (v >= 0) {
            start = inc_ptr[v];
            cnt = inc_ptr[v + 1] - start;
        }
    }
    const int off = wave_excl_scan(cnt, lane);
    const int total = __shfl(off + cnt, 63, 64);
    __syncthreads();
    if (total > RO_ITEMS || c > 64) {                       // (wave-uniform)
        readout_bwd_serial(n, lane, ns, n_edges, c, H, w, max_deg, inc_ptr, inc_edge, inc_sign, edge_nodes, act, dz, clear, dl, nb);
        return;
    }
    if (lane < max_deg) {
        d_ptr[lane] = off;
        d_start[lane] = start;
    }
    for (int j = 0; j < cnt; ++j) it_d[off + j] = (unsigned char)lane;
    __syncthreads();
    for (int t = lane; t < total; t += 64) {               // one item per lane: edge, skip rule, coefficient
        const int d = it_d[t], v = nb[d];
        const int j = d_start[d] + (t - d_ptr[d]);
        const int e = inc_edge[j];
        const int tl = edge_nodes[2 * e], hd = edge_nodes[2 * e + 1];
        const int other = (tl == v) ? hd : tl;
        int dk = -1;
        for (int q = 0; q < max_deg; ++q)
            if (nb[q] == other) dk = q;
        const bool skip = dk >= 0 && other < v;            // handled from the other endpoint's side
        it_e[t] = skip ? -1 : e;
        // the two endpoints carry opposite incidence signs (B1[tail]=-1, B1[head]=+1; a flip scales both)
        it_coef[t] = clear ? 0.f : inc_sign[j] * (dl[d] - (dk >= 0 ? dl[dk] : 0.f));
    }
    __syncthreads();
    const int cpad = c <= 16 ? 16 : (c <= 32 ? 32 : 64), G = 64 / cpad;
    const int gq = lane / cpad, cc = lane - gq * cpad;
    const float wc = (!clear && cc < c) ? w[cc] : 0.f;
    const size_t nbase = ((size_t)s * n_edges * ns + i) * c + cc, estride = (size_t)ns * c;
    for (int t = gq; t < total; t += G) {
        const int e = it_e[t];
        if (e < 0 || cc >= c) continue;
        const size_t o = nbase + (size_t)e * estride;
        dz[o] = clear ? 0.f : it_coef[t] * wc * act_grad_from_output(act, H[o]);
    }
}

// d_w[c] += sum_{n,d} dl[n,d] * bh[n,d,c]: one block; thread = (row group, channel) with 1024 / cpad row groups, each summing its
// rows in order; the row groups are combined in a fixed order
__global__ __launch_bounds__(1024) void readout_dw_kernel(int nd, int c, const float* __restrict__ dl,
                                                          const float* __restrict__ bh, float* __restrict__ d_w) {
    __shared__ float part[1024];
    for (int c0 = 0; c0 < c; c0 += 64) {
        const int cw = c - c0 < 64 ? c - c0 : 64;
        const int cpad = cw <= 16 ? 16 : (cw <= 32 ? 32 : 64), K = 1024 / cpad;
        const int k = threadIdx.x / cpad, cc = threadIdx.x - k * cpad;
        float acc = 0.f;
        if (cc < cw) {
            int q = k;
            for (; q + 3 * K < nd; q += 4 * K) {
                const float a0 = dl[q], a1 = dl[q + K], a2 = dl[q + 2 * K], a3 = dl[q + 3 * K];
                const float b0 = bh[(size_t)q * c + c0 + cc], b1 = bh[(size_t)(q + K) * c + c0 + cc],
                            b2 = bh[(size_t)(q + 2 * K) * c + c0 + cc], b3 = bh[(size_t)(q + 3 * K) * c + c0 + cc];
                acc = fmaf(a0, b0, acc); acc = fmaf(a1, b1, acc); acc = fmaf(a2, b2, acc); acc = fmaf(a3, b3, acc);
            }
            for (; q < nd; q += K) acc = fmaf(dl[q], bh[(size_t)q * c + c0 + cc], acc);
        }
        part[threadIdx.x] = acc;
        __syncthreads();
        if (k == 0 && cc < cw) {
            float sum = 0.f;
            for (int j = 0; j < K; ++j) sum += part[j * cpad + cc];
            d_w[c0 + cc] += sum;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void node_readout_fwd_kernel(int ns, int n_nodes, const float* __restrict__ X,
                                                              const int32_t* __restrict__ nbr, int max_deg,
                                                              const int32_t* __restrict__ last_nodes,
                                                              float* __restrict__ logits, float* __restrict__ logp) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const int s = n / ns, i = n - s * ns;
    const int vlast = last_nodes[n];
    float x = -INFINITY, lg = 0.f;
    if (lane < max_deg) {
        int v = nbr[(size_t)vlast * max_deg + lane];
        if (v < 0) v += n_nodes;                      // index -1 wraps to the last node (TE:201)
        lg = X[((size_t)s * n_nodes + v) * ns + i];
        x = lg;
    }
    const float m = wave_max(x);
    const float se = wave_sum(lane < max_deg ? expf(x - m) : 0.f);
    const float lse = m + logf(se);
    if (lane < max_deg) {
        logits[(size_t)n * max_deg + lane] = lg;
        logp[(size_t)n * max_deg + lane] = lg - lse;
    }
}

__global__ __launch_bounds__(64) void node_readout_bwd_kernel(int ns, int n_nodes, const float* __restrict__ X,
                                                              const int32_t* __restrict__ nbr, int max_deg,
                                                              const int32_t* __restrict__ last_nodes,
                                                              const float* __restrict__ d_logp,
                                                              const float* __restrict__ logp, int act,
                                                              float* __restrict__ dz) {
    __shared__ float dl[RO_MAXD];
    const int n = blockIdx.x, lane = threadIdx.x;
    const int s = n / ns, i = n - s * ns;
    const int vlast = last_nodes[n];
    const float g = lane < max_deg ? d_logp[(size_t)n * max_deg + lane] : 0.f;
    const float gs = wave_sum(g);
    if (lane < max_deg) dl[lane] = g - expf(logp[(size_t)n * max_deg + lane]) * gs;
    __syncthreads();
    if (lane == 0) {                                   // serial: several padded entries may hit the same node
        for (int d = 0; d < max_deg; ++d) {
            int v = nbr[(size_t)vlast * max_deg + d];
            if (v < 0) v += n_nodes;
            const size_t o = ((size_t)s * n_nodes + v) * ns + i;
            dz[o] += dl[d] * act_grad_from_output(act, X[o]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same four kernels for neighbourhoods wider than one wave's lanes (RO_MAXD < max_deg <= RO_WIDE): hub nodes.  One wave per
// trajectory as before; the per-slot values (logits, d_logits, neighbour ids) live in LDS arrays of RO_WIDE entries and the lanes
// stride over the slots.  Plain loops: a complex with such a node is rare and its readout is still a sliver of the step.
// ------------------------------------------------------------------------------------------------
constexpr int RO_WIDE = 1024;

__device__ __forceinline__ float lds_logsumexp(const float* vals, int n, int lane) {
    float m = -INFINITY;
    for (int d = lane; d < n; d += 64) m = fmaxf(m, vals[d]);
    m = wave_max(m);
    float se = 0.f;
    for (int d = lane; d < n; d += 64) se += expf(vals[d] - m);
    se = wave_sum(se);
    return m + logf(se);
}

__global__ __launch_bounds__(64) void readout_fwd_wide_kernel(int ns, int n_edges, int c, const float* __restrict__ H,
                                                              const float* __restrict__ w, const int32_t* __restrict__ nbr,
                                                              int max_deg, const int32_t* __restrict__ last_nodes,
                                                              const int32_t* __restrict__ inc_ptr,
                                                              const int32_t* __restrict__ inc_edge,
                                                              const float* __restrict__ inc_sign, float* __restrict__ bh,
                                                              float* __restrict__ logits, float* __restrict__ logp) {
    __shared__ float lgs[RO_WIDE];
    const int n = blockIdx.x, lane = threadIdx.x;
    const int s = n / ns, i = n - s * ns;
    const int vlast = last_nodes[n];
    for (int d = 0; d < max_deg; ++d) {
        const int v = nbr[(size_t)vlast * max_deg + d];
        float lg = 0.f;
        for (int c0 = 0; c0 < c; c0 += 64) {
            const int cc = c0 + lane;
            float acc = 0.f;
            if (v >= 0 && cc < c)
                for (int j = inc_ptr[v]; j < inc_ptr[v + 1]; ++j)
                    acc = fmaf(inc_sign[j], H[(((size_t)s * n_edges + inc_edge[j]) * ns + i) * c + cc], acc);
            if (cc < c) {
                bh[((size_t)n * max_deg + d) * c + cc] = acc;
                lg = fmaf(acc, w[cc], lg);
            }
        }
        lg = wave_sum(lg);
        if (lane == 0) lgs[d] = lg;
    }
    __syncthreads();
    const float lse = lds_logsumexp(lgs, max_deg, lane);       // padding rows (logit 0) take part, as in the reference (TE:151-152)
    for (int d = lane; d < max_deg; d += 64) {
        logits[(size_t)n * max_deg + d] = lgs[d];
        logp[(size_t)n * max_deg + d] = lgs[d] - lse;
    }
}

__global__ __launch_bounds__(64) void readout_bwd_wide_kernel(int ns, int n_edges, int c, const float* __restrict__ H,
                                                              const float* __restrict__ w, const int32_t* __restrict__ nbr,
                                                              int max_deg, const int32_t* __restrict__ last_nodes,
                                                              const int32_t* __restrict__ inc_ptr,
                                                              const int32_t* __restrict__ inc_edge,
                                                              const float* __restrict__ inc_sign,
                                                              const int32_t* __restrict__ edge_nodes,
                                                              const float* __restrict__ d_logp,
                                                              const float* __restrict__ logp, int act,
                                                              float* __restrict__ dz, float* __restrict__ dl_out, int clear) {
    __shared__ float dl[RO_WIDE];
    __shared__ int nb[RO_WIDE];
    const int n = blockIdx.x, lane = threadIdx.x;
    const int s = n / ns, i = n - s * ns;
    const int vlast = last_nodes[n];
    float g = 0.f;
    if (!clear)
        for (int d = lane; d < max_deg; d += 64) g += d_logp[(size_t)n * max_deg + d];
    const float gs = wave_sum(g);
    for (int d = lane; d < max_deg; d += 64) {
        if (!clear) {
            const float v = d_logp[(size_t)n * max_deg + d] - expf(logp[(size_t)n * max_deg + d]) * gs;
            dl[d] = v;
            dl_out[(size_t)n * max_deg + d] = v;
        } else {
            dl[d] = 0.f;
        }
        nb[d] = nbr[(size_t)vlast * max_deg + d];
    }
    __syncthreads();
    for (int d = 0; d < max_deg; ++d) {
        const int v = nb[d];
        if (v < 0) continue;
        for (int j = inc_ptr[v]; j < inc_ptr[v + 1]; ++j) {
            const int e = inc_edge[j];
            const int t = edge_nodes[2 * e], h = edge_nodes[2 * e + 1];
            const int other = (t == v) ? h : t;
            int dk = -1;                                       // slot of the edge's other endpoint, if it is a neighbour too (the last such slot)
            for (int q = lane; q < max_deg; q += 64)
                if (nb[q] == other) dk = q;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) dk = max(dk, __shfl_xor(dk, o, 64));
            if (dk >= 0 && other < v) continue;                // handled from the other endpoint's side
            const size_t base = (((size_t)s * n_edges + e) * ns + i) * c;
            if (clear) {
                for (int cc = lane; cc < c; cc += 64) dz[base + cc] = 0.f;
                continue;
            }
            const float coef = inc_sign[j] * (dl[d] - (dk >= 0 ? dl[dk] : 0.f));
            for (int cc = lane; cc < c; cc += 64) dz[base + cc] = coef * w[cc] * act_grad_from_output(act, H[base + cc]);
        }
    }
}

__global__ __launch_bounds__(64) void node_readout_fwd_wide_kernel(int ns, int n_nodes, const float* __restrict__ X,
                                                                   const int32_t* __restrict__ nbr, int max_deg,
                                                                   const int32_t* __restrict__ last_nodes,
                                                                   float* __restrict__ logits, float* __restrict__ logp) {
    __shared__ float lgs[RO_WIDE];
    const int n = blockIdx.x, lane = threadIdx.x;
    const int s = n / ns, i = n - s * ns;
    const int vlast = last_nodes[n];
    for (int d = lane; d < max_deg; d += 64) {
        int v = nbr[(size_t)vlast * max_deg + d];
        if (v < 0) v += n_nodes;                      // index -1 wraps to the last node (TE:201)
        lgs[d] = X[((size_t)s * n_nodes + v) * ns + i];
    }
    __syncthreads();
    const float lse = lds_logsumexp(lgs, max_deg, lane);
    for (int d = lane; d < max_deg; d += 64) {
        logits[(size_t)n * max_deg + d] = lgs[d];
        logp[(size_t)n * max_deg + d] = lgs[d] - lse;
    }
}

__global__ __launch_bounds__(64) void node_readout_bwd_wide_kernel(int ns, int n_nodes, const float* __restrict__ X,
                                                                   const int32_t* __restrict__ nbr, int max_deg,
                                                                   const int32_t* __restrict__ last_nodes,
                                                                   const float* __restrict__ d_logp,
                                                                   const float* __restrict__ logp, int act,
                                                                   float* __restrict__ dz) {
    __shared__ float dl[RO_WIDE];
    const int n = blockIdx.x, lane = threadIdx.x;
    const int s = n / ns, i = n - s * ns;
    const int vlast = last_nodes[n];
    float g = 0.f;
    for (int d = lane; d < max_deg; d += 64) g += d_logp[(size_t)n * max_deg + d];
    const float gs = wave_sum(g);
    for (int d = lane; d < max_deg; d += 64) dl[d] = d_logp[(size_t)n * max_deg + d] - expf(logp[(size_t)n * max_deg + d]) * gs;
    __syncthreads();
    if (lane == 0) {                                   // serial: several padded entries may hit the same node
        for (int d = 0; d < max_deg; ++d) {
            int v = nbr[(size_t)vlast * max_deg + d];
            if (v < 0) v += n_nodes;
            const size_t o = ((size_t)s * n_nodes + v) * ns + i;
            dz[o] += dl[d] * act_grad_from_output(act, X[o]);
        }
    }
}

__global__ void scatter_flows_kernel(int ns, int n_edges, int64_t n_entries, const int32_t* __restrict__ sample_of,
                                     const int32_t* __restrict__ edge_idx, const float* __restrict__ val,
                                     float* __restrict__ x) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_entries) return;
    const int n = sample_of[t];
    const int s = n / ns, i = n - s * ns;
    // accumulate: the reference builds a flow with f[k] += +-1 per traversed edge (SDG:327-344), so a repeated
    // (trajectory, edge) entry adds up -- like SparseFlows.todense() on the host (x is zeroed just before)
    atomicAdd(&x[((size_t)s * n_edges + edge_idx[t]) * ns + i], val[t]);
}

// masked cross-entropy of STM:54 for one micro-batch: d_logp = y * scale (scale = -1 / global batch size) and
// loss[0] += sum logp * d_logp in fp64, one block, fixed summation order
// logp[n][:] = z - logsumexp(z), z = sum over the parts of logits_k[n][:]  (one wave per trajectory, lanes stride over the slots)
struct LogitParts { const float* p[4]; };
__global__ __launch_bounds__(64) void logits_sum_log_softmax_kernel(int max_deg, int n_parts, LogitParts L, float* __restrict__ logits,
                                                                    float* __restrict__ logp) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const size_t base = (size_t)n * max_deg;
    float m = -INFINITY;
    for (int d = lane; d < max_deg; d += 64) {
        float z = L.p[0][base + d];
        if (n_parts > 1) z += L.p[1][base + d];
        if (n_parts > 2) z += L.p[2][base + d];
        if (n_parts > 3) z += L.p[3][base + d];
        logits[base + d] = z;
        m = fmaxf(m, z);
    }
    m = wave_max(m);
    float se = 0.f;
    for (int d = lane; d < max_deg; d += 64) se += expf(logits[base + d] - m);       // (this lane's own stores)
    se = wave_sum(se);
    const float lse = m + logf(se);
    for (int d = lane; d < max_deg; d += 64) logp[base + d] = logits[base + d] - lse;
}

__global__ __launch_bounds__(1024) void masked_ce_kernel(int64_t n, const float* __restrict__ logp,
                                                         const float* __restrict__ y, float scale,
                                                         float* __restrict__ d_logp, double* __restrict__ loss,
                                                         int overwrite = 0, float* __restrict__ zero_buf = nullptr, int64_t zero_n = 0) {
    __shared__ double part[1024];
    for (int64_t t = threadIdx.x; t < zero_n; t += 1024) zero_buf[t] = 0.f;
    double acc = 0.0;
    for (int64_t t = threadIdx.x; t < n; t += 1024) {
        const float d = y[t] * scale;
        d_logp[t] = d;
        acc += (double)logp[t] * (double)d;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = overwrite ? part[0] : loss[0] + part[0];
}

__global__ void adam_kernel(int64_t n, float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, float lr, float b1, float b2, float eps, float c1, float c2,
                            float wd2, float g_scale) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    adam_update(w + t, g[t], m + t, v + t, lr, b1, b2, eps, c1, c2, wd2, g_scale);
}

// The same update (adam_update, scn_internal.h) with the step index in DEVICE memory: one workgroup (the flat buffer of every model of the reference is a few
// thousand floats), which reads step[0] = i, applies update i and leaves i + 1 there -- a launch whose arguments never change, so
// the optimiser step can sit inside a captured graph behind the gradient launches (on the reference's own problem sizes the
// hand-over from a graph replay to a plain launch is ~8 us of an ~80 us step).
constexpr int ADAM_DEV_THREADS = 1024;
__global__ __launch_bounds__(ADAM_DEV_THREADS) void adam_dev_kernel(int64_t n, float* __restrict__ w, const float* __restrict__ g,
                                                                  float* __restrict__ m, float* __restrict__ v, float lr, float b1,
                                                                  float b2, float eps, int32_t* step, float wd2, float g_scale) {
    __shared__ float c[2];
    const int i = step[0];                                       // (every thread, before the barrier: thread 0 writes it after the last one)
    if (threadIdx.x == 0) adam_corrections(b1, b2, i, c[0], c[1]);
    __syncthreads();
    const float c1 = c[0], c2 = c[1];
    for (int64_t t = threadIdx.x; t < n; t += ADAM_DEV_THREADS) adam_update(w + t, g[t], m + t, v + t, lr, b1, b2, eps, c1, c2, wd2, g_scale);
    __syncthreads();
    if (threadIdx.x == 0) step[0] = i + 1;
}

}  // namespace scn

using namespace scn;

extern "C" {

int scn_readout_forward(int32_t n_slabs, int32_t ns, int32_t n_edges, int32_t c, const float* H,
                        const float* w_last, const int32_t* nbr, int32_t n_nodes, int32_t max_deg,
                        const int32_t* last_nodes, const int32_t* inc_ptr, const int32_t* inc_edge,
                        const float* inc_sign, float* bh, float* logits, float* logp, void* stream) {
    if (!H || !w_last || !nbr || !last_nodes || !inc_ptr || !inc_edge || !inc_sign || !bh || !logits || !logp)
        return SCN_ERR_BAD_ARG;
    if (n_slabs <= 0 || ns <= 0 || n_edges <= 0 || c <= 0 || n_nodes <= 0 || max_deg <= 0) return SCN_ERR_BAD_SHAPE;
    if (max_deg > RO_WIDE) return SCN_ERR_UNSUPPORTED;
    if (max_deg > RO_MAXD)
        hipLaunchKernelGGL(readout_fwd_wide_kernel, dim3(n_slabs * ns), dim3(64), 0, (hipStream_t)stream, ns, n_edges, c, H,
                           w_last, nbr, max_deg, last_nodes, inc_ptr, inc_edge, inc_sign, bh, logits, logp);
    else
        hipLaunchKernelGGL(readout_fwd_kernel, dim3(n_slabs * ns), dim3(64), 0, (hipStream_t)stream, ns, n_edges, c, H,
                           w_last, nbr, max_deg, last_nodes, inc_ptr, inc_edge, inc_sign, bh, logits, logp);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

int scn_readout_backward(int32_t n_slabs, int32_t ns, int32_t n_edges, int32_t c, const float* H,
                         const float* w_last, const int32_t* nbr, int32_t n_nodes, int32_t max_deg,
                         const int32_t* last_nodes, const int32_t* inc_ptr, const int32_t* inc_edge,
                         const float* inc_sign, const int32_t* edge_nodes, const float* bh, const float* d_logp,
                         const float* logp, int32_t act, float* d_logits, float* dz, int32_t dz_is_zero,
                         float* d_w_last, void* stream) {
    if (!H || !w_last || !nbr || !last_nodes || !inc_ptr || !inc_edge || !inc_sign || !edge_nodes || !bh ||
        !d_logp || !logp || !d_logits || !dz || !d_w_last)
        return SCN_ERR_BAD_ARG;
    if (n_slabs <= 0 || ns <= 0 || n_edges <= 0 || c <= 0 || n_nodes <= 0 || max_deg <= 0) return SCN_ERR_BAD_SHAPE;
    if (max_deg > RO_WIDE) return SCN_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int N = n_slabs * ns;
    const bool self_zero = dz_is_zero == 2 && max_deg <= RO_MAXD;          // (the kernel of wide neighbourhoods has no such form)
    if (!dz_is_zero || (dz_is_zero == 2 && !self_zero)) SCN_HIP_TRY(hipMemsetAsync(dz, 0, sizeof(float) * (size_t)N * n_edges * c, st));
    float* dl = d_logits;
    if (max_deg > RO_MAXD)
        hipLaunchKernelGGL(readout_bwd_wide_kernel, dim3(N), dim3(64), 0, st, ns, n_edges, c, H, w_last, nbr, max_deg,
                           last_nodes, inc_ptr, inc_edge, inc_sign, edge_nodes, d_logp, logp, act, dz, dl, 0);
    else
        hipLaunchKernelGGL(readout_bwd_kernel, dim3(N), dim3(self_zero ? RO_ZERO_THREADS : 64), 0, st, ns, n_edges, c, H, w_last, nbr,
                           max_deg, last_nodes, inc_ptr, inc_edge, inc_sign, edge_nodes, d_logp, logp, act, dz, dl, self_zero ? 2 : 0);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(readout_dw_kernel, dim3(1), dim3(1024), 0, st, N * max_deg, c, dl, bh, d_w_last);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

int scn_readout_clear_dz(int32_t n_slabs, int32_t ns, int32_t n_edges, int32_t c, const int32_t* nbr, int32_t n_nodes,
                         int32_t max_deg, const int32_t* last_nodes, const int32_t* inc_ptr, const int32_t* inc_edge,
                         const int32_t* edge_nodes, float* dz, void* stream) {
    if (!nbr || !last_nodes || !inc_ptr || !inc_edge || !edge_nodes || !dz) return SCN_ERR_BAD_ARG;
    if (n_slabs <= 0 || ns <= 0 || n_edges <= 0 || c <= 0 || n_nodes <= 0 || max_deg <= 0) return SCN_ERR_BAD_SHAPE;
    if (max_deg > RO_WIDE) return SCN_ERR_UNSUPPORTED;
    if (max_deg > RO_MAXD)
        hipLaunchKernelGGL(readout_bwd_wide_kernel, dim3(n_slabs * ns), dim3(64), 0, (hipStream_t)stream, ns, n_edges, c, nullptr,
                           nullptr, nbr, max_deg, last_nodes, inc_ptr, inc_edge, nullptr, edge_nodes, nullptr, nullptr, 0, dz,
                           nullptr, 1);
    else
        hipLaunchKernelGGL(readout_bwd_kernel, dim3(n_slabs * ns), dim3(64), 0, (hipStream_t)stream, ns, n_edges, c, nullptr,
                           nullptr, nbr, max_deg, last_nodes, inc_ptr, inc_edge, nullptr, edge_nodes, nullptr, nullptr, 0, dz,
                           nullptr, 1);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

int scn_node_readout_forward(int32_t n_slabs, int32_t ns, int32_t n_nodes, const float* nodes_out,
                             const int32_t* nbr, int32_t max_deg, const int32_t* last_nodes, float* logits,
                             float* logp, void* stream) {
    if (!nodes_out || !nbr || !last_nodes || !logits || !logp) return SCN_ERR_BAD_ARG;
    if (n_slabs <= 0 || ns <= 0 || n_nodes <= 0 || max_deg <= 0) return SCN_ERR_BAD_SHAPE;
    if (max_deg > RO_WIDE) return SCN_ERR_UNSUPPORTED;
    if (max_deg > RO_MAXD)
        hipLaunchKernelGGL(node_readout_fwd_wide_kernel, dim3(n_slabs * ns), dim3(64), 0, (hipStream_t)stream, ns, n_nodes,
                           nodes_out, nbr, max_deg, last_nodes, logits, logp);
    else
        hipLaunchKernelGGL(node_readout_fwd_kernel, dim3(n_slabs * ns), dim3(64), 0, (hipStream_t)stream, ns, n_nodes,
                           nodes_out, nbr, max_deg, last_nodes, logits, logp);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

int scn_node_readout_backward(int32_t n_slabs, int32_t ns, int32_t n_nodes, const float* nodes_out,
                              const int32_t* nbr, int32_t max_deg, const int32_t* last_nodes, const float* d_logp,
                              const float* logp, int32_t act, float* dz, void* stream) {
    if (!nodes_out || !nbr || !last_nodes || !d_logp || !logp || !dz) return SCN_ERR_BAD_ARG;
    if (n_slabs <= 0 || ns <= 0 || n_nodes <= 0 || max_deg <= 0) return SCN_ERR_BAD_SHAPE;
    if (max_deg > RO_WIDE) return SCN_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int N = n_slabs * ns;
    SCN_HIP_TRY(hipMemsetAsync(dz, 0, sizeof(float) * (size_t)N * n_nodes, st));
    if (max_deg > RO_MAXD)
        hipLaunchKernelGGL(node_readout_bwd_wide_kernel, dim3(N), dim3(64), 0, st, ns, n_nodes, nodes_out, nbr, max_deg,
                           last_nodes, d_logp, logp, act, dz);
    else
        hipLaunchKernelGGL(node_readout_bwd_kernel, dim3(N), dim3(64), 0, st, ns, n_nodes, nodes_out, nbr, max_deg,
                           last_nodes, d_logp, logp, act, dz);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

int scn_scatter_flows(int32_t n_slabs, int32_t ns, int32_t n_edges, int64_t n_entries, const int32_t* sample_of,
                      const int32_t* edge_idx, const float* val, float* x, void* stream) {
    if (n_slabs <= 0 || ns <= 0 || n_edges <= 0 || n_entries < 0) return SCN_ERR_BAD_SHAPE;
    if (!x || (n_entries > 0 && (!sample_of || !edge_idx || !val))) return SCN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    SCN_HIP_TRY(hipMemsetAsync(x, 0, sizeof(float) * (size_t)n_slabs * n_edges * ns, st));
    if (n_entries) {
        hipLaunchKernelGGL(scatter_flows_kernel, dim3((unsigned)((n_entries + 255) / 256)), dim3(256), 0, st, ns,
                           n_edges, n_entries, sample_of, edge_idx, val, x);
        SCN_LAUNCH_CHECK();
    }
    return SCN_OK;
}

// Host-only: one batch of ragged flows, last nodes and targets into the caller's (pinned) staging words -- what a graph-replayed step
// copies to the device in ONE transfer.  Plain loops: the same assembly in NumPy was 85 us of a 155 us optimiser step on the
// reference's own configuration (a dozen array calls on ~1000 elements each).
int64_t scn_host_stage_batch(int32_t m, const int32_t* traj, int32_t n_total, const int64_t* ptr, const int32_t* edge, const float* val,
                             const int32_t* last_nodes, const float* y, int32_t d, double total, int32_t e_cap,
                             int32_t n_cap, int32_t* out) {
    if (m < 0 || n_total < 0 || d <= 0 || e_cap <= 0 || n_cap <= 0 || !(total > 0.0)) return SCN_ERR_BAD_SHAPE;
    if (!out || (m > 0 && (!traj || !ptr || !edge || !val || !last_nodes || !y))) return SCN_ERR_BAD_ARG;
    if (m > n_cap) return SCN_ERR_UNSUPPORTED;
    for (int32_t j = 0; j < m; ++j)
        if (traj[j] < 0 || traj[j] >= n_total) return SCN_ERR_BAD_ARG;          // an index outside the data set: nothing is read
    int64_t k = 0;
    for (int32_t j = 0; j < m; ++j) k += ptr[traj[j] + 1] - ptr[traj[j]];
    if (k > e_cap) return SCN_ERR_UNSUPPORTED;
    const size_t n_words = (size_t)3 * e_cap + n_cap + (size_t)n_cap * d;
    std::memset(out, 0, n_words * sizeof(int32_t));
    int32_t* sample = out;
    int32_t* eidx = out + e_cap;
    float* v = (float*)(out + 2 * (size_t)e_cap);
    int32_t* last = out + 3 * (size_t)e_cap;
    float* yo = (float*)(out + 3 * (size_t)e_cap + n_cap);
    int64_t o = 0;
    for (int32_t j = 0; j < m; ++j) {
        const int32_t n = traj[j];
        for (int64_t t = ptr[n]; t < ptr[n + 1]; ++t, ++o) {
            sample[o] = j;
            eidx[o] = edge[t];
            v[o] = val[t];
        }
        last[j] = last_nodes[n];
        for (int32_t c = 0; c < d; ++c) yo[(size_t)j * d + c] = (float)((double)y[(size_t)n * d + c] / total);
    }
    return k;
}

int scn_logits_sum_log_softmax(int32_t n_traj, int32_t max_deg, int32_t n_parts, const float* const* logits_parts, float* logits,
                               float* logp, void* stream) {
    if (n_traj <= 0 || max_deg <= 0 || n_parts <= 0 || n_parts > 4) return SCN_ERR_BAD_SHAPE;
    if (!logits_parts || !logits || !logp) return SCN_ERR_BAD_ARG;
    LogitParts L{};
    for (int k = 0; k < n_parts; ++k) {
        if (!logits_parts[k]) return SCN_ERR_BAD_ARG;
        L.p[k] = logits_parts[k];
    }
    hipLaunchKernelGGL(logits_sum_log_softmax_kernel, dim3(n_traj), dim3(64), 0, (hipStream_t)stream, max_deg, n_parts, L, logits, logp);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

int scn_masked_ce(int64_t n, const float* logp, const float* y, float scale, float* d_logp, double* loss, void* stream) {
    if (n <= 0) return SCN_ERR_BAD_SHAPE;
    if (!logp || !y || !d_logp || !loss) return SCN_ERR_BAD_ARG;
    hipLaunchKernelGGL(masked_ce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, logp, y, scale, d_logp, loss);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

int scn_masked_ce_begin(int64_t n, const float* logp, const float* y, float scale, float* d_logp, double* loss, int32_t overwrite,
                        float* zero_buf, int64_t zero_n, void* stream) {
    if (n <= 0 || zero_n < 0) return SCN_ERR_BAD_SHAPE;
    if (!logp || !y || !d_logp || !loss || (zero_n > 0 && !zero_buf)) return SCN_ERR_BAD_ARG;
    hipLaunchKernelGGL(masked_ce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, logp, y, scale, d_logp, loss, overwrite ? 1 : 0,
                       zero_buf, zero_n);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

int scn_adam_step(int64_t n, float* w, const float* g, float* m, float* v, float lr, float b1, float b2, float eps,
                  int32_t step_i, float weight_decay, float g_scale, void* stream) {
    if (n <= 0 || step_i < 0) return SCN_ERR_BAD_SHAPE;
    if (!w || !g || !m || !v) return SCN_ERR_BAD_ARG;
    float c1, c2;
    adam_corrections(b1, b2, step_i, c1, c2);            // (as the kernels that read the index from device memory: the same bits)
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, w, g, m,
                       v, lr, b1, b2, eps, c1, c2, 2.f * weight_decay, g_scale);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

int scn_adam_step_dev(int64_t n, float* w, const float* g, float* m, float* v, float lr, float b1, float b2, float eps,
                      int32_t* step_dev, float weight_decay, float g_scale, void* stream) {
    if (n <= 0) return SCN_ERR_BAD_SHAPE;
    if (n > SCN_ADAM_DEV_MAX) return SCN_ERR_UNSUPPORTED;
    if (!w || !g || !m || !v || !step_dev) return SCN_ERR_BAD_ARG;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(1), dim3(ADAM_DEV_THREADS), 0, (hipStream_t)stream, n, w, g, m, v, lr, b1, b2, eps,
                       step_dev, 2.f * weight_decay, g_scale);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

}  // extern "C"
