// Shift-convolution operator objects and the GENERIC kernels (any channel width, any number of groups).
// The LDS-blocked MFMA kernels for the scone/ebli hot configurations live in scn_blocked.hip and are
// dispatched from here when the shape qualifies.
#include <algorithm>
#include <cstring>
#include <new>

#include "scn_internal.h"

namespace scn {

static thread_local std::string g_hip_err;
void set_hip_error(hipError_t e, const char* where) {
    g_hip_err = std::string(where) + ": " + hipGetErrorString(e);
}

// implemented in scn_blocked.hip
int build_block_plan(scn_conv_s* c);
void free_block_plan(scn_conv_s* c);
bool blocked_forward_supported(const scn_conv_s* c, int ns, const int32_t* c_in, int c_out);
int blocked_forward(scn_conv_s* c, int n_slabs, int ns, const float* const* src, const int32_t* c_in,
                    const float* const* W, int c_out, int act, float* out, float* y_out, const WorkList* wl,
                    hipStream_t st, const float* partial = nullptr);
bool blocked_backward_supported(const scn_conv_s* c, int ns, const int32_t* c_dz, int c_aux, bool has_dx);
size_t blocked_backward_workspace(const scn_conv_s* c, int n_slabs, int ns, const int32_t* c_dz, int c_aux);
int blocked_backward(scn_conv_s* c, int n_slabs, int ns, const float* const* dz, const int32_t* c_dz,
                     const float* const* W, const float* aux, int c_aux, int act, float* dx,
                     float* const* dW, void* ws, size_t ws_bytes, const WorkList* wl, hipStream_t st,
                     const float* dx_partial = nullptr);
bool blocked_spmm_supported(const scn_conv_s* c, int k);
bool blocked_power_supported(const scn_conv_s* c, int ns, int ch);
size_t blocked_power_backward_workspace(const scn_conv_s* c, int n_slabs, int ns, int ch);
int blocked_power_forward(scn_conv_s* c, int n_slabs, const float* x0, const float* x, const float* const* W, int ch, int act,
                          float* out, hipStream_t st);
int blocked_power_backward(scn_conv_s* c, int n_slabs, const float* dz, const float* g1, const float* const* W,
                           const float* aux, int ch, int act, float* dx, float* const* dW, void* ws, hipStream_t st);
bool blocked_backward_first_supported(const scn_conv_s* c, int ns, int ch);
size_t blocked_backward_first_workspace(const scn_conv_s* c, int n_slabs, int ns, int ch);
int blocked_backward_first(scn_conv_s* c, int n_slabs, const float* dz, const float* const* W, const float* aux, int ch, int act,
                           const float* y, float* const* dW, float* const* dW_first, void* ws, const WorkList* wlp,
                           hipStream_t st);
int build_terms_plan(scn_conv_s* c, const uint8_t* term, const int32_t* lvl_row0, const uint8_t* merged, const int32_t* bins,
                     int group_rows);
int terms_forward(scn_conv_s* c, int n_slabs, const float* const* x, const float* const* W, int act, float* const* out,
                  hipStream_t st);
size_t terms_backward_workspace(const scn_conv_s* c, int n_slabs);
int terms_backward(scn_conv_s* c, int n_slabs, const float* const* dz, const float* const* W, const float* const* aux, int act,
                   float* const* dx, float* const* dW, const float* const* y, float* const* dW_first, void* ws, hipStream_t st);
bool blocked_dw_first_supported(const scn_conv_s* c, int ns, int cd);
size_t blocked_dw_first_workspace(const scn_conv_s* c, int n_slabs, int ns, int cd);
int blocked_dw_first(scn_conv_s* c, int n_slabs, const float* x, const float* y, const float* dz, int cd,
                     float* const* dW, void* ws, const WorkList* wl, hipStream_t st);
int blocked_clear_list(scn_conv_s* c, int ns, int ch, float* t, const WorkList* wl, hipStream_t st);
int blocked_spmm(scn_conv_s* c, int n_slabs, int k, const float* x, float* ya, float* yb, hipStream_t st);

// ------------------------------------------------------------------------------------------------
// generic kernels
// ------------------------------------------------------------------------------------------------
struct OpArrays {
    int32_t n_rows, n_groups, n_slots;
    const int32_t* rowptr[SCN_MAX_GROUPS];
    const int32_t* col[SCN_MAX_GROUPS];
    const float* val0[SCN_MAX_GROUPS];
    const float* val1[SCN_MAX_GROUPS];
    int32_t identity[SCN_MAX_GROUPS], n_vals[SCN_MAX_GROUPS], n_cols[SCN_MAX_GROUPS];
    int32_t slot_base[SCN_MAX_GROUPS], g_slots[SCN_MAX_GROUPS];
};

struct FwdArgs {
    OpArrays op;
    int32_t n_slabs, ns, c_out, act;
    int32_t c_in[SCN_MAX_GROUPS];
    const float* src[SCN_MAX_GROUPS];
    const float* W[SCN_MAX_SLOTS];
    float* out;
};

// gather every slot of every group for (row r, slab s) into LDS: Z[slot][ns*c_g]
template <typename ArgsT>
__device__ __forceinline__ void gather_row(const ArgsT& a, const float* const* srcs, const int32_t* cw,
                                           int r, int s, float* Z) {
    int zoff = 0;
    for (int g = 0; g < a.op.n_groups; ++g) {
        const int K = a.ns * cw[g];
        const float* X = srcs[g] + (size_t)s * a.op.n_cols[g] * K;
        const int j0 = a.op.rowptr[g][r], j1 = a.op.rowptr[g][r + 1];
        const int nv = a.op.n_vals[g], idn = a.op.identity[g];
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            float a0 = 0.f, a1 = 0.f;
            for (int j = j0; j < j1; ++j) {
                const float x = X[(size_t)a.op.col[g][j] * K + k];
                if (nv > 0) a0 = fmaf(a.op.val0[g][j], x, a0);
                if (nv > 1) a1 = fmaf(a.op.val1[g][j], x, a1);
            }
            int sl = 0;
            if (idn) Z[zoff + (sl++) * K + k] = X[(size_t)r * K + k];
            if (nv > 0) Z[zoff + (sl++) * K + k] = a0;
            if (nv > 1) Z[zoff + (sl++) * K + k] = a1;
        }
        zoff += a.op.g_slots[g] * K;
    }
}

__global__ __launch_bounds__(128) void conv_fwd_generic(FwdArgs a) {
    extern __shared__ float Z[];
    const int r = blockIdx.x, s = blockIdx.y;
    gather_row(a, a.src, a.c_in, r, s, Z);
    __syncthreads();
    const int KO = a.ns * a.c_out;
    for (int o = threadIdx.x; o < KO; o += blockDim.x) {
        const int n = o / a.c_out, co = o - n * a.c_out;
        float acc = 0.f;
        int zoff = 0;
        for (int g = 0; g < a.op.n_groups; ++g) {
            const int cg = a.c_in[g], K = a.ns * cg;
            for (int sl = 0; sl < a.op.g_slots[g]; ++sl) {
                const float* Wp = a.W[a.op.slot_base[g] + sl];
                const float* z = Z + zoff + sl * K + n * cg;
                for (int ci = 0; ci < cg; ++ci) acc = fmaf(z[ci], Wp[ci * a.c_out + co], acc);
            }
            zoff += a.op.g_slots[g] * K;
        }
        a.out[((size_t)(s * (size_t)a.op.n_rows + r) * a.ns) * a.c_out + o] = act_apply(a.act, acc);
    }
}

constexpr int BWD_THREADS = 256;
constexpr int BWD_MAXP = 48;   // dW pairs per thread: c_aux * sum(c_dz over slots) <= 256 * 48

struct BwdArgs {
    OpArrays op;
    int32_t n_slabs, ns, c_aux, act;
    int32_t c_dz[SCN_MAX_GROUPS];
    const float* dz[SCN_MAX_GROUPS];
    const float* W[SCN_MAX_SLOTS];
    const float* aux;
    float* dx;
    float* partial;          // [gridDim.x][total_pairs]
    int32_t total_cols;      // sum over slots of c_dz(group of slot)
    int32_t total_pairs;     // c_aux * total_cols
    int64_t n_items;         // n_rows * n_slabs
};

__global__ __launch_bounds__(BWD_THREADS) void conv_bwd_generic(BwdArgs a) {
    extern __shared__ float sm[];
    // LDS: G [sum_slots ns*c] | AUX [ns*c_aux]
    int gtot = 0;
    for (int g = 0; g < a.op.n_groups; ++g) gtot += a.op.g_slots[g] * a.ns * a.c_dz[g];
    float* G = sm;
    float* AUX = sm + gtot;

    // per-thread dW pair bookkeeping: pair p = ca * total_cols + col ; col -> (slot, c)
    float accW[BWD_MAXP];
    int pz[BWD_MAXP];    // LDS offset of G[slot][n=0][c]
    int pstride[BWD_MAXP];
    int pca[BWD_MAXP];
#pragma unroll
    for (int i = 0; i < BWD_MAXP; ++i) {
        accW[i] = 0.f;
        pz[i] = 0; pstride[i] = 0; pca[i] = 0;
        const int p = threadIdx.x + i * BWD_THREADS;
        if (p < a.total_pairs) {
            const int ca = p / a.total_cols;
            int colc = p - ca * a.total_cols;
            int zoff = 0;
            for (int g = 0; g < a.op.n_groups; ++g) {
                const int cg = a.c_dz[g], K = a.ns * cg;
                bool found = false;
                for (int sl = 0; sl < a.op.g_slots[g]; ++sl) {
                    if (!found && colc < cg) { pz[i] = zoff + sl * K + colc; pstride[i] = cg; found = true; }
                    if (!found) colc -= cg;
                }
                if (found) break;
                zoff += a.op.g_slots[g] * K;
            }
            pca[i] = ca;
        }
    }

    for (int64_t item = blockIdx.x; item < a.n_items; item += gridDim.x) {
        const int s = (int)(item / a.op.n_rows);
        const int r = (int)(item - (int64_t)s * a.op.n_rows);
        gather_row(a, a.dz, a.c_dz, r, s, G);
        const size_t abase = ((size_t)s * a.op.n_rows + r) * a.ns * a.c_aux;
        for (int k = threadIdx.x; k < a.ns * a.c_aux; k += blockDim.x) AUX[k] = a.aux[abase + k];
        __syncthreads();
        if (a.dx) {
            for (int o = threadIdx.x; o < a.ns * a.c_aux; o += blockDim.x) {
                const int n = o / a.c_aux, ca = o - n * a.c_aux;
                float acc = 0.f;
                int zoff = 0;
                for (int g = 0; g < a.op.n_groups; ++g) {
                    const int cg = a.c_dz[g], K = a.ns * cg;
                    for (int sl = 0; sl < a.op.g_slots[g]; ++sl) {
                        const float* Wp = a.W[a.op.slot_base[g] + sl] + (size_t)ca * cg;
                        const float* z = G + zoff + sl * K + n * cg;
                        for (int c = 0; c < cg; ++c) acc = fmaf(z[c], Wp[c], acc);
                    }
                    zoff += a.op.g_slots[g] * K;
                }
                a.dx[abase + o] = acc * act_grad_from_output(a.act, AUX[o]);
            }
        }
#pragma unroll
        for (int i = 0; i < BWD_MAXP; ++i) {
            if (threadIdx.x + i * BWD_THREADS < a.total_pairs) {
                float acc = accW[i];
                for (int n = 0; n < a.ns; ++n)
                    acc = fmaf(AUX[n * a.c_aux + pca[i]], G[pz[i] + n * pstride[i]], acc);
                accW[i] = acc;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < BWD_MAXP; ++i) {
        const int p = threadIdx.x + i * BWD_THREADS;
        if (p < a.total_pairs) a.partial[(size_t)blockIdx.x * a.total_pairs + p] = accW[i];
    }
}

struct ReduceArgs {
    const float* partial;
    int32_t n_partials, total_pairs, total_cols, n_slots;
    int32_t slot_c[SCN_MAX_SLOTS];   // c_dz of each slot
    float* dW[SCN_MAX_SLOTS];
};

// dW_slot[ca][c] += sum_b partial[b][ca*total_cols + col]   (fixed order -> bitwise reproducible)
__global__ void dw_reduce_kernel(ReduceArgs a) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.total_pairs) return;
    float acc = 0.f;
    for (int b = 0; b < a.n_partials; ++b) acc += a.partial[(size_t)b * a.total_pairs + p];
    const int ca = p / a.total_cols;
    int col = p - ca * a.total_cols;
    for (int sl = 0; sl < a.n_slots; ++sl) {
        if (col < a.slot_c[sl]) {
            if (a.dW[sl]) a.dW[sl][ca * a.slot_c[sl] + col] += acc;
            return;
        }
        col -= a.slot_c[sl];
    }
}

// generic dual SpMM: one block per (row, slab)
__global__ __launch_bounds__(128) void spmm_dual_generic(const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ col,
                                                         const float* __restrict__ v0,
                                                         const float* __restrict__ v1,
                                                         int n_rows, int n_cols, int k,
                                                         const float* __restrict__ x,
                                                         float* __restrict__ ya, float* __restrict__ yb) {
    const int r = blockIdx.x, s = blockIdx.y;
    const float* X = x + (size_t)s * n_cols * k;
    const int j0 = rowptr[r], j1 = rowptr[r + 1];
    for (int t = threadIdx.x; t < k; t += blockDim.x) {
        float a0 = 0.f, a1 = 0.f;
        for (int j = j0; j < j1; ++j) {
            const float xv = X[(size_t)col[j] * k + t];
            a0 = fmaf(v0[j], xv, a0);
            if (yb) a1 = fmaf(v1[j], xv, a1);
        }
        const size_t o = ((size_t)s * n_rows + r) * k + t;
        ya[o] = a0;
        if (yb) yb[o] = a1;
    }
}

static void fill_op(const scn_conv_s* c, OpArrays& op) {
    op.n_rows = c->n_rows; op.n_groups = c->n_groups; op.n_slots = c->n_slots;
    for (int g = 0; g < SCN_MAX_GROUPS; ++g) {
        const Group& G = c->g[g];
        op.rowptr[g] = G.d_rowptr; op.col[g] = G.d_col; op.val0[g] = G.d_val0; op.val1[g] = G.d_val1;
        op.identity[g] = G.identity; op.n_vals[g] = G.n_vals; op.n_cols[g] = G.n_cols;
        op.slot_base[g] = G.slot_base; op.g_slots[g] = G.n_slots;
    }
}

}  // namespace scn

using namespace scn;

extern "C" {

int scn_version(void) { return 100; }

const char* scn_error_string(int status) {
    switch (status) {
        case SCN_OK: return "ok";
        case SCN_ERR_BAD_ARG: return "bad argument";
        case SCN_ERR_BAD_SHAPE: return "bad shape";
        case SCN_ERR_HIP: return "HIP runtime error (see scn_last_hip_error)";
        case SCN_ERR_UNSUPPORTED: return "unsupported configuration";
        case SCN_ERR_NOMEM: return "out of memory";
        case SCN_ERR_WORKSPACE: return "workspace too small";
        case SCN_ERR_INTERNAL: return "internal self-check failed";
        default: return "unknown status";
    }
}

const char* scn_last_hip_error(void) { return g_hip_err.c_str(); }

int scn_conv_destroy(scn_conv_t c) {
    if (!c) return SCN_OK;
    for (int g = 0; g < SCN_MAX_GROUPS; ++g) {
        if (c->g[g].d_rowptr) (void)hipFree(c->g[g].d_rowptr);
        if (c->g[g].d_col) (void)hipFree(c->g[g].d_col);
        if (c->g[g].d_val0) (void)hipFree(c->g[g].d_val0);
        if (c->g[g].d_val1) (void)hipFree(c->g[g].d_val1);
    }
    free_block_plan(c);
    delete c;
    return SCN_OK;
}

int scn_conv_create(int32_t n_rows, int32_t n_groups, const scn_group_desc* groups, scn_conv_t* out) {
    return scn_conv_create_blocked(n_rows, n_groups, groups, nullptr, out);
}

int scn_conv_create_blocked(int32_t n_rows, int32_t n_groups, const scn_group_desc* groups, const uint8_t* block_start,
                            scn_conv_t* out) {
    if (!out || !groups || n_rows <= 0 || n_groups <= 0 || n_groups > SCN_MAX_GROUPS) return SCN_ERR_BAD_ARG;
    *out = nullptr;
    scn_conv_s* c = new (std::nothrow) scn_conv_s();
    if (!c) return SCN_ERR_NOMEM;
    c->n_rows = n_rows;
    c->n_groups = n_groups;
    if (block_start) c->block_start.assign(block_start, block_start + n_rows);
    int slot = 0;
    for (int g = 0; g < n_groups; ++g) {
        const scn_group_desc& d = groups[g];
        if (d.n_cols <= 0 || d.n_vals < 0 || d.n_vals > 2 || d.nnz < 0 || (d.identity && d.n_cols != n_rows) ||
            (d.n_vals == 0 && !d.identity) || !d.rowptr || (d.nnz > 0 && !d.col) ||
            (d.n_vals > 0 && d.nnz > 0 && !d.val0) || (d.n_vals > 1 && d.nnz > 0 && !d.val1)) {
            delete c;
            return SCN_ERR_BAD_ARG;
        }
        if (d.rowptr[0] != 0 || d.rowptr[n_rows] != d.nnz) { delete c; return SCN_ERR_BAD_SHAPE; }
        for (int r = 0; r < n_rows; ++r)
            if (d.rowptr[r + 1] < d.rowptr[r]) { delete c; return SCN_ERR_BAD_SHAPE; }
        for (int64_t j = 0; j < d.nnz; ++j)
            if (d.col[j] < 0 || d.col[j] >= d.n_cols) { delete c; return SCN_ERR_BAD_SHAPE; }
        Group& G = c->g[g];
        G.n_cols = d.n_cols; G.identity = d.identity ? 1 : 0; G.n_vals = d.n_vals; G.nnz = d.nnz;
        G.slot_base = slot;
        G.n_slots = G.identity + G.n_vals;
        if (slot + G.n_slots > SCN_MAX_SLOTS) { delete c; return SCN_ERR_UNSUPPORTED; }
        if (G.identity) { c->slot_group[slot] = g; c->slot_kind[slot++] = 0; }
        if (G.n_vals > 0) { c->slot_group[slot] = g; c->slot_kind[slot++] = 1; }
        if (G.n_vals > 1) { c->slot_group[slot] = g; c->slot_kind[slot++] = 2; }
        try {
            G.h_rowptr.assign(d.rowptr, d.rowptr + n_rows + 1);
            G.h_col.assign(d.col, d.col + d.nnz);
            if (G.n_vals > 0) G.h_val0.assign(d.val0, d.val0 + d.nnz);
            if (G.n_vals > 1) G.h_val1.assign(d.val1, d.val1 + d.nnz);
        } catch (...) { delete c; return SCN_ERR_NOMEM; }
    }
    c->n_slots = slot;
    for (int g = 0; g < n_groups; ++g) {
        Group& G = c->g[g];
        const size_t nz = (size_t)std::max<int64_t>(G.nnz, 1);
#define SCN_CREATE_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { set_hip_error(_e, #expr); scn_conv_destroy(c); return SCN_ERR_HIP; } } while (0)
        SCN_CREATE_TRY(hipMalloc(&G.d_rowptr, sizeof(int32_t) * (n_rows + 1)));
        SCN_CREATE_TRY(hipMemcpy(G.d_rowptr, G.h_rowptr.data(), sizeof(int32_t) * (n_rows + 1), hipMemcpyHostToDevice));
        SCN_CREATE_TRY(hipMalloc(&G.d_col, sizeof(int32_t) * nz));
        if (G.nnz) SCN_CREATE_TRY(hipMemcpy(G.d_col, G.h_col.data(), sizeof(int32_t) * G.nnz, hipMemcpyHostToDevice));
        if (G.n_vals > 0) {
            SCN_CREATE_TRY(hipMalloc(&G.d_val0, sizeof(float) * nz));
            if (G.nnz) SCN_CREATE_TRY(hipMemcpy(G.d_val0, G.h_val0.data(), sizeof(float) * G.nnz, hipMemcpyHostToDevice));
        }
        if (G.n_vals > 1) {
            SCN_CREATE_TRY(hipMalloc(&G.d_val1, sizeof(float) * nz));
            if (G.nnz) SCN_CREATE_TRY(hipMemcpy(G.d_val1, G.h_val1.data(), sizeof(float) * G.nnz, hipMemcpyHostToDevice));
        }
    }
    int st = build_block_plan(c);
    if (st == SCN_OK) st = small_prepare(c);
    if (st != SCN_OK) { scn_conv_destroy(c); return st; }
    *out = c;
    return SCN_OK;
}

int scn_terms_create(int32_t n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const uint8_t* term,
                     const int32_t* level_row0, const uint8_t* merged, const int32_t* bins, int32_t rows_per_wave,
                     scn_conv_t* out) {
    if (!out || !rowptr || !level_row0 || !merged || !bins || n_rows <= 0) return SCN_ERR_BAD_ARG;
    if (rows_per_wave != 4 && rows_per_wave != 8) return SCN_ERR_BAD_ARG;
    {
        int64_t seen[3] = {0, 0, 0};
        for (int i = 0; i < n_rows; ++i) {
            if (merged[i] > 2) return SCN_ERR_BAD_SHAPE;
            ++seen[merged[i]];
        }
        for (int l = 0; l < 3; ++l)
            if (seen[l] != level_row0[l + 1] - level_row0[l]) return SCN_ERR_BAD_SHAPE;   // every row exactly once
    }
    *out = nullptr;
    const int64_t nnz = rowptr[n_rows];
    if (rowptr[0] != 0 || nnz < 0 || (nnz > 0 && (!col || !val || !term))) return SCN_ERR_BAD_ARG;
    if (level_row0[0] != 0 || level_row0[1] < 0 || level_row0[2] < level_row0[1] || level_row0[3] != n_rows ||
        level_row0[2] > n_rows)
        return SCN_ERR_BAD_SHAPE;
    for (int r = 0; r < n_rows; ++r)
        if (rowptr[r + 1] < rowptr[r]) return SCN_ERR_BAD_SHAPE;
    for (int64_t j = 0; j < nnz; ++j) {
        if (col[j] < 0 || col[j] >= n_rows || term[j] > 2) return SCN_ERR_BAD_SHAPE;
        if (term[j] != (col[j] >= level_row0[1]) + (col[j] >= level_row0[2])) return SCN_ERR_BAD_SHAPE;   // term = level of the column
    }
    scn_conv_s* c = new (std::nothrow) scn_conv_s();
    if (!c) return SCN_ERR_NOMEM;
    c->n_rows = n_rows;
    c->n_groups = 1;
    Group& G = c->g[0];
    G.n_cols = n_rows; G.identity = 0; G.n_vals = 1; G.nnz = nnz; G.slot_base = 0; G.n_slots = 1;
    c->n_slots = 1;
    c->slot_group[0] = 0; c->slot_kind[0] = 1;
    try {
        G.h_rowptr.assign(rowptr, rowptr + n_rows + 1);
        G.h_col.assign(col, col + nnz);
        G.h_val0.assign(val, val + nnz);
    } catch (...) { delete c; return SCN_ERR_NOMEM; }
    const int st = build_terms_plan(c, term, level_row0, merged, bins, rows_per_wave);
    if (st != SCN_OK) { scn_conv_destroy(c); return st; }
    *out = c;
    return SCN_OK;
}

int scn_terms_forward(scn_conv_t c, int32_t n_slabs, int32_t ns, const float* const* x, const float* const* W, int32_t channels,
                      int32_t act, float* const* out, void* stream) {
    if (!c || !x || !W || !out) return SCN_ERR_BAD_ARG;
    if (!c->terms.built) return SCN_ERR_UNSUPPORTED;
    if (n_slabs <= 0 || act < 0 || act > 3) return SCN_ERR_BAD_SHAPE;
    if (ns != BK_NS || channels != 32 || n_slabs > 65535) return SCN_ERR_UNSUPPORTED;
    for (int l = 0; l < 3; ++l)
        if (out[l] && c->terms.lvl_row0[l + 1] == c->terms.lvl_row0[l]) return SCN_ERR_BAD_SHAPE;
    return terms_forward(c, n_slabs, x, W, act, out, (hipStream_t)stream);
}

size_t scn_terms_backward_workspace(scn_conv_t c, int32_t n_slabs, int32_t ns, int32_t channels) {
    if (!c || !c->terms.built || c->terms.group_rows != 8 || n_slabs <= 0 || ns != BK_NS || channels != 32) return 0;
    return terms_backward_workspace(c, n_slabs);
}

int scn_terms_backward(scn_conv_t c, int32_t n_slabs, int32_t ns, const float* const* dz, const float* const* W,
                       const float* const* aux, int32_t channels, int32_t act, float* const* dx, float* const* dW,
                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!c || !dz || !W || !aux || !dx || !dW || !workspace) return SCN_ERR_BAD_ARG;
    if (!c->terms.built) return SCN_ERR_UNSUPPORTED;
    if (n_slabs <= 0 || act < 0 || act > 3) return SCN_ERR_BAD_SHAPE;
    if (ns != BK_NS || channels != 32 || n_slabs > 65535) return SCN_ERR_UNSUPPORTED;
    if (workspace_bytes < scn_terms_backward_workspace(c, n_slabs, ns, channels)) return SCN_ERR_WORKSPACE;
    return terms_backward(c, n_slabs, dz, W, aux, act, dx, dW, nullptr, nullptr, workspace, (hipStream_t)stream);
}

int scn_terms_backward_fused_first(scn_conv_t c, int32_t n_slabs, int32_t ns, const float* const* dz, const float* const* W,
                                   const float* const* aux, int32_t channels, int32_t act, const float* const* y, float* const* dW,
                                   float* const* dW_first, void* workspace, size_t workspace_bytes, void* stream) {
    if (!c || !dz || !W || !aux || !y || !dW || !dW_first || !workspace) return SCN_ERR_BAD_ARG;
    if (!c->terms.built) return SCN_ERR_UNSUPPORTED;
    if (n_slabs <= 0 || act < 0 || act > 3) return SCN_ERR_BAD_SHAPE;
    if (ns != BK_NS || channels != 32 || n_slabs > 65535) return SCN_ERR_UNSUPPORTED;
    if (workspace_bytes < scn_terms_backward_workspace(c, n_slabs, ns, channels)) return SCN_ERR_WORKSPACE;
    for (int l = 0; l < 3; ++l)
        if ((y[l] == nullptr) != (dW_first[l] == nullptr)) return SCN_ERR_BAD_ARG;
    return terms_backward(c, n_slabs, dz, W, aux, act, nullptr, dW, y, dW_first, workspace, (hipStream_t)stream);
}

int scn_conv_n_slots(scn_conv_t c) { return c ? c->n_slots : SCN_ERR_BAD_ARG; }

int scn_conv_plan_info(scn_conv_t c, int32_t* n_blocks, float* mean_sources_per_row) {
    if (!c) return SCN_ERR_BAD_ARG;
    const bool has = c->plan.built || c->terms.built;
    if (n_blocks) *n_blocks = has ? c->plan.dev.n_blocks : 0;
    if (mean_sources_per_row) *mean_sources_per_row = has ? (float)c->plan.mean_src_per_row : 0.f;
    return SCN_OK;
}

int scn_conv_plan_blocks(scn_conv_t c, int32_t* row0_out) {
    if (!c || !row0_out) return SCN_ERR_BAD_ARG;
    if (!c->plan.built) return SCN_ERR_UNSUPPORTED;
    std::memcpy(row0_out, c->plan.h_row0.data(), c->plan.h_row0.size() * sizeof(int32_t));
    return SCN_OK;
}

static bool valid_list(const scn_work_list* wl) {
    return !wl || (wl->n_work >= 0 && wl->block && wl->ptr && (wl->slab || wl->n_work == 0));
}
static WorkList to_list(const scn_work_list* wl) { return WorkList{wl->n_work, wl->block, wl->ptr, wl->slab}; }

int scn_conv_forward(scn_conv_t c, int32_t n_slabs, int32_t ns, const float* const* src, const int32_t* c_in,
                     const float* const* W, int32_t c_out, int32_t act, float* out, void* stream) {
    return scn_conv_forward_list(c, n_slabs, ns, src, c_in, W, c_out, act, out, nullptr, stream);
}

int scn_conv_forward_accumulate(scn_conv_t c, int32_t n_slabs, int32_t ns, const float* const* src, const int32_t* c_in,
                                const float* const* W, int32_t c_out, int32_t act, const float* partial, float* out, void* stream) {
    if (!c || !src || !c_in || !W || !out || !partial) return SCN_ERR_BAD_ARG;
    if (n_slabs <= 0 || ns <= 0 || c_out <= 0 || act < 0 || act > 3) return SCN_ERR_BAD_SHAPE;
    if (n_slabs > 65535) return SCN_ERR_UNSUPPORTED;
    if (!src[0] || c_in[0] <= 0) return SCN_ERR_BAD_ARG;
    for (int s = 0; s < c->n_slots; ++s)
        if (!W[s]) return SCN_ERR_BAD_ARG;
    if (c->n_groups != 1 || c_in[0] != 32 || c_out != 32 || !blocked_forward_supported(c, ns, c_in, c_out)) return SCN_ERR_UNSUPPORTED;
    return blocked_forward(c, n_slabs, ns, src, c_in, W, c_out, act, out, nullptr, nullptr, (hipStream_t)stream, partial);
}

int scn_conv_forward_list(scn_conv_t c, int32_t n_slabs, int32_t ns, const float* const* src, const int32_t* c_in,
                          const float* const* W, int32_t c_out, int32_t act, float* out, const scn_work_list* wl,
                          void* stream) {
    if (!c || !src || !c_in || !W || !out || !valid_list(wl)) return SCN_ERR_BAD_ARG;
    if (n_slabs <= 0 || ns <= 0 || c_out <= 0 || act < 0 || act > 3) return SCN_ERR_BAD_SHAPE;
    if (n_slabs > 65535) return SCN_ERR_UNSUPPORTED;
    for (int g = 0; g < c->n_groups; ++g)
        if (!src[g] || c_in[g] <= 0) return SCN_ERR_BAD_ARG;
    for (int s = 0; s < c->n_slots; ++s)
        if (!W[s]) return SCN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    WorkList wlist{0, nullptr, nullptr, nullptr};
    if (wl) wlist = to_list(wl);
    if (blocked_forward_supported(c, ns, c_in, c_out))
        return blocked_forward(c, n_slabs, ns, src, c_in, W, c_out, act, out, nullptr, wl ? &wlist : nullptr, st);
    if (wl) return SCN_ERR_UNSUPPORTED;            // work lists exist for the LDS-blocked kernels only
    FwdArgs a;
    std::memset(&a, 0, sizeof(a));
    fill_op(c, a.op);
    a.n_slabs = n_slabs; a.ns = ns; a.c_out = c_out; a.act = act; a.out = out;
    size_t lds = 0;
    for (int g = 0; g < c->n_groups; ++g) {
        a.c_in[g] = c_in[g]; a.src[g] = src[g];
        lds += (size_t)c->g[g].n_slots * ns * c_in[g] * sizeof(float);
    }
    for (int s = 0; s < c->n_slots; ++s) a.W[s] = W[s];
    if (lds > 64 * 1024) return SCN_ERR_UNSUPPORTED;
    dim3 grid(c->n_rows, n_slabs);
    hipLaunchKernelGGL(conv_fwd_generic, grid, dim3(128), lds, st, a);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

static int generic_bwd_blocks(const scn_conv_s* c, int n_slabs) {
    int64_t items = (int64_t)c->n_rows * n_slabs;
    return (int)std::min<int64_t>(items, 2048);
}

size_t scn_conv_backward_workspace(scn_conv_t c, int32_t n_slabs, int32_t ns, const int32_t* c_dz, int32_t c_aux) {
    if (!c || !c_dz || n_slabs <= 0 || ns <= 0 || c_aux <= 0) return 0;
    size_t cols = 0;
    for (int s = 0; s < c->n_slots; ++s) cols += c_dz[c->slot_group[s]];
    size_t generic = (size_t)generic_bwd_blocks(c, n_slabs) * cols * c_aux * sizeof(float);
    size_t blocked = blocked_backward_workspace(c, n_slabs, ns, c_dz, c_aux);
    return std::max(generic, blocked) + 256;
}

size_t scn_conv_dw_first_workspace(scn_conv_t c, int32_t n_slabs, int32_t ns, int32_t c_dz) {
    if (!c || n_slabs <= 0) return 0;
    return blocked_dw_first_workspace(c, n_slabs, ns, c_dz);
}

int scn_conv_dw_first(scn_conv_t c, int32_t n_slabs, int32_t ns, const float* x, const float* y, const float* dz,
                      int32_t c_dz, float* const* dW, void* workspace, size_t workspace_bytes, const scn_work_list* wl,
                      void* stream) {
    if (!c || (!x && !y) || !dz || !dW || !dW[0] || !dW[1] || !dW[2] || !workspace || !valid_list(wl) || (wl && !y))
        return SCN_ERR_BAD_ARG;
    if (n_slabs <= 0) return SCN_ERR_BAD_SHAPE;
    if (!blocked_dw_first_supported(c, ns, c_dz)) return SCN_ERR_UNSUPPORTED;
    if (workspace_bytes < blocked_dw_first_workspace(c, n_slabs, ns, c_dz)) return SCN_ERR_WORKSPACE;
    WorkList wlist{0, nullptr, nullptr, nullptr};
    if (wl) wlist = to_list(wl);
    return blocked_dw_first(c, n_slabs, x, y, dz, c_dz, dW, workspace, wl ? &wlist : nullptr, (hipStream_t)stream);
}

size_t scn_conv_backward_fused_first_workspace(scn_conv_t c, int32_t n_slabs, int32_t ns, int32_t channels) {
    if (!c || n_slabs <= 0) return 0;
    return blocked_backward_first_workspace(c, n_slabs, ns, channels);
}

int scn_conv_backward_fused_first(scn_conv_t c, int32_t n_slabs, int32_t ns, const float* dz, const float* const* W,
                                  const float* aux, int32_t channels, int32_t act, const float* y, float* const* dW,
                                  float* const* dW_first, void* workspace, size_t workspace_bytes, const scn_work_list* wl,
                                  void* stream) {
    if (!c || !dz || !W || !W[0] || !W[1] || !W[2] || !aux || !y || !dW || !dW[0] || !dW[1] || !dW[2] || !dW_first ||
        !dW_first[0] || !dW_first[1] || !dW_first[2] || !workspace || !valid_list(wl))
        return SCN_ERR_BAD_ARG;
    if (n_slabs <= 0 || act < 0 || act > 3) return SCN_ERR_BAD_SHAPE;
    if (n_slabs > 65535 || !blocked_backward_first_supported(c, ns, channels)) return SCN_ERR_UNSUPPORTED;
    if (workspace_bytes < blocked_backward_first_workspace(c, n_slabs, ns, channels)) return SCN_ERR_WORKSPACE;
    WorkList wlist{0, nullptr, nullptr, nullptr};
    if (wl) wlist = to_list(wl);
    return blocked_backward_first(c, n_slabs, dz, W, aux, channels, act, y, dW, dW_first, workspace, wl ? &wlist : nullptr,
                                  (hipStream_t)stream);
}

int scn_conv_forward_first(scn_conv_t c, int32_t n_slabs, int32_t ns, const float* x, const float* const* W, int32_t c_out,
                           int32_t act, float* out, float* y_out, const scn_work_list* wl, void* stream) {
    if (!c || !x || !W || !W[0] || !W[1] || !W[2] || !out || !y_out || !valid_list(wl)) return SCN_ERR_BAD_ARG;
    if (n_slabs <= 0 || act < 0 || act > 3) return SCN_ERR_BAD_SHAPE;
    const int32_t c_in = 1;
    if (c->n_groups != 1 || c->n_slots != 3 || !blocked_forward_supported(c, ns, &c_in, c_out)) return SCN_ERR_UNSUPPORTED;
    WorkList wlist{0, nullptr, nullptr, nullptr};
    if (wl) wlist = to_list(wl);
    return blocked_forward(c, n_slabs, ns, &x, &c_in, W, c_out, act, out, y_out, wl ? &wlist : nullptr, (hipStream_t)stream);
}

int scn_clear_list(scn_conv_t c, int32_t ns, int32_t channels, float* tensor, const scn_work_list* wl, void* stream) {
    if (!c || !tensor || !wl || !valid_list(wl) || channels <= 0) return SCN_ERR_BAD_ARG;
    WorkList wlist = to_list(wl);
    return blocked_clear_list(c, ns, channels, tensor, &wlist, (hipStream_t)stream);
}

int scn_conv_forward_power(scn_conv_t c, int32_t n_slabs, int32_t ns, const float* x0, const float* x,
                           const float* const* W, int32_t channels, int32_t act, float* out, void* stream) {
    if (!c || !x0 || !x || !W || !W[0] || !W[1] || !W[2] || !out) return SCN_ERR_BAD_ARG;
    if (n_slabs <= 0 || act < 0 || act > 3) return SCN_ERR_BAD_SHAPE;
    if (!blocked_power_supported(c, ns, channels)) return SCN_ERR_UNSUPPORTED;
    return blocked_power_forward(c, n_slabs, x0, x, W, channels, act, out, (hipStream_t)stream);
}

size_t scn_conv_backward_power_workspace(scn_conv_t c, int32_t n_slabs, int32_t ns, int32_t channels) {
    if (!c || n_slabs <= 0) return 0;
    const size_t b = blocked_power_backward_workspace(c, n_slabs, ns, channels);
    return b ? b + 256 : 0;
}

int scn_conv_backward_power(scn_conv_t c, int32_t n_slabs, int32_t ns, const float* dz, const float* g1,
                            const float* const* W, const float* aux, int32_t channels, int32_t act, float* dx,
                            float* const* dW, void* workspace, size_t workspace_bytes, void* stream) {
    if (!c || !dz || !g1 || !W || !W[0] || !W[1] || !W[2] || !aux || !dW || !dW[0] || !dW[1] || !dW[2] || !workspace)
        return SCN_ERR_BAD_ARG;
    if (n_slabs <= 0 || act < 0 || act > 3) return SCN_ERR_BAD_SHAPE;
    if (!blocked_power_supported(c, ns, channels)) return SCN_ERR_UNSUPPORTED;
    if (workspace_bytes < scn_conv_backward_power_workspace(c, n_slabs, ns, channels)) return SCN_ERR_WORKSPACE;
    return blocked_power_backward(c, n_slabs, dz, g1, W, aux, channels, act, dx, dW, workspace, (hipStream_t)stream);
}

int scn_conv_backward(scn_conv_t c, int32_t n_slabs, int32_t ns, const float* const* dz, const int32_t* c_dz,
                      const float* const* W, const float* aux, int32_t c_aux, int32_t act, float* dx,
                      float* const* dW, void* workspace, size_t workspace_bytes, void* stream) {
    return scn_conv_backward_list(c, n_slabs, ns, dz, c_dz, W, aux, c_aux, act, dx, dW, workspace, workspace_bytes, nullptr,
                                  stream);
}

int scn_conv_backward_accumulate(scn_conv_t c, int32_t n_slabs, int32_t ns, const float* const* dz, const int32_t* c_dz,
                                 const float* const* W, const float* aux, int32_t c_aux, int32_t act, const float* dx_partial,
                                 float* dx, float* const* dW, void* workspace, size_t workspace_bytes, void* stream) {
    if (!c || !dz || !c_dz || !W || !aux || !dW || !workspace || !dx || !dx_partial) return SCN_ERR_BAD_ARG;
    if (n_slabs <= 0 || ns <= 0 || c_aux <= 0 || act < 0 || act > 3) return SCN_ERR_BAD_SHAPE;
    if (!dz[0] || c_dz[0] <= 0) return SCN_ERR_BAD_ARG;
    if (c->n_groups != 1 || c_dz[0] != 32 || c_aux != 32 || !blocked_backward_supported(c, ns, c_dz, c_aux, true)) return SCN_ERR_UNSUPPORTED;
    if (workspace_bytes < scn_conv_backward_workspace(c, n_slabs, ns, c_dz, c_aux)) return SCN_ERR_WORKSPACE;
    return blocked_backward(c, n_slabs, ns, dz, c_dz, W, aux, c_aux, act, dx, dW, workspace, workspace_bytes, nullptr,
                            (hipStream_t)stream, dx_partial);
}

int scn_conv_backward_list(scn_conv_t c, int32_t n_slabs, int32_t ns, const float* const* dz, const int32_t* c_dz,
                           const float* const* W, const float* aux, int32_t c_aux, int32_t act, float* dx,
                           float* const* dW, void* workspace, size_t workspace_bytes, const scn_work_list* wl,
                           void* stream) {
    if (!c || !dz || !c_dz || !W || !aux || !dW || !workspace || !valid_list(wl)) return SCN_ERR_BAD_ARG;
    if (n_slabs <= 0 || ns <= 0 || c_aux <= 0 || act < 0 || act > 3) return SCN_ERR_BAD_SHAPE;
    for (int g = 0; g < c->n_groups; ++g)
        if (!dz[g] || c_dz[g] <= 0) return SCN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (workspace_bytes < scn_conv_backward_workspace(c, n_slabs, ns, c_dz, c_aux)) return SCN_ERR_WORKSPACE;
    WorkList wlist{0, nullptr, nullptr, nullptr};
    if (wl) wlist = to_list(wl);
    if (blocked_backward_supported(c, ns, c_dz, c_aux, dx != nullptr))
        return blocked_backward(c, n_slabs, ns, dz, c_dz, W, aux, c_aux, act, dx, dW, workspace, workspace_bytes,
                                wl ? &wlist : nullptr, st);
    if (wl) return SCN_ERR_UNSUPPORTED;

    BwdArgs a;
    std::memset(&a, 0, sizeof(a));
    fill_op(c, a.op);
    a.n_slabs = n_slabs; a.ns = ns; a.c_aux = c_aux; a.act = act; a.aux = aux; a.dx = dx;
    a.partial = (float*)workspace;
    size_t lds = (size_t)ns * c_aux * sizeof(float);
    for (int g = 0; g < c->n_groups; ++g) {
        a.c_dz[g] = c_dz[g]; a.dz[g] = dz[g];
        lds += (size_t)c->g[g].n_slots * ns * c_dz[g] * sizeof(float);
    }
    int cols = 0;
    for (int s = 0; s < c->n_slots; ++s) { a.W[s] = W[s]; cols += c_dz[c->slot_group[s]]; }
    a.total_cols = cols;
    a.total_pairs = cols * c_aux;
    a.n_items = (int64_t)c->n_rows * n_slabs;
    if (a.total_pairs > BWD_THREADS * BWD_MAXP || lds > 64 * 1024) return SCN_ERR_UNSUPPORTED;
    for (int s = 0; s < c->n_slots; ++s)
        if (dx && !W[s]) return SCN_ERR_BAD_ARG;
    const int nb = generic_bwd_blocks(c, n_slabs);
    hipLaunchKernelGGL(conv_bwd_generic, dim3(nb), dim3(BWD_THREADS), lds, st, a);
    SCN_LAUNCH_CHECK();
    ReduceArgs ra;
    std::memset(&ra, 0, sizeof(ra));
    ra.partial = a.partial; ra.n_partials = nb; ra.total_pairs = a.total_pairs; ra.total_cols = cols;
    ra.n_slots = c->n_slots;
    for (int s = 0; s < c->n_slots; ++s) { ra.slot_c[s] = c_dz[c->slot_group[s]]; ra.dW[s] = dW[s]; }
    hipLaunchKernelGGL(dw_reduce_kernel, dim3((a.total_pairs + 255) / 256), dim3(256), 0, st, ra);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

int scn_spmm_dual(scn_conv_t c, int32_t n_slabs, int32_t k, const float* x, float* ya, float* yb, void* stream) {
    if (!c || !x || !ya) return SCN_ERR_BAD_ARG;
    if (n_slabs <= 0 || k <= 0 || n_slabs > 65535) return SCN_ERR_BAD_SHAPE;
    const Group& G = c->g[0];
    if (G.n_vals < 1 || (yb && G.n_vals < 2)) return SCN_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (blocked_spmm_supported(c, k)) return blocked_spmm(c, n_slabs, k, x, ya, yb, st);
    hipLaunchKernelGGL(spmm_dual_generic, dim3(c->n_rows, n_slabs), dim3(128), 0, st,
                       G.d_rowptr, G.d_col, G.d_val0, G.d_val1, c->n_rows, G.n_cols, k, x, ya, yb);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

}  // extern "C"
