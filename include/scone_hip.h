/*
 * scone_hip.h -- C-ABI of libscone_hip.so: the MI355X (gfx950) implementation of SCoNe's
 * Hodge-Laplacian convolution hot path.
 *
 * The reference (nglaze00/SCoNe_GCN) has no FFI: the path is plain Python/JAX.  Each entry point below
 * names the reference code it replaces (TE = trajectory_analysis/trajectory_experiments.py,
 * STM = trajectory_analysis/scone_trajectory_model.py).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every function returns an int status: 0 = SCN_OK, negative = error (scn_error_string()).
 *     No C++ exception crosses this boundary.
 *   - "device" pointers are HIP device memory owned by the CALLER (PyTorch-ROCm tensors in the host
 *     mirror); the library never allocates per call.  Handles own only the device copies of the index /
 *     value arrays made at create time and are immutable afterwards (shareable across streams).
 *   - all launches go to the caller's stream (hipStream_t passed as void*; NULL = default stream) and
 *     never synchronise.
 *
 * Activation layout ("flow slabs"): fp32 [n_slabs][n_rows][ns][c] row-major.  A slab holds ns
 * trajectories; row r of slab s is the contiguous piece of ns*c floats at ((s*n_rows + r)*ns)*c.
 * The reference's batched (N, E, C) tensor (vmap axis 0, STM:256) maps to slab = n / ns, sample = n % ns.
 */
#ifndef SCONE_HIP_H
#define SCONE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCN_OK               0
#define SCN_ERR_BAD_ARG     -1
#define SCN_ERR_BAD_SHAPE   -2
#define SCN_ERR_HIP         -3
#define SCN_ERR_UNSUPPORTED -4
#define SCN_ERR_NOMEM       -5
#define SCN_ERR_WORKSPACE   -6
#define SCN_ERR_INTERNAL    -7   /* a self-check of the library failed (scn_plan_gather_stats: the block layout lost or moved an entry) */

/* activation codes (TE:124-134) */
#define SCN_ACT_NONE        0
#define SCN_ACT_TANH        1   /* scone   TE:149 */
#define SCN_ACT_RELU        2   /* bunch   TE:195 */
#define SCN_ACT_LEAKY_RELU  3   /* ebli    TE:167, slope 0.01, x >= 0 */

#define SCN_MAX_GROUPS 3
#define SCN_MAX_SLOTS  4

int         scn_version(void);
const char* scn_error_string(int status);
/* last HIP error text captured by a failing call on this thread ("" if none) */
const char* scn_last_hip_error(void);

/* ---------------------------------------------------------------------------------------------------
 * Shift-convolution operator: everything that feeds ONE output level of one layer.
 *
 *   out[s,r,n,:] = act( sum_over_slots  ( sum_j val_slot[r,j] * src_g[s, col[r,j], n, :] ) @ W_slot )
 *
 * A group g is one source tensor + one CSR pattern (n_rows x n_cols) carrying 1 or 2 value arrays and,
 * optionally, an identity shift.  Weight slots are numbered group by group: [identity][val0][val1].
 *   scone / ebli layer (TE:145-147 / 163-165): 1 group, identity + {S_lower, S_upper} on their shared
 *       pattern -> slots (W[3i], W[3i+1], W[3i+2]).
 *   bunch node / edge / face level (TE:184-192): 2 / 3 / 2 groups with 1 value array each.
 * Host arrays are copied at create time.
 * --------------------------------------------------------------------------------------------------- */
typedef struct scn_conv_s* scn_conv_t;

typedef struct {
    int32_t        n_cols;    /* rows of this group's source tensor */
    int32_t        identity;  /* 1: also apply the identity shift (needs n_cols == n_rows) */
    int32_t        n_vals;    /* 0, 1 or 2 value arrays on the pattern (0 only with identity) */
    int32_t        reserved;
    int64_t        nnz;
    const int32_t* rowptr;    /* host, n_rows + 1 */
    const int32_t* col;       /* host, nnz, ascending within a row */
    const float*   val0;      /* host, nnz (or NULL when n_vals == 0) */
    const float*   val1;      /* host, nnz (or NULL when n_vals < 2) */
} scn_group_desc;

int scn_conv_create(int32_t n_rows, int32_t n_groups, const scn_group_desc* groups, scn_conv_t* out);
/* Same, with a layout hint: block_start[r] != 0 (host, n_rows bytes, or NULL) forces a block of the LDS-blocked plan to
 * begin at row r (the plan may still cut more often).  Produced by scn_plan_refine_order for the row order it returns. */
int scn_conv_create_blocked(int32_t n_rows, int32_t n_groups, const scn_group_desc* groups,
                            const uint8_t* block_start, scn_conv_t* out);
int scn_conv_destroy(scn_conv_t conv);
int scn_conv_n_slots(scn_conv_t conv);
/* rows per LDS-staged block and mean staged source rows per block of the blocked plan (0 if none) */
int scn_conv_plan_info(scn_conv_t conv, int32_t* n_blocks, float* mean_sources_per_row);

/* Forward of one output level (replaces the body of the layer loops TE:144-149, 162-167, 181-195).
 *   src[g]  device [n_slabs][n_cols_g][ns][c_in[g]]
 *   W[slot] device [c_in(group of slot)][c_out] row-major
 *   out     device [n_slabs][n_rows][ns][c_out]                                                  */
int scn_conv_forward(scn_conv_t conv, int32_t n_slabs, int32_t ns,
                     const float* const* src, const int32_t* c_in,
                     const float* const* W, int32_t c_out, int32_t act,
                     float* out, void* stream);

/* The same with a tensor of partial pre-activations added before the activation:
 *   out[s,r,n,:] = act( partial[s,r,n,:] + sum_over_slots (...) @ W_slot )      partial: device [n_slabs][n_rows][ns][c_out]
 * `partial` may be `out` itself (in place).  This is how a layer wider than 32 channels adds up its (input block, output block)
 * pairs of 32 channels (-hidden_layers takes any width, TE:103-110) without a summation pass of its own: the first pair of an output
 * block runs scn_conv_forward with SCN_ACT_NONE, the last one this entry point with the layer's activation.  Served for one group,
 * c_in = c_out = 32, ns = 4 on an LDS-blocked plan; SCN_ERR_UNSUPPORTED otherwise. */
int scn_conv_forward_accumulate(scn_conv_t conv, int32_t n_slabs, int32_t ns,
                                const float* const* src, const int32_t* c_in,
                                const float* const* W, int32_t c_out, int32_t act,
                                const float* partial, float* out, void* stream);

/* Backward of one INPUT level (what jax.grad generates for the same loops, STM:307).  `conv_t` is the
 * operator whose slots are the TRANSPOSED shifts feeding this input level (for scone's symmetric
 * L_lower / L_upper it is the forward object itself).
 *   dz[g]    device [n_slabs][n_cols_g][ns][c_dz[g]]   grad w.r.t. the pre-activation of output level g
 *   W[slot]  device [c_aux][c_dz(group of slot)]        the FORWARD weights (used transposed)
 *   aux      device [n_slabs][n_rows][ns][c_aux]        this level's forward input (= previous layer's output)
 *   dx       device [n_slabs][n_rows][ns][c_aux] or NULL:
 *               dx = ( sum_slots G_slot @ W_slot^T ) * act'(aux)    with G_slot the gathered dz
 *   dW[slot] device [c_aux][c_dz]   dW_slot += sum_{s,r,n} aux[s,r,n,:]^T G_slot[s,r,n,:]   (deterministic order)
 *   workspace: device scratch of at least scn_conv_backward_workspace() bytes                       */
size_t scn_conv_backward_workspace(scn_conv_t conv_t, int32_t n_slabs, int32_t ns,
                                   const int32_t* c_dz, int32_t c_aux);
int scn_conv_backward(scn_conv_t conv_t, int32_t n_slabs, int32_t ns,
                      const float* const* dz, const int32_t* c_dz,
                      const float* const* W, const float* aux, int32_t c_aux, int32_t act,
                      float* dx, float* const* dW,
                      void* workspace, size_t workspace_bytes, void* stream);

/* The same with a partial input gradient added:  dx = dx_partial + ( sum_slots G_slot @ W_slot^T ) * act'(aux)
 * (dx_partial: device [n_slabs][n_rows][ns][c_aux]; may be `dx` itself).  The backward counterpart of scn_conv_forward_accumulate: the
 * input gradient of an input block of a layer wider than 32 channels is the sum over the layer's output blocks (act' of the layer
 * below is a common factor).  Served for one group, c_dz = c_aux = 32, ns = 4 on an LDS-blocked plan; SCN_ERR_UNSUPPORTED otherwise. */
int scn_conv_backward_accumulate(scn_conv_t conv_t, int32_t n_slabs, int32_t ns,
                                 const float* const* dz, const int32_t* c_dz,
                                 const float* const* W, const float* aux, int32_t c_aux, int32_t act,
                                 const float* dx_partial, float* dx, float* const* dW,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* The two Hodge shifts alone: ya = val0-operator * x, yb = val1-operator * x on group 0 of `conv`
 * (the L_down X / L_up X products of TE:146-147 without the dense part; the SpMM GB/s metric).
 *   x, ya, yb device [n_slabs][rows][k]; yb may be NULL for a single-operator product.               */
int scn_spmm_dual(scn_conv_t conv, int32_t n_slabs, int32_t k,
                  const float* x, float* ya, float* yb, void* stream);

/* Dense contraction over already-gathered terms of one Bunch level (TE:184-195 once the S_k X products exist):
 *   forward : out[p,:] = act( sum_k G[k][p,:] @ W[k] ),  G[k] device [n_points][c_in[k]], W[k] device [c_in[k]][c_out]
 *   backward: dx[p,:] = ( sum_k G[k][p,:] @ W[k]^T ) * act'(aux[p,:]) ; dW[k] += sum_p aux[p,:]^T G[k][p,:]
 *             with G[k] = S_k^T dZ device [n_points][c[k]], W[k] device [c_aux][c[k]] (forward weights), aux = this
 *             level's forward input [n_points][c_aux]; dx may be NULL; deterministic reduction order.
 * n_points counts every (slab, row, trajectory) of the level; n_terms <= 3 (<= 6 when every term is one channel wide). */
int scn_dense_terms_forward(int64_t n_points, int32_t n_terms, const float* const* G, const int32_t* c_in,
                            const float* const* W, int32_t c_out, int32_t act, float* out, void* stream);
size_t scn_dense_terms_backward_workspace(int64_t n_points, int32_t n_terms, const int32_t* c, int32_t c_aux);
int scn_dense_terms_backward(int64_t n_points, int32_t n_terms, const float* const* G, const int32_t* c,
                             const float* const* W, const float* aux, int32_t c_aux, int32_t act, float* dx,
                             float* const* dW, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Readout (TE:151-152 with Bconds_func TE:298-303, B1_jax TE:288, nbrhoods TE:279):
 *   logits[n,d] = sum_{e incident to v} sign(v,e) * (H[e,n,:] . w_last),  v = nbr[last[n]][d];  v = -1 -> 0
 *   logp = logits - logsumexp_d(logits)    over ALL D entries, padding included.
 * inc_* is B1 (optionally flipped, TE:291) as node-major CSR on device.
 * max_deg (the neighbourhood width D = the largest node degree, TE:279) up to 1024: one slot per lane up to 64, LDS-resident slot
 * arrays beyond (hub nodes); SCN_ERR_UNSUPPORTED above 1024.  The same holds for the node readout of the Bunch model below.
 * --------------------------------------------------------------------------------------------------- */
int scn_readout_forward(int32_t n_slabs, int32_t ns, int32_t n_edges, int32_t c,
                        const float* H, const float* w_last,
                        const int32_t* nbr, int32_t n_nodes, int32_t max_deg,
                        const int32_t* last_nodes,
                        const int32_t* inc_ptr, const int32_t* inc_edge, const float* inc_sign,
                        float* bh /* [N][max_deg][c]: Bcond(last) @ H, kept for the backward */,
                        float* logits, float* logp, void* stream);

/* Backward of the readout + log-softmax, fused with the last layer's activation derivative:
 *   d_logits = d_logp - exp(logp) * sum_d d_logp
 *   dz[e,n,:] = (sum_v sign(v,e) d_logits[n,d(v)]) * w_last * act'(H[e,n,:])    (zeros elsewhere)
 *   d_w_last += sum_n sum_d d_logits[n,d] * bh[n,d,:]          (fixed summation order)               */
int scn_readout_backward(int32_t n_slabs, int32_t ns, int32_t n_edges, int32_t c,
                         const float* H, const float* w_last,
                         const int32_t* nbr, int32_t n_nodes, int32_t max_deg,
                         const int32_t* last_nodes,
                         const int32_t* inc_ptr, const int32_t* inc_edge, const float* inc_sign,
                         const int32_t* edge_nodes, /* [n_edges][2] endpoints (tail, head) */
                         const float* bh, const float* d_logp, const float* logp, int32_t act,
                         float* d_logits /* [N][max_deg] out */, float* dz,
                         int32_t dz_is_zero /* 1: caller guarantees dz is all zeros (skips the memset of the dense tensor);
                                               2: dz may hold anything and the launch zeroes it itself, every trajectory's wave its own
                                                  column first (small complexes: no fill over the buffer) */,
                         float* d_w_last, void* stream);

/* Writes zeros to exactly the dz entries scn_readout_backward fills for these last_nodes, so a buffer handed in with
 * dz_is_zero = 1 is all-zero again once the layer above has consumed it (a few hundred rows instead of a full memset). */
int scn_readout_clear_dz(int32_t n_slabs, int32_t ns, int32_t n_edges, int32_t c,
                         const int32_t* nbr, int32_t n_nodes, int32_t max_deg, const int32_t* last_nodes,
                         const int32_t* inc_ptr, const int32_t* inc_edge, const int32_t* edge_nodes,
                         float* dz, void* stream);

/* Bunch readout (TE:198-203): logits[n,d] = nodes_out[nbr[last[n]][d]] with -1 wrapping to the last node. */
int scn_node_readout_forward(int32_t n_slabs, int32_t ns, int32_t n_nodes,
                             const float* nodes_out, const int32_t* nbr, int32_t max_deg,
                             const int32_t* last_nodes, float* logits, float* logp, void* stream);
int scn_node_readout_backward(int32_t n_slabs, int32_t ns, int32_t n_nodes,
                              const float* nodes_out, const int32_t* nbr, int32_t max_deg,
                              const int32_t* last_nodes, const float* d_logp, const float* logp,
                              int32_t act, float* dz, void* stream);

/* The first two Bunch layers without a 32-channel gather (bunch_func, TE:179-195): the model starts from [0, flow, 0], so layer
 * one's output of every level is relu of ONE rank-one term, H1_j[p][:] = relu(g_j[p] w_j) with g_j = S x one channel wide, and
 *   relu(g w) = max(g, 0) relu(w) + min(g, 0) min(w, 0)      =>      (S_k H1_j) W_k = (S_k g_j^+) (relu(w_j) W_k) + (S_k g_j^-) (min(w_j, 0) W_k).
 * scn_split_sign:     g_pos = max(g, 0), g_neg = min(g, 0) elementwise (n floats).
 * scn_fold1_forward:  a_pos[c2] = relu(w1)[c1] @ W2[c1][c2],  a_neg = min(w1, 0) @ W2   -- the rank-one weights of layer two's
 *                     expansion (scn_dense_terms_forward over the shifted scalars S_k g^+, S_k g^-).
 * scn_fold1_backward: given u_pos[c2] = sum_p (S_k g^+)[p] dZ2[p][:] and u_neg likewise (scn_dense_terms_backward with dZ2 as aux),
 *                     dW2[a][c] += relu(w1[a]) u_pos[c] + min(w1[a], 0) u_neg[c],
 *                     dw1[a]    += [w1[a] > 0] (u_pos @ W2^T)[a] + [w1[a] < 0] (u_neg @ W2^T)[a]      (relu'(0) = 0, as everywhere). */
int scn_split_sign(int64_t n, const float* g, float* g_pos, float* g_neg, void* stream);

/* out[i] = act(terms[0][i] + ... + terms[n_terms-1][i]), n floats (a multiple of 4, 16-byte aligned tensors), 1 <= n_terms <= 4, summed
 * in term order; out may be terms[0].  Hidden widths above 32 (-hidden_layers, TE:103-110): a layer of scone_func / ebli_func
 * (TE:143-149) runs as its (input block, output block) pairs of 32 channels on the fused kernels; this adds the partial
 * pre-activations of an output block (forward) and the partial input gradients of an input block (backward, act = none). */
int scn_sum_act(int64_t n, int32_t n_terms, const float* const* terms, int32_t act, float* out, void* stream);
int scn_fold1_forward(const float* w1, const float* W2, int32_t c1, int32_t c2, float* a_pos, float* a_neg, void* stream);
int scn_fold1_backward(const float* w1, const float* W2, const float* u_pos, const float* u_neg, int32_t c1, int32_t c2,
                       float* dW2, float* dw1, void* stream);

/* Scatter ragged edge flows into a zeroed slab tensor [n_slabs][n_edges][ns][1] (the flows_in input,
 * SDG:327-344): x[slab(n)][idx][n % ns] += val (repeated (trajectory, edge) entries accumulate, like the reference's
 * f[k] += 1).  sample_of[i] gives the trajectory of entry i. */
int scn_scatter_flows(int32_t n_slabs, int32_t ns, int32_t n_edges, int64_t n_entries,
                      const int32_t* sample_of, const int32_t* edge_idx, const float* val,
                      float* x, void* stream);

/* HOST-ONLY helper of the launch-amortised step (nothing is launched, no device pointer is touched): assemble one batch for
 * scn_scatter_flows + the readout / loss in the caller's staging words (typically pinned memory, copied to the device in one transfer):
 *   out = [sample_of: e_cap][edge: e_cap][val: e_cap floats][last_nodes: n_cap][y / total: n_cap x d floats], unused words 0
 * from the ragged flows (ptr [N + 1], edge / val per entry; path_to_flow, SDG:327-344), last_nodes [N] and targets y [N][d] of the
 * whole data set of n_total trajectories and the batch's trajectory indices traj [m].  Returns the number of flow entries written,
 * SCN_ERR_UNSUPPORTED when the batch does not fit (m > n_cap or more than e_cap entries), SCN_ERR_BAD_ARG when an index lies outside
 * [0, n_total) (nothing is read through it), or another negative status. */
int64_t scn_host_stage_batch(int32_t m, const int32_t* traj, int32_t n_total, const int64_t* ptr, const int32_t* edge, const float* val,
                             const int32_t* last_nodes, const float* y, int32_t d, double total, int32_t e_cap,
                             int32_t n_cap, int32_t* out);

/* Layers whose second shift is the square of the first (Ebli / SNN: S_lower = L1, S_upper = L1^2, TE:155-167, 251-253) on
 * complexes where the rows of the square no longer fit the LDS-blocked plan.  `conv` holds S alone (identity + ONE value
 * array); the caller forms y = S x (forward) or g1 = S^T dz (backward) with scn_spmm_dual and these calls do the rest:
 *   forward :  out = act(x W[0] + y W[1] + (S y) W[2])                                    x0 = x, x = y below
 *   backward:  dx = (dz W[0]^T + g1 W[1]^T + (S^T g1) W[2]^T) * act'(aux),  dW[k] += aux^T (dz | g1 | S^T g1)
 * All tensors [n_slabs][n_rows][ns][channels]; served for channels = 32 and 16 (16: two slabs per visit, like the plain layers;
 * SCN_ERR_UNSUPPORTED / workspace 0 otherwise). */
int scn_conv_forward_power(scn_conv_t conv, int32_t n_slabs, int32_t ns, const float* x0, const float* x,
                           const float* const* W, int32_t channels, int32_t act, float* out, void* stream);
size_t scn_conv_backward_power_workspace(scn_conv_t conv_t, int32_t n_slabs, int32_t ns, int32_t channels);
int scn_conv_backward_power(scn_conv_t conv_t, int32_t n_slabs, int32_t ns, const float* dz, const float* g1,
                            const float* const* W, const float* aux, int32_t channels, int32_t act, float* dx,
                            float* const* dW, void* workspace, size_t workspace_bytes, void* stream);

/* Zero-skipping mode.  Activations of this path have no bias terms, so a layer's output is exactly zero wherever the
 * one-hop closure of its input's support does not reach; a trajectory batch on a large complex leaves most (block, slab)
 * work items all-zero.  A work list names the items that may be non-zero: listed plan blocks (scn_conv_plan_blocks gives
 * their row ranges) and, per block, the slabs to process.  The *_list entry points compute exactly these items and touch
 * nothing else, so the output / gradient buffers must be all-zero outside them on entry (keep persistent zeroed buffers and
 * wipe the listed items with scn_clear_list after use).  Results equal the dense calls.  All arrays are DEVICE pointers.
 * Served by the LDS-blocked C = 32 kernels (and the first layer); other shapes return SCN_ERR_UNSUPPORTED. */
typedef struct scn_work_list {
    int32_t n_work;          /* listed blocks */
    const int32_t* block;    /* [n_work]     plan block index */
    const int32_t* ptr;      /* [n_work + 1] offsets into slab */
    const int32_t* slab;     /* slabs of each listed block, ascending */
} scn_work_list;

/* first row of every plan block (host array of n_blocks + 1 entries; n_blocks from scn_conv_plan_info) */
int scn_conv_plan_blocks(scn_conv_t conv, int32_t* row0_out);
int scn_conv_forward_list(scn_conv_t conv, int32_t n_slabs, int32_t ns, const float* const* src, const int32_t* c_in,
                          const float* const* W, int32_t c_out, int32_t act, float* out, const scn_work_list* wl,
                          void* stream);
int scn_conv_backward_list(scn_conv_t conv_t, int32_t n_slabs, int32_t ns, const float* const* dz, const int32_t* c_dz,
                           const float* const* W, const float* aux, int32_t c_aux, int32_t act, float* dx,
                           float* const* dW, void* workspace, size_t workspace_bytes, const scn_work_list* wl,
                           void* stream);
/* zero the listed items of a [n_slabs][n_rows][ns][channels] tensor */
int scn_clear_list(scn_conv_t conv, int32_t ns, int32_t channels, float* tensor, const scn_work_list* wl, void* stream);

/* First layer (c_in = 1) fast path.
 * scn_conv_forward_first = scn_conv_forward for one 1-channel input, which also stores the shifted input
 *   y[n_slabs][n_rows][ns][4] = (x, S_val0 x, S_val1 x, 0)  -- the three scalars per point the kernel forms anyway, as
 *   16-byte records.
 * scn_conv_dw_first: weight gradient of that layer (no input gradient) with the shift on the 1-channel side,
 *   dW_slot[0][c] += sum_p y[p][slot] * dz[p][c]   (= what jax.grad of TE:144-149 yields for weights[0:3], STM:307):
 *   dz is read exactly once, coalesced.  Pass the y saved by the forward, or y = NULL and x: it is then recomputed into
 *   the workspace.  `conv` is the FORWARD operator (identity + 2 value arrays) in both calls.
 *   x   device [n_slabs][n_rows][ns][1]      dz  device [n_slabs][n_rows][ns][c_dz]     dW[3] device [1][c_dz], accumulated
 * Both return SCN_ERR_UNSUPPORTED (workspace query: 0) when the operator has no blocked plan or the width is not 16 / 32;
 * callers then use scn_conv_forward / scn_conv_backward. */
int scn_conv_forward_first(scn_conv_t conv, int32_t n_slabs, int32_t ns, const float* x, const float* const* W,
                           int32_t c_out, int32_t act, float* out, float* y_out,
                           const scn_work_list* wl /* NULL: dense */, void* stream);
size_t scn_conv_dw_first_workspace(scn_conv_t conv, int32_t n_slabs, int32_t ns, int32_t c_dz);
int scn_conv_dw_first(scn_conv_t conv, int32_t n_slabs, int32_t ns, const float* x, const float* y, const float* dz,
                      int32_t c_dz, float* const* dW, void* workspace, size_t workspace_bytes,
                      const scn_work_list* wl /* NULL: dense; a list needs y */, void* stream);

/* Backward of the layer that FOLLOWS the first one, fused with the first layer's weight gradient (what jax.grad of TE:144-149
 * yields for weights[0:6], STM:307, in one pass):
 *   this layer :  dW[slot] += aux^T G_slot                       (as scn_conv_backward; aux = the first layer's output)
 *   first layer:  dW_first[slot][0][c] += sum_p y[p][slot] * dx[p][c],   dx = (sum_slots G_slot W_slot^T) * act'(aux)
 * dx -- the gradient w.r.t. the first layer's pre-activation, whose only consumer is that weight gradient -- stays in
 * registers and is never written: no [n_slabs][n_rows][ns][channels] tensor, no second pass over it.
 * `conv_t` as in scn_conv_backward; y from scn_conv_forward_first; channels = 32 (SCN_ERR_UNSUPPORTED / workspace 0
 * otherwise: use scn_conv_backward + scn_conv_dw_first).  A work list names the items of dz's support, as usual. */
size_t scn_conv_backward_fused_first_workspace(scn_conv_t conv_t, int32_t n_slabs, int32_t ns, int32_t channels);
int scn_conv_backward_fused_first(scn_conv_t conv_t, int32_t n_slabs, int32_t ns, const float* dz, const float* const* W,
                                  const float* aux, int32_t channels, int32_t act, const float* y, float* const* dW,
                                  float* const* dW_first, void* workspace, size_t workspace_bytes,
                                  const scn_work_list* wl /* NULL: dense */, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Fused Bunch (SCCONV) layer, TE:173-195: the seven shifts S_ab as ONE square operator on the concatenated row space
 * [nodes | edges | faces] (an index space only: the three level tensors stay separate allocations).  Entry (r, c) carries
 * TERM = the level of column c; row r has CLASS = its own level; level_row0[l] = first concatenated index of level l
 * (level_row0[0] = 0, level_row0[3] = n_rows).  Weights are addressed [class][term] (9 pointers, NULL = the class has no such
 * term: nodes have no face term, faces no node term).
 *   scn_terms_forward :  out_l = act( sum_j (S_{j->l} x_j) W[l][j] )   -- one launch for the three levels of a layer
 *       x[j]   device [n_slabs][rows_j][ns][32] or NULL (level identically zero)
 *       out[l] device [n_slabs][rows_l][ns][32] or NULL (level not wanted: skipped)
 * Served for ns = 4, 32 -> 32 channels (SCN_ERR_UNSUPPORTED otherwise: use the per-shift scn_spmm_dual + scn_dense_terms_*).
 * The handle is a scn_conv_t (scn_conv_destroy frees it).
 * --------------------------------------------------------------------------------------------------- */
int scn_terms_create(int32_t n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const uint8_t* term,
                     const int32_t* level_row0 /* [4] */,
                     const uint8_t* merged /* [n_rows]: level of the i-th simplex along ONE locality curve through all three
                                              levels (each level's rows in their own order): blocks are patches along it */,
                     const int32_t* bins /* [3]: rows of each level a block may hold; multiples of rows_per_wave, sum <= 64; 0 = that level's rows belong to no block (its output is never asked for) */,
                     int32_t rows_per_wave /* 4: plan for scn_terms_forward, 8: for scn_terms_backward */, scn_conv_t* out);
int scn_terms_forward(scn_conv_t op, int32_t n_slabs, int32_t ns, const float* const* x /* [3] */,
                      const float* const* W /* [9] = [class][term], each [32][32] */, int32_t channels, int32_t act,
                      float* const* out /* [3] */, void* stream);

/* Backward of the fused Bunch layer on the TRANSPOSED operator (rows = the layer's INPUT rows of every level, terms = the levels
 * a row feeds; a plan created with rows_per_wave = 8):
 *   dx_l = ( sum_j (S_{l->j}^T dz_j) W[l][j]^T ) * act'(aux_l),    dW[l][j] += aux_l^T (S_{l->j}^T dz_j)
 *   dz[j]  device [n_slabs][rows_j][ns][32] or NULL (no gradient reaches level j)
 *   aux[l] device: the layer's forward input of level l, or NULL (level was identically zero: nothing is computed for it)
 *   dx[l]  device or NULL (input gradient of level l not wanted; its weight gradients are still accumulated)
 *   W / dW: [class l][term j] = the FORWARD weight of the shift from level l to level j ([32][32], used transposed) / its gradient
 *           (accumulated, fixed summation order), NULL where the shift does not exist. */
size_t scn_terms_backward_workspace(scn_conv_t op_t, int32_t n_slabs, int32_t ns, int32_t channels);
int scn_terms_backward(scn_conv_t op_t, int32_t n_slabs, int32_t ns, const float* const* dz /* [3] */,
                       const float* const* W /* [9] */, const float* const* aux /* [3] */, int32_t channels, int32_t act,
                       float* const* dx /* [3] */, float* const* dW /* [9] */, void* workspace, size_t workspace_bytes,
                       void* stream);
/* The same backward for the layer that FOLLOWS the 1-channel first layer of bunch_func (TE:179-195): its input gradient is the
 * gradient of the first layer's pre-activation and is needed for the first layer's weight gradient only, so it is contracted
 * with the first layer's shifted input inside the kernel and never written:
 *   dW_first[l][c] += sum_p y_l[p] * dx_l[p][c],   y_l = S_{1->l} x, device [n_slabs][rows_l][ns] (one float per point) or NULL
 *   dW_first[l] device [32] (the (1, 32) weight of the shift into level l) or NULL.  dW as above; same workspace size. */
int scn_terms_backward_fused_first(scn_conv_t op_t, int32_t n_slabs, int32_t ns, const float* const* dz /* [3] */,
                                   const float* const* W /* [9] */, const float* const* aux /* [3] */, int32_t channels,
                                   int32_t act, const float* const* y /* [3] */, float* const* dW /* [9] */,
                                   float* const* dW_first /* [3] */, void* workspace, size_t workspace_bytes, void* stream);

/* Host-only layout helper (no device work, no reference counterpart: the reference's dense operators, TE:240-257, have
 * no storage order).  For a SQUARE CSR pattern (rows and columns share one index space, e.g. L_lower in device order)
 * returns order[new] = old that sorts the rows of every block the plan would cut by descending entry count, so the
 * 8-row groups a wave walks in lock-step are padded less, and block_start[new] = 1 where those blocks begin.
 * identity != 0: the operator also reads the row itself. */
int scn_plan_refine_order(int32_t n, const int32_t* rowptr, const int32_t* col, int32_t identity, int32_t* order,
                          uint8_t* block_start);

/* Host-only diagnostic (no device work, no reference counterpart): simulated LDS cost of one gather pass over the block plan
 * of a SQUARE operator (pattern as for scn_plan_refine_order; val1, nnz floats or NULL, marks the second operator's entries;
 * block_start as for scn_conv_create_blocked or NULL).  out4[0] = lane-group reads (one per quad of rows and entry position),
 * out4[1] = their LDS cycles with the sources in row order and the entries in CSR order, out4[2] = with the plan's layout
 * (slot colours + entry order, csrc/scn_blk_layout.inc; out4[2] == out4[0] means no bank conflict is left), out4[3] = blocks.
 * Every block's layout is also CHECKED: each row's entries placed exactly once inside its quad's width, second-operator entries
 * inside the leading positions, slots a permutation of the sources -- SCN_ERR_INTERNAL otherwise. */
int scn_plan_gather_stats(int32_t n, const int32_t* rowptr, const int32_t* col, const float* val1, int32_t identity,
                          const uint8_t* block_start, int64_t* out4);

/* The gradient step of a scone_func model (TE:137-152 + the loss of STM:42-56) on a SMALL complex in ONE launch -- the reference's own
 * problem sizes (TE:86-90: |E| = 1001, batch 100).  One workgroup per trajectory keeps that trajectory's activations in LDS through
 * every layer, the readout, the cross-entropy and the whole backward (csrc/scn_small.hip); a second launch sums the per-trajectory
 * weight-gradient partials in trajectory order (bitwise reproducible).  Equivalent to
 *   scn_conv_forward_first, scn_conv_forward x (L - 1), scn_readout_forward, scn_masked_ce, scn_readout_backward,
 *   scn_conv_backward x (L - 2), scn_conv_backward_fused_first
 * on the same buffers:  dW[k] += d/dW[k] of  scale * sum_n <logp_n, y_n>,   loss[0] += scale * sum_n <logp_n, y_n>
 * (overwrite != 0: "=" instead of "+=" for both -- the first micro-batch of a step then needs no zeroing launch before it).
 *   conv / conv_t : the operator (identity + S_lower + S_upper on a shared pattern) and its transpose (the same handle for symmetric shifts)
 *   x [n_slabs][n_edges][ns][1], last_nodes [n_slabs*ns], y [n_slabs*ns][max_deg] (zero rows = padding trajectories)
 *   W / dW : 3 * n_layers + 1 matrices in the reference's order (TE:139-152), first layer (1, hidden), last (hidden, 1)
 *   max_items : caller's bound on sum_d (incident edges of neighbour d) over the neighbourhood of any last node
 * Served (scn_small_step_supported): hidden = 16, 2 <= n_layers <= 6, max_deg <= 64, max_items <= 512 and 128 * |E| + 64 KB of LDS
 * within 160 KB (|E| <= ~1100); SCN_ERR_UNSUPPORTED otherwise -- the caller then runs the layer-by-layer entry points.
 * Like every launch here it neither allocates nor copies nor synchronises: the handle's entry pack (col, val_lower, val_upper per
 * entry) is built by scn_conv_create* for every operator of this shape and size, so the FIRST call on a fresh handle may already be
 * captured into a HIP graph (tests/test_gpu_small_step.py).
 * Paired form: when |E| > 384 (at least two blocks of 128 rows) and 2 * n_traj <= the device's CUs, TWO workgroups share a trajectory:
 * alternating 128-row blocks each, full activations in both LDS, rows handed over through memory after every layer (agent-scope stores,
 * a flag per phase; the wait is bounded -- a partner that never arrives makes the loss and the weight gradients NaN, it cannot hang).  Results are those of the
 * single form up to the order of the weight-gradient sums (2 N partials instead of N), still bitwise reproducible.  The launch assumes
 * it has the device to itself for its ~60 us (all workgroups resident together); scn_small_step_pairing(1) turns the form off
 * process-wide (0: back on, the default).  The workspace size covers both forms. */
int scn_small_step_pairing(int32_t mode);
int scn_small_step_supported(scn_conv_t conv, int32_t n_layers, int32_t hidden, int32_t max_deg, int32_t max_items);
size_t scn_small_step_workspace(int32_t n_edges, int32_t n_traj, int32_t n_layers);
int scn_small_step(scn_conv_t conv, scn_conv_t conv_t, int32_t n_slabs, int32_t ns, int32_t n_layers, int32_t hidden,
                   const float* x, const int32_t* last_nodes, const float* y, float scale, const int32_t* nbr, int32_t n_nodes,
                   int32_t max_deg, int32_t max_items, const int32_t* inc_ptr, const int32_t* inc_edge, const float* inc_sign,
                   const float* const* W, int32_t act, float* const* dW, double* loss, int32_t overwrite, void* workspace,
                   size_t workspace_bytes, void* stream);
/* The same step when the batch is ONE micro-batch on ONE rank, ending with the optimiser step: dW[k] = gradient, loss[0] = loss
 * (the overwrite form), and the launch that sums the gradient applies scn_adam_step's update (g_scale = 1) to the weights, which must
 * be ONE flat buffer w_flat with W[k] pointing into it in list order (m_flat / v_flat alike; SCN_ERR_BAD_ARG otherwise).  The step index
 * comes from step_dev[0] as for scn_adam_step_dev (step_dev[1] is scratch: the index in flight); step_dev[0] is advanced.  Bitwise the
 * weights of scn_small_step(overwrite) followed by scn_adam_step_dev. */
int scn_small_step_adam(scn_conv_t conv, scn_conv_t conv_t, int32_t n_slabs, int32_t ns, int32_t n_layers, int32_t hidden,
                        const float* x, const int32_t* last_nodes, const float* y, float scale, const int32_t* nbr, int32_t n_nodes,
                        int32_t max_deg, int32_t max_items, const int32_t* inc_ptr, const int32_t* inc_edge, const float* inc_sign,
                        const float* const* W, int32_t act, float* const* dW, double* loss, void* workspace, size_t workspace_bytes,
                        float* w_flat, float* m_flat, float* v_flat, float lr, float b1, float b2, float eps, int32_t* step_dev,
                        float weight_decay, void* stream);

/* Readout of a last layer kept as channel blocks (hidden widths above 32): logits are linear in H, so the blocks' logits
 * (scn_readout_forward per block with its rows of W_last) are added and normalised here:
 *   logits[n,:] = sum_k logits_parts[k][n,:],   logp = logits - logsumexp_d(logits)   (all D entries, padding included: TE:151-152)
 * n_parts <= 4; logits may be logits_parts[0]. */
int scn_logits_sum_log_softmax(int32_t n_traj, int32_t max_deg, int32_t n_parts, const float* const* logits_parts,
                               float* logits, float* logp, void* stream);

/* Masked cross-entropy of one micro-batch (the data term of STM:54 and its gradient w.r.t. the log-probabilities):
 *   d_logp[i] = y[i] * scale   (scale = -1 / number of trajectories in the GLOBAL batch; padding rows have y = 0)
 *   loss[0]  += sum_i logp[i] * d_logp[i]    (fp64 accumulator on the device, fixed summation order)
 * n = trajectories x max_deg entries of logp / y / d_logp. */
int scn_masked_ce(int64_t n, const float* logp, const float* y, float scale, float* d_logp, double* loss, void* stream);
/* The same as the FIRST loss launch of an optimiser step: overwrite != 0 SETS loss[0] instead of adding to it, and zero_buf (or NULL)
 * -- the flat weight-gradient buffer the backward launches that follow accumulate into, zero_n floats -- is zeroed by the same launch
 * (on the reference's own problem sizes a step is ~15 launches of 5-15 us: two fill launches are a tenth of it). */
int scn_masked_ce_begin(int64_t n, const float* logp, const float* y, float scale, float* d_logp, double* loss,
                        int32_t overwrite, float* zero_buf, int64_t zero_n, void* stream);

/* Fused Adam + ridge step on the flat parameter buffer (jax.experimental.optimizers.adam as driven by
 * STM:300-326; ridge term of STM:54-56):   g' = g * g_scale + 2*weight_decay*w ; m,v EMA ;
 *   w -= lr * (m / (1 - b1^(i+1))) / (sqrt(v / (1 - b2^(i+1))) + eps)                                */
int scn_adam_step(int64_t n, float* w, const float* g, float* m, float* v,
                  float lr, float b1, float b2, float eps, int32_t step_i,
                  float weight_decay, float g_scale, void* stream);
/* The same update with the step index in device memory: reads i = step_dev[0], applies update i, leaves i + 1 in step_dev[0].
 * The launch's arguments do not change from step to step, so it can be captured into a HIP graph behind the gradient launches
 * (scone_trajectory_model.py does, on one rank: the plain launch after a graph replay costs ~8 us of an ~80 us step on the
 * reference's own problem sizes).  One workgroup: n <= SCN_ADAM_DEV_MAX (every model of the reference is far below),
 * SCN_ERR_UNSUPPORTED beyond.  Bitwise the same weights as scn_adam_step(step_i = i). */
#define SCN_ADAM_DEV_MAX 65536
int scn_adam_step_dev(int64_t n, float* w, const float* g, float* m, float* v,
                      float lr, float b1, float b2, float eps, int32_t* step_dev,
                      float weight_decay, float g_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCONE_HIP_H */
