// LDS-blocked kernels for the scone / ebli layer shape: ONE group, identity + two value arrays on a shared
// pattern (slots: identity, S_lower, S_upper), ns = 4, channel widths 16 or 32 (MFMA) and 1 (first layer).
//
// Execution plan (built once per operator on the host, build_block_plan):
//   the output rows are cut into blocks of <= 64 consecutive rows whose gather sources (<= 128 distinct source
//   rows) are staged once per slab into LDS; the block's CSR slice is re-expressed as a padded ELL tile with
//   LOCAL source slots (uint8) so the inner loop touches LDS only.  Rows are expected in a locality order
//   (Hilbert order of the edge midpoints, scone_gcn_amd/complex.py) so a block's sources are ~2x its rows.
//
// Kernel structure: one workgroup per CU (8 or 16 waves), grid-strided over blocks with an XCD-contiguous mapping.
//   for block: ELL tile + source list -> LDS
//     for slab: [s_waitcnt vmcnt(0); s_barrier]  -> slab s has landed in buffer s&1, everybody left buffer (s+1)&1
//               issue LDS-DMA (global_load_lds, 16 B/lane, per-lane source = gather) of slab s+1 into the other buffer
//               compute slab s: each wave gathers its rows x 4 trajectories with lane = (point, channel slice)
//               from the XOR-swizzled LDS image and feeds [self | lower | upper] into f16 MFMAs on a two-way hi + lo split under
//               power-of-two row scales (three products, fp32 accumulation: scn_blk_fwd.inc) against split weights in LDS;
//               the activation epilogue's stores are deferred by one slab so the next vmcnt(0) never waits on stores
//               that were just issued.
// Parts (one translation unit): scn_blk_layout.inc + scn_blk_plan.inc (host: block plan), scn_blk_common.inc (LDS layout, gather, pipeline helpers),
// scn_blk_spmm.inc (dual SpMM), scn_blk_fwd.inc / scn_blk_bwd.inc (fused layer kernels), scn_blk_first.inc (first-layer
// gradient streams, reductions), scn_blk_dispatch.inc (host dispatch), scn_terms.inc (fused Bunch layer).
#include <algorithm>
#include <array>
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
// per wave INDEX of the workgroup (summed over workgroups): [0..5] the segment sums, [6] sum of skew^2 / 1024 (segment 1 = the slab
// barrier), [7] barrier intervals, [8] sum of the SIMD id the wave ran on (HW_REG_HW_ID bits 5:4), [9] waves
__device__ unsigned long long g_stamps_w[16][10];
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_DECL unsigned long long t0_ = 0, t1_ = 0, acc_[6] = {0, 0, 0, 0, 0, 0}, sq_ = 0, nint_ = 0
#define STAMP_ADD(i) do { STAMP(t1_); acc_[i] += t1_ - t0_; if ((i) == 1) { sq_ += ((t1_ - t0_) * (t1_ - t0_)) >> 10; ++nint_; } t0_ = t1_; } while (0)
#define STAMP_START() STAMP(t0_)
#define STAMP_FLUSH() do { if ((threadIdx.x & 63) == 0) { unsigned long long s_ = 0; const int w_ = (threadIdx.x >> 6) & 15; \
    for (int i_ = 0; i_ < 6; ++i_) { atomicAdd(&g_stamps[i_], acc_[i_]); atomicAdd(&g_stamps_w[w_][i_], acc_[i_]); s_ += acc_[i_]; } \
    atomicMax(&g_stamps[6], s_); atomicAdd(&g_stamps[7], 1ull); atomicAdd(&g_stamps_w[w_][6], sq_); atomicAdd(&g_stamps_w[w_][7], nint_); \
    atomicAdd(&g_stamps_w[w_][8], (unsigned long long)((__builtin_amdgcn_s_getreg((4 << 0) | (4 << 6) | ((2 - 1) << 11))) & 3)); \
    atomicAdd(&g_stamps_w[w_][9], 1ull); } } while (0)
#else
#define STAMP_DECL
#define STAMP_ADD(i)
#define STAMP_START()
#define STAMP_FLUSH()
#endif

#include "scn_blk_layout.inc"
#include "scn_blk_plan.inc"
#include "scn_blk_common.inc"
#include "scn_blk_spmm.inc"
#include "scn_blk_fwd.inc"
#include "scn_blk_bwd.inc"
#include "scn_blk_first.inc"
#include "scn_blk_dispatch.inc"
#include "scn_terms.inc"

}  // namespace scn


// Host-only layout helper (no reference counterpart: the reference's dense operators have no storage order).
// All lanes of a wave walk the ELL rows of their 8-row group to the group's widest row, so a block whose rows are sorted
// by entry count wastes fewer gather iterations (1.25x -> 1.09x of nnz at |E| = 1M); the sorted 8-row groups are then dealt to
// the SIMDs evenly (see below).  For a SQUARE pattern whose rows and
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
        // the 8-row groups now come widest first, and wave i of a workgroup runs on SIMD i mod 4: the first four groups go in
        // REVERSE (width ranks 3 2 1 0 4 5 6 7), so that the two groups a SIMD hosts in the 8-wave kernels -- and the four quads
        // it hosts in the 16-wave kernels -- add up to about the same gather work (widest with narrowest, which may be cut short)
        if (rows > 32) {
            std::swap_ranges(order + r0, order + r0 + 8, order + r0 + 24);
            std::swap_ranges(order + r0 + 8, order + r0 + 16, order + r0 + 16);
        }
        r0 += rows;
    }
    return SCN_OK;
}

// Host-only diagnostic (no GPU involved): the simulated LDS cost of one gather pass over a square operator's block plan --
// lane-group reads (out[0]) and their LDS cycles with the sources in row order and the entries in CSR order (out[1]) and with
// the plan's layout (out[2], scn_blk_layout.inc); out[3] = blocks.  val1 (nullable) marks the entries of the second operator.
extern "C" int scn_plan_gather_stats(int32_t n, const int32_t* rowptr, const int32_t* col, const float* val1, int32_t identity,
                                     const uint8_t* block_start, int64_t* out4) {
    if (n < 0 || !rowptr || !out4 || (n > 0 && !col && rowptr[n] > 0)) return SCN_ERR_BAD_ARG;
    for (int64_t j = 0; j < rowptr[n]; ++j)
        if (col[j] < 0 || col[j] >= n) return SCN_ERR_BAD_SHAPE;
    std::vector<int32_t> mark(n, -1), local(n, 0), cur;
    scn::BlockLayout L;
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    int bid = 0;
    for (int r0 = 0; r0 < n; ++bid) {
        int rows = 0, w = 0, r_limit = n;
        if (block_start)
            for (int r = r0 + 1; r < std::min(n, r0 + scn::BK_R + 1); ++r)
                if (block_start[r]) { r_limit = r; break; }
        scn::grow_block(rowptr, col, r_limit, identity != 0, r0, bid, mark, cur, rows, w);
        if (rows == 0) return SCN_ERR_UNSUPPORTED;
        std::sort(cur.begin(), cur.end());
        for (size_t i = 0; i < cur.size(); ++i) local[cur[i]] = (int32_t)i;
        for (int mode = 0; mode < 2; ++mode) {
            scn::layout_input(L, rowptr, col, val1, identity != 0, r0, rows, w, (int)cur.size(), local);
            scn::layout_block(L, mode == 1);
            if (!scn::layout_valid(L)) return SCN_ERR_INTERNAL;
            out4[1 + mode] += L.cycles;
        }
        out4[0] += L.positions;
        ++out4[3];
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
extern "C" int scn_debug_stamps_waves(unsigned long long* out160, int reset) {
    if (hipMemcpyFromSymbol(out160, HIP_SYMBOL(scn::g_stamps_w), 1280) != hipSuccess) return -3;
    if (reset) { unsigned long long z[160] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(scn::g_stamps_w), z, 1280) != hipSuccess) return -3; }
    return 0;
}
#endif
