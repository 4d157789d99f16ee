// Internal declarations shared by the translation units of libscone_hip.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <map>
#include <vector>

#include "scone_hip.h"

namespace scn {

void set_hip_error(hipError_t e, const char* where);

#define SCN_HIP_TRY(expr)                                   \
    do {                                                    \
        hipError_t _e = (expr);                             \
        if (_e != hipSuccess) {                             \
            ::scn::set_hip_error(_e, #expr);                \
            return SCN_ERR_HIP;                             \
        }                                                   \
    } while (0)

#define SCN_LAUNCH_CHECK()                                  \
    do {                                                    \
        hipError_t _e = hipGetLastError();                  \
        if (_e != hipSuccess) {                             \
            ::scn::set_hip_error(_e, "kernel launch");      \
            return SCN_ERR_HIP;                             \
        }                                                   \
    } while (0)

// activation and its derivative expressed through the OUTPUT value y (what the backward has at hand)
// tanh to ~3e-7 absolute: odd Taylor polynomial below 1/8, 1 - 2/(e^{2|x|}+1) on v_exp_f32 / v_rcp_f32 above
__device__ __forceinline__ float fast_tanh(float x) {
    const float ax = fabsf(x), x2 = x * x;
    const float poly = x * fmaf(x2, fmaf(x2, fmaf(x2, -17.f / 315.f, 2.f / 15.f), -1.f / 3.f), 1.f);
    const float t = __builtin_amdgcn_exp2f(ax * 2.885390081777927f);
    const float r = 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
    return ax < 0.125f ? poly : copysignf(r, x);
}
__device__ __forceinline__ float act_apply_fast(int act, float z) {
    switch (act) {
        case SCN_ACT_TANH: return fast_tanh(z);
        case SCN_ACT_RELU: return fmaxf(z, 0.f);
        case SCN_ACT_LEAKY_RELU: return z >= 0.f ? z : 0.01f * z;
        default: return z;
    }
}
__device__ __forceinline__ float act_apply(int act, float z) {
    switch (act) {
        case SCN_ACT_TANH: return tanhf(z);
        case SCN_ACT_RELU: return fmaxf(z, 0.f);
        case SCN_ACT_LEAKY_RELU: return z >= 0.f ? z : 0.01f * z;
        default: return z;
    }
}
__device__ __forceinline__ float act_grad_from_output(int act, float y) {
    switch (act) {
        case SCN_ACT_TANH: return 1.f - y * y;
        case SCN_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case SCN_ACT_LEAKY_RELU: return y >= 0.f ? 1.f : 0.01f;
        default: return 1.f;
    }
}

// One parameter's ridge + Adam update (scn_adam_step, STM:300-326), every multiply-add spelled out: all kernels that apply it must round
// alike (left to the compiler's contraction, `a * b + c * d` became fma(a, b, c * d) in one kernel and fma(c, d, a * b) in another).
// c1 = 1 - b1^(i+1), c2 = 1 - b2^(i+1) (adam_corrections); wd2 = 2 * weight_decay.
__device__ __forceinline__ void adam_update(float* __restrict__ w, float g_raw, float* __restrict__ m, float* __restrict__ v, float lr,
                                            float b1, float b2, float eps, float c1, float c2, float wd2, float g_scale) {
#pragma clang fp contract(off)
    const float wi = *w;
    const float gi = fmaf(g_raw, g_scale, wd2 * wi);
    const float mi = fmaf(1.f - b1, gi, b1 * *m);
    const float vi = fmaf((1.f - b2) * gi, gi, b2 * *v);
    *m = mi;
    *v = vi;
    *w = wi - lr * (mi / c1) / (sqrtf(vi / c2) + eps);
}
__host__ __device__ __forceinline__ void adam_corrections(float b1, float b2, int i, float& c1, float& c2) {
    c1 = 1.f - (float)pow((double)b1, (double)(i + 1));
    c2 = 1.f - (float)pow((double)b2, (double)(i + 1));
}

struct Group {
    int32_t n_cols = 0, identity = 0, n_vals = 0;
    int64_t nnz = 0;
    int32_t slot_base = 0, n_slots = 0;
    // device copies
    int32_t* d_rowptr = nullptr;
    int32_t* d_col = nullptr;
    float* d_val0 = nullptr;
    float* d_val1 = nullptr;
    // host copies (plan building)
    std::vector<int32_t> h_rowptr, h_col;
    std::vector<float> h_val0, h_val1;
};

// ---- LDS-blocked execution plan of a single-group operator (scn_blocked.hip) ----
constexpr int BK_WAVES = 8;
constexpr int BK_R = 8 * BK_WAVES;   // output rows per block (upper bound; 8 per wave)
constexpr int BK_SRC = 128;     // staged source pieces per block (upper bound)
constexpr int BK_NS = 4;        // trajectories per slab the blocked kernels are built for
constexpr int BK_THREADS = 64 * BK_WAVES;
constexpr int BK_ELL_CAP = 1152;  // ELL entries (rows-in-block x padded width) a block may carry (64 rows x 18; 4 % of the blocks of the
                                  // benchmark complex are cut a few rows short by it: the 2 KB go to the backward's selection fragments)

// device view passed to kernels by value
struct PlanDev {
    int32_t n_blocks, ell_w_max;
    const int32_t* blk_row0;    // [n_blocks] first output row
    const uint8_t* blk_rows;    // [n_blocks] rows in block (<= BK_R)
    const int32_t* src_ptr;     // [n_blocks+1]
    const int32_t* src_rows;    // staged source rows by slot (slot & 3 = the LDS bank class of the source: scn_blk_layout.inc)
    const int32_t* ell_ptr;     // [n_blocks] entry offset (entries are [t][BK_R])
    const uint8_t* width;       // [n_blocks] padded entries per row
    const uint8_t* tile_w;      // [n_blocks][BK_WAVES] entries needed by each wave's 8 rows
    const uint8_t* tile_w4;     // [n_blocks][2*BK_WAVES] the same per QUAD: rows {0,3,5,6} / {1,2,4,7} of an 8-row group (16-wave kernels)
    const uint8_t* tile_wu;     // [n_blocks][BK_WAVES] leading entries that carry a non-zero val1 (rows are ordered that way)
    const int32_t* assign;      // optional [n_blocks]: visit position -> block (cost-balanced static assignment, per launch grid)
    const uint8_t* tile_wu4;    // [n_blocks][2*BK_WAVES]
    const uint8_t* ell_slot;    // local slot of entry
    const uint16_t* ell_enc;    // the same for 512-byte pieces, ready to XOR into an LDS address: slot*512 | (slot&3)*32
    const float2* ell_v;        // (val0, val1) of entry
    const uint8_t* self_slot;   // [n_blocks][BK_R] slot of the row itself (identity shift)
};

// optional (block, slab) work list of the zero-skipping mode (scn_work_list); block == nullptr: every block, every slab
struct WorkList {
    int32_t n_work;
    const int32_t* block;   // [n_work] plan block index
    const int32_t* ptr;     // [n_work + 1] offsets into slab
    const int32_t* slab;    // active slabs of each listed block, ascending
};

// extra plan arrays of a "terms" operator (fused Bunch layer, scn_terms.inc); the common part lives in BlockPlan
struct TermsPlan {
    bool built = false;
    int group_rows = 0;                 // rows per wave of the kernel the plan was cut for (4: forward, 8: backward)
    int32_t bins[3] = {0, 0, 0};        // rows of each level a block can hold
    const int32_t* blk_row0 = nullptr;  // [n_blocks][3]
    const uint8_t* blk_rows = nullptr;  // [n_blocks][4]
    const uint8_t* blk_w = nullptr;     // [n_blocks][4]
    const uint8_t* grp = nullptr;       // [n_blocks][16][5]
    int32_t lvl_row0[4] = {0, 0, 0, 0};
};

struct BlockPlan {
    bool built = false;
    PlanDev dev{};
    double mean_src_per_row = 0.0;
    int64_t gather_positions = 0, gather_cycles = 0;   // simulated lane-group reads of one gather pass and their LDS cycles (scn_blk_layout.inc)
    std::vector<int32_t> h_row0;   // first row of every block (+ n_rows): scn_conv_plan_blocks
    std::vector<float> h_cost;     // relative cost of one slab of every block (balanced_assignment)
    std::map<int, const int32_t*> assign_by_grid;   // grid.x -> device table, built with the plan
    std::vector<void*> allocs;
};

}  // namespace scn

struct scn_conv_s {
    int32_t n_rows = 0, n_groups = 0, n_slots = 0;
    scn::Group g[SCN_MAX_GROUPS];
    int32_t slot_group[SCN_MAX_SLOTS] = {0, 0, 0, 0};
    int32_t slot_kind[SCN_MAX_SLOTS] = {0, 0, 0, 0};   // 0 identity, 1 val0, 2 val1
    scn::BlockPlan plan;
    scn::TermsPlan terms;
    void* small_pack = nullptr;         // (col, val0, val1, 0) per entry for scn_small_step, built by scn_conv_create* (small_prepare; owned through plan.allocs)
    void* small_ell = nullptr;          // the first twelve entries of every row at a fixed stride, columns as LDS offsets: [n_rows][12] float4 (same owner)
    void* small_ovf = nullptr;          // the entries past the twelfth of each row in the same form, small_n_ovf of them,
    int32_t* small_ovf_ptr = nullptr;   // [n_rows + 1] a row's range among them
    int32_t small_n_ovf = 0;
    std::vector<uint8_t> block_start;   // optional layout hint: 1 where a block of the plan must start (see scn_plan_refine_order)
};

namespace scn {
int small_prepare(scn_conv_s* c);      // scn_small.hip: the entry pack + LDS limit of scn_small_step, at create time
}
