// LDS-blocked kernels (placeholder: plan building and MFMA kernels are added next).
#include "scn_internal.h"

namespace scn {
int build_block_plan(scn_conv_s*) { return SCN_OK; }
void free_block_plan(scn_conv_s*) {}
bool blocked_forward_supported(const scn_conv_s*, int, const int32_t*, int) { return false; }
int blocked_forward(scn_conv_s*, int, int, const float* const*, const int32_t*, const float* const*, int, int,
                    float*, hipStream_t) { return SCN_ERR_UNSUPPORTED; }
bool blocked_backward_supported(const scn_conv_s*, int, const int32_t*, int, bool) { return false; }
size_t blocked_backward_workspace(const scn_conv_s*, int, int, const int32_t*, int) { return 0; }
int blocked_backward(scn_conv_s*, int, int, const float* const*, const int32_t*, const float* const*, const float*,
                     int, int, float*, float* const*, void*, size_t, hipStream_t) { return SCN_ERR_UNSUPPORTED; }
bool blocked_spmm_supported(const scn_conv_s*, int) { return false; }
int blocked_spmm(scn_conv_s*, int, int, const float*, float*, float*, hipStream_t) { return SCN_ERR_UNSUPPORTED; }
}  // namespace scn
