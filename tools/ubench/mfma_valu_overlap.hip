// Does an f32 MFMA chain on one wave overlap with VALU (v_pk_fma_f32 / v_fma_f32) work of the co-resident wave of the
// same SIMD on gfx950?  8 waves per CU (512-thread workgroup, one per CU): waves 0-3 and 4-7 share SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: all MFMA, 1: all VALU, 2: waves 0-3 MFMA + 4-7 VALU, 3: waves 0-3 MFMA only (4-7 idle), 4: 4-7 VALU only
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float a = threadIdx.x * 1e-3f, b = 1.0001f;
    bool do_mfma = MODE == 0 || ((MODE == 2 || MODE == 3) && wave < 4);
    bool do_valu = MODE == 1 || ((MODE == 2 || MODE == 4) && wave >= 4);
    float res = 0.f;
    if (do_mfma) {
        f32x16 acc = {0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        res = acc[0] + acc[5];
    } else if (do_valu) {
        f32x4 x0 = {a, a, a, a}, x1 = x0 * 2.f, x2 = x0 * 3.f, x3 = x0 * 4.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {   // 16 * 4 float4 fma = 64 v_fma lanes-instrs (32 pk) per iteration
#ifdef SCALAR_FMA
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0[j]) : "v"(b), "v"(x1[j]));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1[j]) : "v"(b), "v"(x2[j]));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2[j]) : "v"(b), "v"(x3[j]));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3[j]) : "v"(b), "v"(x0[j]));
                }
#else
                x0 = x0 * b + x1; x1 = x1 * b + x2; x2 = x2 * b + x3; x3 = x3 * b + x0;
#endif
            }
        }
        res = x0[0] + x1[1] + x2[2] + x3[3];
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}
template <int MODE> float run(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 512>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<256, 512>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int it = 20000;
    printf("all-MFMA %.3f ms | all-VALU %.3f ms | half MFMA + half VALU %.3f ms | half MFMA alone %.3f | half VALU alone %.3f\n",
           run<0>(d, it), run<1>(d, it), run<2>(d, it), run<3>(d, it), run<4>(d, it));
    return 0;
}
