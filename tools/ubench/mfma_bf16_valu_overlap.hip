// -DSCALAR_FMA: the VALU work is v_fma_f32 (inline asm) instead of compiler-packed v_pk_fma_f32.
// Does a bf16 MFMA chain (v_mfma_f32_32x32x16_bf16) overlap with VALU work of the co-resident wave of the same SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>   // 0: all MFMA, 1: all VALU, 2: waves 0-3 MFMA + 4-7 VALU, 3: waves 0-3 MFMA only, 4: waves 4-7 VALU only, 5: every wave alternates MFMA and VALU
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float a = threadIdx.x * 1e-3f, b = 1.0001f;
    bool do_mfma = MODE == 0 || ((MODE == 2 || MODE == 3) && wave < 4);
    bool do_valu = MODE == 1 || ((MODE == 2 || MODE == 4) && wave >= 4);
    float res = 0.f;
    bf16x8 va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = (__bf16)(a + i); vb[i] = (__bf16)(b + i); }
    if (MODE == 5) {
        f32x16 acc = {0};
        f32x4 x0 = {a, a, a, a}, x1 = x0 * 2.f, x2 = x0 * 3.f, x3 = x0 * 4.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc, 0, 0, 0);
                x0 = x0 * b + x1; x1 = x1 * b + x2; x2 = x2 * b + x3; x3 = x3 * b + x0;
            }
        }
        res = acc[0] + acc[5] + x0[0] + x1[1] + x2[2] + x3[3];
    } else if (do_mfma) {
        f32x16 acc = {0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc, 0, 0, 0);
        }
        res = acc[0] + acc[5];
    } else if (do_valu) {
        f32x4 x0 = {a, a, a, a}, x1 = x0 * 2.f, x2 = x0 * 3.f, x3 = x0 * 4.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
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
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<256, 512>>>(d, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<MODE><<<256, 512>>>(d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    const int it = 20000;
    printf("bf16: all-MFMA %.3f | all-VALU %.3f | half MFMA + half VALU %.3f | half MFMA alone %.3f | half VALU alone %.3f | interleaved in every wave %.3f\n",
           run<0>(d, it), run<1>(d, it), run<2>(d, it), run<3>(d, it), run<4>(d, it), run<5>(d, it));
    return 0;
}
