// Issue rate of the VALU instructions the bf16x3 split is made of (v_and_b32 / v_sub_f32 / v_perm_b32), of v_fma_f32 and of
// v_pk_fma_f32, with one and with two resident waves per SIMD.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, int active_waves) {
    const int wave = threadIdx.x >> 6;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
    float r = 0.f;
    if (wave < active_waves) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (MODE == 0) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x[i]));
                    if (MODE == 1) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
                    if (MODE == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 1) & 7]), "s"(0x07060302));
                    if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
                    if (MODE == 4 && (i & 1) == 0)
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&x[i]) : "v"(*(double*)&x[(i + 2) & 7]));
                    if (MODE == 5) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
                    if (MODE == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
                }
            }
        }
        for (int i = 0; i < 8; ++i) r += x[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int MODE> void run(float* d, const char* name, int per_iter) {
    const int iters = 20000;
    for (int aw = 4; aw <= 8; aw += 4) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<MODE><<<256, 512>>>(d, iters, aw); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); k<MODE><<<256, 512>>>(d, iters, aw); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double ops_per_simd = (double)iters * per_iter * (aw / 4);
        printf("%-14s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instruction per SIMD\n", name, aw / 4, ms, ms * 1e6 / ops_per_simd);
    }
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    run<0>(d, "v_and_b32", 64); run<1>(d, "v_sub_f32", 64); run<2>(d, "v_perm_b32", 64); run<3>(d, "v_fma_f32", 64);
    run<4>(d, "v_pk_fma_f32", 32); run<5>(d, "v_mov_b32", 64); run<6>(d, "v_cndmask_b32", 64);
    return 0;
}
