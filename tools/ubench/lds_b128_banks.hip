// Which lanes of a ds_read_b128 are served together?  addr(lane) = 128*lane*K + 16*g(lane): every lane reads its own
// 16 bytes, g(lane) picks the bank group (address bits 4-6).  Cycles per wave-instruction for several g reveal the grouping.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ int gsel(int mode, int l) {
    switch (mode) {
        case 0: return l & 7;                 // distinct within 8 consecutive lanes
        case 1: return (l >> 3) & 7;          // same within 8 consecutive, distinct across lanes l, l+8, ...
        case 2: return (l >> 1) & 7;
        case 3: return (l >> 2) & 7;
        case 4: return l & 3;                 // 4 groups
        case 5: return 0;                     // one group
        case 6: return (l & 3) | (((l >> 4) & 1) << 2);   // distinct within {l&3, bit4}
        case 7: return (l & 3) | (((l >> 5) & 1) << 2);   // distinct within {l&3, bit5}
        case 8: return (l & 3) | (((l >> 2) & 1) << 2);   // = l&7
        case 9: return (l & 1) | (((l >> 4) & 3) << 1);   // {bit0, bit4, bit5}
        case 10: return ((l >> 4) & 3) | (((l >> 3) & 1) << 2);  // {bit4,bit5,bit3}
        default: return (l >> 4) & 3;         // 4 groups by l>>4
    }
}
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, int waves) {
    extern __shared__ char lds[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) ((float*)lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc = {0, 0, 0, 0};
    if (wave < waves) {
        const int base = (lane * 128 * 7 + 16 * gsel(mode, lane)) & 0xffff & ~15;   // 128*7*lane keeps bits 4-6 from g only
        const f32x4* p = (const f32x4*)(lds + ((base & ~0x70) | (16 * gsel(mode, lane))));
        const long long t0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                f32x4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)p), "n"(0));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                acc += v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long t1 = clock64();
        if (lane == 0 && wave == 0 && blockIdx.x == 0) out[1024] = (float)(t1 - t0) / (iters * 16.0f);
    }
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main() {
    float* d; (void)hipMalloc(&d, 4096 * 4);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int waves = 1; waves <= 8; waves *= 8)
        for (int mode = 0; mode <= 11; ++mode) {
            k<<<1, 512, 65536>>>(d, 2000, mode, waves); (void)hipDeviceSynchronize();
            float r; (void)hipMemcpy(&r, d + 1024, 4, hipMemcpyDeviceToHost);
            printf("waves %d mode %2d: %.1f clock64 ticks per ds_read_b128 (wave 0)\n", waves, mode, r);
        }
    return 0;
}
