// micro-benchmark 3: the sweep's inner loop in isolation.  W waves per SIMD, each: per "half-tile" 9 ds_read_b128 of B fragments
// (row stride 304 B, conflict-free) feeding 18 MFMAs on UA accumulator chains.  Reports the matrix-pipe utilisation.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int WPS, int UA, int PF, bool LDSB>
__global__ void __launch_bounds__(256 * WPS) k(unsigned* out, int iters, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 4 * 9728 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = seed * i + 17;
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    u32x4 a[UA][9];
    for (int m = 0; m < 9; ++m)
        for (int u = 0; u < UA; ++u) a[u][m] = u32x4{seed + m, seed + 1 + u, seed + 2 + threadIdx.x, seed + 3};
    f32x16 acc[UA];
    const unsigned char* base = smem + j * 304 + 16 * h;
    u32x4 bq[PF];
    for (int m = 0; m < PF; ++m) bq[m] = *reinterpret_cast<const u32x4*>(base + 32 * m);
    unsigned sink = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const unsigned char* tb = base + (it & 3) * 9728;
        const unsigned char* tbn = base + ((it + 1) & 3) * 9728;
#pragma unroll
        for (int m = 0; m < 9; ++m) {
#pragma unroll
            for (int u = 0; u < UA; ++u)
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[u][m]), __builtin_bit_cast(bf16x8, bq[m % PF]), m == 0 ? (f32x16)(0.f) : acc[u], 0, 0, 0);
            if constexpr (LDSB) {
                if (m + PF < 9) bq[m % PF] = *reinterpret_cast<const u32x4*>(tb + 32 * (m + PF));
                else bq[m % PF] = *reinterpret_cast<const u32x4*>(tbn + 32 * (m + PF - 9));
            }
        }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int u = 0; u < UA; ++u) asm volatile("" ::"v"(acc[u]));
#endif
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (unsigned)(t1 - t0); out[1] = sink; }
}
template <int WPS, int UA, int PF, bool LDSB>
void run(unsigned* d, const char* name) {
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<WPS, UA, PF, LDSB>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<WPS, UA, PF, LDSB>), dim3(256), dim3(256 * WPS), 40960, 0, d, iters, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<WPS, UA, PF, LDSB>), dim3(256), dim3(256 * WPS), 40960, 0, d, iters, 12345u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)iters * 9 * UA * WPS;      // per SIMD
    printf("%-34s waves/SIMD %d  chains %d  prefetch %d: %6.2f ns per MFMA per SIMD  (%5.1f TFLOP/s chip)\n", name, WPS, UA, PF, ms * 1e6 / mfmas,
           mfmas * 1024 * 32768.0 / (ms * 1e-3) / 1e12);
}
int main() {
    unsigned* d;
    hipMalloc(&d, 64);
    run<1, 2, 4, false>(d, "B in registers");
    run<1, 4, 4, false>(d, "B in registers");
    run<1, 8, 4, false>(d, "B in registers");
    run<1, 8, 4, true>(d, "B from LDS");
    run<2, 2, 4, false>(d, "B in registers");
    run<2, 4, 4, false>(d, "B in registers");
    run<2, 4, 4, true>(d, "B from LDS");
    run<3, 2, 4, false>(d, "B in registers");
    run<3, 2, 4, true>(d, "B from LDS");
    run<4, 1, 4, false>(d, "B in registers");
    run<4, 2, 4, false>(d, "B in registers");
    return 0;
}
