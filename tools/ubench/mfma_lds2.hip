// micro-benchmark 4: the sweep's block loop as it is now -- W MFMA waves per SIMD, per block 18 MFMAs on 2 accumulator chains (two
// half-tiles of 32 items), ONE ds_read_b128 per MFMA (row stride 304 B), prefetch distance PF, optional drain (a VALU read of the
// accumulators) at the end of every block, optional extra LDS traffic from idle waves (EXTRA waves per SIMD that only read).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int HB = 9728;

template <int WPS, int PF, bool DRAIN, int RPM2>   // RPM2: B reads per 2 MFMAs (2 = one each, 1 = one per pair, 0 = none)
__global__ void __launch_bounds__(256 * WPS) k(unsigned* out, int iters, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 4 * HB / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = seed * i + 17;
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    u32x4 a[9];
    for (int m = 0; m < 9; ++m) a[m] = u32x4{seed + m, seed + 1, seed + 2 + threadIdx.x, seed + 3};
    f32x16 acc[2];
    const unsigned char* base = smem + j * 304 + 16 * h;
    constexpr int S = 18;
    u32x4 bq[PF];
    for (int s = 0; s < PF; ++s) bq[s] = *reinterpret_cast<const u32x4*>(base + (s & 1) * HB + 32 * (s >> 1));
    for (int it = 0; it < iters; ++it) {
        const unsigned char* tb = base + (it & 1) * 2 * HB;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int m = s >> 1, cb = s & 1;
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m]), __builtin_bit_cast(bf16x8, bq[s % PF]),
                                                              m == 0 ? (f32x16)(0.f) : acc[cb], 0, 0, 0);
            if (RPM2 == 2 || (RPM2 == 1 && (s & 1))) {
                const int s2 = (s + PF) % S;
                bq[s % PF] = *reinterpret_cast<const u32x4*>(tb + (s2 & 1) * HB + 32 * (s2 >> 1));
            }
        }
#if defined(__HIP_DEVICE_COMPILE__)
        if (DRAIN) { asm volatile("" ::"v"(acc[0])); asm volatile("" ::"v"(acc[1])); }
#endif
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(acc[0])); asm volatile("" ::"v"(acc[1]));
#endif
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = iters;
}
template <int WPS, int PF, bool DRAIN, int RPM2>
void run(unsigned* d) {
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<WPS, PF, DRAIN, RPM2>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<WPS, PF, DRAIN, RPM2>), dim3(256), dim3(256 * WPS), 40960, 0, d, iters, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<WPS, PF, DRAIN, RPM2>), dim3(256), dim3(256 * WPS), 40960, 0, d, iters, 12345u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)iters * 18 * WPS;      // per SIMD
    printf("waves/SIMD %d  prefetch %d  drain %d  B reads per 2 MFMAs %d: %6.2f ns per MFMA per SIMD  (%6.1f TFLOP/s chip)\n", WPS, PF, (int)DRAIN, RPM2,
           ms * 1e6 / mfmas, mfmas * 1024 * 32768.0 / (ms * 1e-3) / 1e12);
}
int main() {
    unsigned* d;
    hipMalloc(&d, 64);
    run<2, 8, true, 2>(d);
    run<2, 8, false, 2>(d);
    run<2, 8, true, 1>(d);
    run<2, 8, false, 1>(d);
    run<2, 8, true, 0>(d);
    run<2, 8, false, 0>(d);
    run<2, 4, true, 2>(d);
    run<3, 8, true, 2>(d);
    run<3, 8, false, 2>(d);
    run<4, 4, true, 2>(d);
    run<4, 4, false, 2>(d);
    run<1, 8, true, 2>(d);
    run<2, 2, true, 1>(d);
    run<2, 4, true, 1>(d);
    run<2, 6, true, 1>(d);
    run<2, 2, true, 2>(d);
    run<3, 2, true, 1>(d);
    run<3, 4, true, 1>(d);
    return 0;
}
