// micro-benchmark 6 (round 3): what does the LEGACY v_mfma_f32_32x32x8_bf16_1k (k = 8) cost on gfx950 next to v_mfma_f32_32x32x16_bf16
// (k = 16)?  If half, the folded threshold test (one extra k-step per 64-item block, 1/9 of all MFMAs at d = 128) could run as a
// k = 8 step with a 2-piece split of thr and 1 / pop.  Register operands, 2 waves per SIMD x 4 chains, random mantissas.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: 8 x k16;  1: 8 x k8;  2: 8 x k16 + 1 x k16 (today's block shape);  3: 8 x k16 + 1 x k8;  4: 8 x fp8 k16;  5: 8 x k16 + 1 x fp8 k16
__global__ void __launch_bounds__(512) k(unsigned* out, int iters, unsigned seed) {
    const unsigned t = threadIdx.x * 2654435761u + seed;
    u32x4 a16 = {(t & 0x807f807fu) | 0x3c003c80u, ((t * 3) & 0x807f807fu) | 0x3c003c80u, ((t * 5) & 0x807f807fu) | 0x3c003c80u, ((t * 7) & 0x807f807fu) | 0x3c003c80u};
    u32x4 b16 = {((t * 11) & 0x807f807fu) | 0x3c003c80u, ((t * 13) & 0x807f807fu) | 0x3c003c80u, ((t * 17) & 0x807f807fu) | 0x3c003c80u, ((t * 19) & 0x807f807fu) | 0x3c003c80u};
    u32x2 a8 = {a16[0], a16[1]}, b8 = {b16[0], b16[1]};
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) acc[c] = (f32x16)(0.f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (MODE == 4) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(__builtin_bit_cast(long, a8), __builtin_bit_cast(long, b8), acc[c], 0, 0, 0);
                else if (MODE == 1) acc[c] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, a8), __builtin_bit_cast(s16x4, b8), acc[c], 0, 0, 0);
                else acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a16), __builtin_bit_cast(bf16x8, b16), acc[c], 0, 0, 0);
            }
            if (MODE == 2) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b16), __builtin_bit_cast(bf16x8, a16), acc[c], 0, 0, 0);
            if (MODE == 5) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(__builtin_bit_cast(long, b8), __builtin_bit_cast(long, a8), acc[c], 0, 0, 0);
            if (MODE == 3) acc[c] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, b8), __builtin_bit_cast(s16x4, a8), acc[c], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 1.2345f) out[0] = 1;
}
template <int MODE>
void run(unsigned* d, const char* what, double mfma_per_iter_k16, double mfma_per_iter_k8) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters, 1u);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters, 1u);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per SIMD: 2 waves
    const double n16 = 2.0 * iters * mfma_per_iter_k16, n8 = 2.0 * iters * mfma_per_iter_k8;
    printf("%-44s %7.3f ms  ns per SIMD-iteration %8.2f  (k16: %.0f  k8: %.0f MFMAs per SIMD)  %7.1f TFLOP/s\n", what, best, best * 1e6 / (2.0 * iters), n16, n8,
           (n16 * 32768.0 + n8 * 16384.0) * 1024 / (best * 1e-3) / 1e12);
}
int main() {
    unsigned* d;
    (void)hipMalloc(&d, 64);
    run<0>(d, "32 x v_mfma_f32_32x32x16_bf16", 32, 0);
    run<1>(d, "32 x v_mfma_f32_32x32x8_bf16_1k", 0, 32);
    run<2>(d, "32 x k16 + 4 x k16 (block shape today)", 36, 0);
    run<3>(d, "32 x k16 + 4 x k8  (half-size test step)", 32, 4);
    run<4>(d, "32 x v_mfma_f32_32x32x16_fp8_fp8", 32, 0);
    run<5>(d, "32 x k16 + 4 x fp8 k16", 36, 0);
    run<0>(d, "32 x v_mfma_f32_32x32x16_bf16 (again)", 32, 0);
    return 0;
}
