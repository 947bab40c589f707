// micro-benchmark 2: the kernel's structure -- two accumulator sets (A, B) of 2 chains x 9 MFMAs per half-tile; the OR tree over
// set A is issued after the MFMAs of set B and vice versa.  One wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned* out, int iters, unsigned seed) {
    u32x4 a[2][9], b[9];
    for (int m = 0; m < 9; ++m) {
        for (int u = 0; u < 2; ++u) a[u][m] = u32x4{seed + m, seed + 1 + u, seed + 2 + threadIdx.x, seed + 3};
        b[m] = u32x4{seed ^ (5 + m), seed ^ 6, seed ^ 7, seed ^ 8};
    }
    f32x16 accA[2], accB[2];
    for (int u = 0; u < 2; ++u) for (int r = 0; r < 16; ++r) { accA[u][r] = 0.f; accB[u][r] = 0.f; }
    unsigned sink = 0;
    auto block = [&](f32x16 (&acc)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 9; ++m)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[u][m]), __builtin_bit_cast(bf16x8, b[m]), m == 0 ? (f32x16)(0.f) : acc[u], 0, 0, 0);
    };
    auto ortree = [&](const f32x16 (&acc)[2]) __attribute__((always_inline)) -> unsigned {
        unsigned m0 = 0;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; r += 2) m0 |= __float_as_uint(acc[u][r]) | __float_as_uint(acc[u][r + 1]);
        return m0;
    };
    long long t0 = __builtin_readcyclecounter();
    block(accA);
    for (int it = 0; it < iters; ++it) {
        block(accB);
        __builtin_amdgcn_sched_barrier(0);
        unsigned m0 = 0;
        if constexpr (MODE == 1) m0 = ortree(accA);
        if constexpr (MODE == 2) m0 = __float_as_uint(accA[1][15]);
        if constexpr (MODE == 0) asm volatile("" ::"v"(accA[0]), "v"(accA[1]));
        if (__builtin_amdgcn_ballot_w64((int)m0 < 0 && MODE != 0) == ~0ull) sink += 1;
        __builtin_amdgcn_sched_barrier(0);
        block(accA);
        __builtin_amdgcn_sched_barrier(0);
        unsigned m1 = 0;
        if constexpr (MODE == 1) m1 = ortree(accB);
        if constexpr (MODE == 2) m1 = __float_as_uint(accB[1][15]);
        if constexpr (MODE == 0) asm volatile("" ::"v"(accB[0]), "v"(accB[1]));
        if (__builtin_amdgcn_ballot_w64((int)m1 < 0 && MODE != 0) == ~0ull) sink += 1;
        a[0][0][1] ^= sink;
        a[1][0][1] ^= sink;
    }
    long long t1 = __builtin_readcyclecounter();
    asm volatile("" ::"v"(accA[0]), "v"(accA[1]), "v"(accB[0]), "v"(accB[1]));
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (unsigned)(t1 - t0); out[1] = sink; }
}
template <int MODE>
void run(unsigned* d, const char* name) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, d, iters, 12345u);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, d, iters, 12345u);
    hipDeviceSynchronize();
    unsigned h[2];
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("%-50s %7.1f cycles per 36 MFMAs (1152 at full rate)\n", name, (double)h[0] / iters);
}
int main() {
    unsigned* d;
    hipMalloc(&d, 64);
    run<0>(d, "MFMAs only");
    run<2>(d, "one read of the other set behind each block");
    run<1>(d, "OR tree over the other set behind each block");
    return 0;
}
