// micro-benchmark: what does a VALU read of MFMA results cost?  One wave per SIMD (256 threads per CU), 36 MFMAs per "tile"
// on 4 accumulators, then NV v_or3 reading (a) the accumulators, (b) other registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV>
__global__ void __launch_bounds__(256) k(unsigned* out, int iters, unsigned seed) {
    u32x4 a0 = {seed, seed + 1, seed + 2, seed + 3}, b0 = {seed ^ 5, seed ^ 6, seed ^ 7, seed ^ 8};
    a0[0] += threadIdx.x;
    u32x4 aq[4];
    for (int q = 0; q < 4; ++q) { aq[q] = a0; aq[q][2] += 77u * q; }
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    unsigned other[32];
    for (int r = 0; r < 32; ++r) other[r] = seed * (r + 3) + threadIdx.x;
    unsigned sink = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 9; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aq[q]), __builtin_bit_cast(bf16x8, b0), m == 0 ? (f32x16)(0.f) : acc[q], 0, 0, 0);
        unsigned m0 = 0;
        if constexpr (MODE == 1) {          // OR tree over the accumulators
#pragma unroll
            for (int v = 0; v < NV; ++v) m0 |= __float_as_uint(acc[(2 * v / 16) & 3][(2 * v) & 15]) | __float_as_uint(acc[(2 * v / 16) & 3][(2 * v + 1) & 15]);
        } else if constexpr (MODE == 2) {   // OR tree over other registers
#pragma unroll
            for (int v = 0; v < NV; ++v) m0 |= other[(2 * v) & 31] | other[(2 * v + 1) & 31];
            asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
        } else if constexpr (MODE == 3) {   // one dependent read, then OR tree over other registers
            m0 = __float_as_uint(acc[3][15]);
#pragma unroll
            for (int v = 0; v < NV; ++v) m0 |= other[(2 * v) & 31] | other[(2 * v + 1) & 31];
            asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
        } else if constexpr (MODE == 4) {   // OR tree over the accumulators of the PREVIOUS iteration's copy?  (v_mov copies first)
            unsigned cp[32];
#pragma unroll
            for (int v = 0; v < NV * 2 && v < 32; ++v) cp[v] = __float_as_uint(acc[(v / 16) & 3][v & 15]);
#pragma unroll
            for (int v = 0; v < NV; ++v) m0 |= cp[(2 * v) & 31] | cp[(2 * v + 1) & 31];
        } else if constexpr (MODE == 5) {   // 3 fresh accumulator registers per v_or3, partial results combined afterwards
            unsigned p[22];
#pragma unroll
            for (int v = 0; v < 21; ++v) p[v] = __float_as_uint(acc[(3 * v / 16) & 3][(3 * v) & 15]) | __float_as_uint(acc[((3 * v + 1) / 16) & 3][(3 * v + 1) & 15]) | __float_as_uint(acc[((3 * v + 2) / 16) & 3][(3 * v + 2) & 15]);
            p[21] = __float_as_uint(acc[3][15]);
#pragma unroll
            for (int v = 0; v < 22; ++v) m0 |= p[v];
        } else if constexpr (MODE == 6) {   // read every accumulator register twice
#pragma unroll
            for (int v = 0; v < NV; ++v) m0 |= __float_as_uint(acc[(2 * v / 16) & 3][(2 * v) & 15]) | __float_as_uint(acc[(2 * v / 16) & 3][(2 * v + 1) & 15]);
            unsigned m1 = 0;
#pragma unroll
            for (int v = 0; v < NV; ++v) m1 += __float_as_uint(acc[(2 * v / 16) & 3][(2 * v) & 15]) ^ __float_as_uint(acc[(2 * v / 16) & 3][(2 * v + 1) & 15]);
            m0 |= m1 << 31;
        } else if constexpr (MODE == 7) {   // s_nop padding between the MFMAs and the reads
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
            for (int v = 0; v < NV; ++v) m0 |= __float_as_uint(acc[(2 * v / 16) & 3][(2 * v) & 15]) | __float_as_uint(acc[(2 * v / 16) & 3][(2 * v + 1) & 15]);
        } else if constexpr (MODE == 8) {   // v_cmp per register (sign test), results OR-ed on the scalar side
            unsigned long long mm = 0;
#pragma unroll
            for (int v = 0; v < NV * 2; ++v) mm |= __builtin_amdgcn_ballot_w64(__float_as_int(acc[(v / 16) & 3][v & 15]) < 0);
            m0 = mm != 0 ? 0x80000000u : 0u;
        } else if constexpr (MODE == 9) {   // v_min3_f32 tree
            float f = 1.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) f = __builtin_fminf(__builtin_fminf(f, acc[(2 * v / 16) & 3][(2 * v) & 15]), acc[(2 * v / 16) & 3][(2 * v + 1) & 15]);
            m0 = __float_as_uint(f);
        } else {
            asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
        }
        if (__builtin_amdgcn_ballot_w64((int)m0 < 0 && MODE != 0) == ~0ull) sink += 1;   // wave-uniform, data dependent
#pragma unroll
        for (int q = 0; q < 4; ++q) aq[q][1] ^= sink;
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (unsigned)(t1 - t0); out[1] = sink; }
}

template <int MODE, int NV>
void run(unsigned* d, const char* name) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(256), 0, 0, d, iters, 12345u);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(256), 0, 0, d, iters, 12345u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned h[2];
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("%-44s NV=%2d: %7.1f ticks per tile, %7.1f ns per tile (36 MFMAs of 32 cycles at 2.4 GHz = 480 ns)\n", name, NV, (double)h[0] / iters, ms * 1e6 / iters);
}
int main() {
    unsigned* d;
    hipMalloc(&d, 64);
    run<0, 0>(d, "MFMAs only");
    run<3, 0>(d, "one dependent read");
    run<1, 8>(d, "v_or3 over accumulators");
    run<1, 16>(d, "v_or3 over accumulators");
    run<1, 32>(d, "v_or3 over accumulators");
    run<2, 32>(d, "v_or3 over other registers (no sync)");
    run<3, 32>(d, "one dependent read + v_or3 over others");
    run<4, 16>(d, "v_mov copies, then v_or3 over the copies");
    run<5, 32>(d, "v_or3 of 3 fresh accumulators, then combine");
    run<6, 32>(d, "every accumulator register read twice");
    run<7, 32>(d, "64 wait states, then v_or3 over accumulators");
    run<8, 32>(d, "v_cmp per accumulator register");
    run<9, 32>(d, "v_min3_f32 over accumulators");
    return 0;
}
