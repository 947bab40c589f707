// micro-benchmark 7 (round 4): pure MFMA streams in the operand configurations of the huge geometry (tools/ubench/gen_pure6.py), one wave
// per SIMD with 512 registers, random bf16 operands (the chip is power-limited: what does each configuration deliver?)
// build: python tools/ubench/gen_pure6.py > tools/ubench/pure6.h && hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_pure6 tools/ubench/mfma_pure6.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "pure6.h"

template <class P>
__global__ void __launch_bounds__(256) k6(const unsigned char* src, unsigned n, float* out) {
    float sink = 0.f;
    const unsigned lane16 = (threadIdx.x & 63) * 16u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    P::run(src + (size_t)((blockIdx.x * 4 + wave) % 61) * 65536, n, lane16, sink);
    if (sink == 1.2345f) out[0] = sink;
}

template <class P>
void run(const unsigned char* src, float* out, const char* what, unsigned n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k6<P>, dim3(256), dim3(256), 0, 0, src, n, out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k6<P>, dim3(256), dim3(256), 0, 0, src, n, out);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double tf = 2.0 * P::kMacs * n * 1024.0 / (best * 1e-3) / 1e12;
    printf("%-110s %7.3f ms  %7.1f TFLOP/s = %.3f of 2.5 PF\n", what, best, tf, tf / 2500.0);
}

int main() {
    const size_t n_bytes = (size_t)8 << 20;
    unsigned char* src; float* out;
    hipMalloc(&src, n_bytes); hipMalloc(&out, 256);
    unsigned short* h = (unsigned short*)malloc(n_bytes);
    unsigned long long x = 88172645463325252ull;
    for (size_t i = 0; i < n_bytes / 2; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        h[i] = (unsigned short)(((x >> 20) & 0x807f) | (((unsigned)(0x3b + ((x >> 40) & 3))) << 7));
    }
    hipMemcpy(src, h, n_bytes, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<P1>(src, out, "P1 32x32x16: A VGPR (x4 in a row), B AGPR, 8 accumulators in VGPRs (sweep5's stream)", 6000);
        run<P2>(src, out, "P2 32x32x16: A VGPR, B VGPR, accumulators in VGPRs", 6000);
        run<P3>(src, out, "P3 16x16x32: A VGPR (x8 in a row), B AGPR, 16 accumulators of 4 VGPRs", 6000);
        run<P4>(src, out, "P4 32x32x16: A VGPR, B VGPR, accumulators in AGPRs", 6000);
    }
    return 0;
}
