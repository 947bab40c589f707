// micro-benchmark 6 (round 4): machine mappings of the sweep with the TRANSPOSED product (A = item fragment from the LDS, B = user
// fragment in registers; no folded test k-step; per-lane threshold compare) and UA = 2 / 4 / 8 user fragments per item fragment.
// The main loop of an MFMA wave is ONE inline-asm statement with hard registers (tools/ubench/gen_loop5.py -> loop5.h); tiles are
// streamed into the LDS by LDS-DMA -- by loader waves (V1, V2) or by the MFMA waves themselves (V3, V4); d = 128 (V4d64: d = 64).
//   V1  8 MFMA waves x 64 users, two per SIMD (168 VGPRs; 512 users per workgroup) + 2 loaders + 2 idle   = today's wide geometry
//   V2  4 MFMA waves x 128 users, one per SIMD (256 VGPRs; 512 users) + 2 loaders + 2 idle
//   V3  8 MFMA waves x 128 users, two per SIMD (256 VGPRs; 1024 users), loading their own tiles
//   V4  4 MFMA waves x 256 users, one per SIMD (512 registers, user fragments in AGPRs; 1024 users), loading their own tiles
// build: python tools/ubench/gen_loop5.py > tools/ubench/loop5.h && hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_struct5 tools/ubench/mfma_struct5.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "loop5.h"

template <class LP, int MW, int LW, int IW>
__global__ void __launch_bounds__(64 * (MW + LW + IW)) k5(const unsigned char* __restrict__ rows, size_t n_bytes, unsigned* out, unsigned n_body, unsigned seed, float thr0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int UB = LP::UA * LP::NK * 1024;                     // user fragments staged for the prologue (every MFMA wave reads the same ones)
    constexpr int NSLOT = 8;
    constexpr int RING = NSLOT * LP::HB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (UB + RING + 1024) / 4; i += blockDim.x) {
        unsigned x = (seed + 977u * i) * 2654435761u;
        x = x * 1664525u + 1013904223u;
        reinterpret_cast<unsigned*>(smem)[i] = (x & 0x807f807fu) | 0x3c003c80u;       // two bf16 of magnitude ~1 with random signs and mantissas
    }
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned tiles0 = lds0 + UB, sync0 = tiles0 + RING;
    if (wave < MW) {
        const int j = lane & 31, h = lane >> 5;
        float sink = 0.f;
        unsigned long long flag = 0;
        const size_t tile0 = (((size_t)blockIdx.x * 977) % 1000) * LP::HB;
        LP::run(sink, flag, lds0 + (unsigned)lane * 16u, tiles0 + (unsigned)(j * LP::RB + 16 * h), tiles0 + RING, 16u * (unsigned)h, sync0, thr0, n_body,
                rows + tile0, (unsigned)(lane * 16 + wave * 1024), tiles0 + 2 * LP::HB + (unsigned)wave * 1024u);
        if (sink == 1.2345f || flag == 0x1234567ull) out[1 + wave] = (unsigned)flag;
    } else if (wave < MW + LW) {
        // loaders: 17 pieces of 1 KiB per body (two half-tiles) over LW waves, two bodies in flight, no hand-over
        const int l = wave - MW;
        constexpr int NP = (2 * LP::HB + 1023) / 1024, MYP = LW > 0 ? (NP + (LW > 0 ? LW : 1) - 1) / (LW > 0 ? LW : 1) : 0;
        const size_t tile0 = (((size_t)blockIdx.x * 977) % 1000) * LP::HB;
        for (unsigned b = 0; b < n_body; ++b) {
            const unsigned char* src = rows + tile0 + (size_t)b * 2 * LP::HB + lane * 16;
            const unsigned dst = tiles0 + (unsigned)((b & 3) * 2 * LP::HB);
#pragma unroll
            for (int c = 0; c < MYP; ++c) {
                const int piece = l + LW * c;
                if (piece < NP) {
#if defined(__HIP_DEVICE_COMPILE__)
                    unsigned keep;
                    const unsigned char* gsrc = src + (size_t)piece * 1024;
                    const unsigned ldst = __builtin_amdgcn_readfirstlane(dst + (unsigned)piece * 1024u);
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(gsrc), "s"(ldst) : "memory");
#endif
                }
            }
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MYP) : "memory");
            // pace the loaders to the MFMA waves (they would otherwise run ahead): a body is 2 NK UA x 32 cycles of MFMAs per wave
            __builtin_amdgcn_s_sleep(10);
#endif
        }
    } else {
        for (unsigned b = 0; b < n_body; ++b) __builtin_amdgcn_s_sleep(16);
    }
    if (tid == 0 && blockIdx.x == 0) out[0] = n_body;
}

template <class LP, int MW, int LW, int IW>
void run(const unsigned char* rows, size_t n_bytes, unsigned* d, const char* what, int grid, unsigned n_body) {
    const size_t lds = 156 * 1024;                      // one workgroup per CU, like the sweep
    auto fn = k5<LP, MW, LW, IW>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(fn));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * (MW + LW + IW)), lds, 0, rows, n_bytes, d, n_body, 12345u, 1e30f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * (MW + LW + IW)), lds, 0, rows, n_bytes, d, n_body, 12345u, 1e30f);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", what); return; }
    const double mf = (double)grid * n_body * MW * LP::MFMA_PER_BODY;
    printf("%-96s regs %3d  %7.3f ms  %7.1f TFLOP/s (all algorithmic: no test k-step) = %.3f of 2.5 PF\n", what, fa.numRegs, best,
           mf * 32768.0 / (best * 1e-3) / 1e12, mf * 32768.0 / (best * 1e-3) / 1e12 / 2500.0);
}

int main() {
    const size_t n_bytes = (size_t)64 << 20;
    unsigned char* rows;
    unsigned* d;
    hipMalloc(&rows, n_bytes);
    {   // random bf16 values of realistic magnitude (the chip is power-limited: constant operands clock higher and prove nothing)
        unsigned short* h = (unsigned short*)malloc(n_bytes);
        unsigned long long x = 88172645463325252ull;
        for (size_t i = 0; i < n_bytes / 2; ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            h[i] = (unsigned short)(((x >> 20) & 0x807f) | (((unsigned)(0x3b + ((x >> 40) & 3))) << 7));     // sign, 7 mantissa bits, exponent 0x3b .. 0x3e
        }
        if (getenv("UB_CONST")) for (size_t i = 0; i < n_bytes / 2; ++i) h[i] = 0x3c00;
        hipMemcpy(rows, h, n_bytes, hipMemcpyHostToDevice);
        free(h);
    }
    hipMalloc(&d, 256);
    // equal work per workgroup-second: a body is 2 half-tiles against the workgroup's users
    run<LoopV1, 8, 2, 2>(rows, n_bytes, d, "V1  8 MFMA waves x  64 users (2 per SIMD, 168 VGPRs), 2 loaders + 2 idle, tests behind a drain", 1024, 1500);
    run<LoopV2, 4, 2, 2>(rows, n_bytes, d, "V2  4 MFMA waves x 128 users (1 per SIMD, 256 VGPRs), 2 loaders + 2 idle, tests in the MFMA shadow", 1024, 1500);
    run<LoopV3, 8, 0, 0>(rows, n_bytes, d, "V3  8 MFMA waves x 128 users (2 per SIMD, 256 VGPRs), self-loading", 512, 1500);
    run<LoopV4, 4, 0, 0>(rows, n_bytes, d, "V4  4 MFMA waves x 256 users (1 per SIMD, 512 registers), self-loading, 2 skew groups", 512, 1500);
    run<LoopV4g1, 4, 0, 0>(rows, n_bytes, d, "V4' the same, one group (all eight tests at the half-tile boundary)", 512, 1500);
    run<LoopV4d64, 4, 0, 0>(rows, n_bytes, d, "V4  d = 64: 4 MFMA waves x 256 users, self-loading, 2 skew groups", 512, 3000);
    return 0;
}
