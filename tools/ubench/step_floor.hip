// micro-benchmark 7 (round 3): where do the ~7 us of a 2 048-triplet step kernel go?  64 kernel nodes per HIP graph (the bench's
// shape), grid = 64 workgroups x 512 threads (d = 64: 16 lanes per triplet), C2-sized tables.
//   0 empty kernel   1 + the three index loads   2 + the three row gathers and the dots   3 + plain store of the user row
//   4 + atomics on the two item rows (= the fused hogwild step without its LDS combining and loss reduction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(512) k(float* U, float* I, const int* users, const int* pos, const int* neg, int B, float* sink) {
    if (MODE == 0) return;
    const int t = blockIdx.x * 32 + threadIdx.x / 16, e = threadIdx.x % 16;
    if (t >= B) return;
    const int u = users[t], p = pos[t], n = neg[t];
    if (MODE == 1) { if (u + p + n == -12345) sink[0] = 1.f; return; }
    const f32x4 ue = *reinterpret_cast<const f32x4*>(U + (size_t)u * 64 + 4 * e);
    const f32x4 pe = *reinterpret_cast<const f32x4*>(I + (size_t)p * 64 + 4 * e);
    const f32x4 ne = *reinterpret_cast<const f32x4*>(I + (size_t)n * 64 + 4 * e);
    float ps = ue[0] * pe[0] + ue[1] * pe[1] + ue[2] * pe[2] + ue[3] * pe[3], ns = ue[0] * ne[0] + ue[1] * ne[1] + ue[2] * ne[2] + ue[3] * ne[3];
    for (int o = 8; o > 0; o >>= 1) { ps += __shfl_xor(ps, o, 64); ns += __shfl_xor(ns, o, 64); }
    const float g = 1e-6f / (1.f + __expf(ps - ns));
    if (MODE == 2) { if (g == 12345.f) sink[0] = g; return; }
    *reinterpret_cast<f32x4*>(U + (size_t)u * 64 + 4 * e) = ue - (pe - ne) * g;
    if (MODE == 3) return;
    for (int q = 0; q < 4; ++q) {
        unsafeAtomicAdd(I + (size_t)p * 64 + 4 * e + q, -g * ue[q]);
        unsafeAtomicAdd(I + (size_t)n * 64 + 4 * e + q, g * ue[q]);
    }
}
template <int MODE>
float run(float* U, float* I, const int* users, const int* pos, const int* neg, int B, float* sink) {
    hipStream_t s; (void)hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 64; ++i) hipLaunchKernelGGL(k<MODE>, dim3((B + 31) / 32), dim3(512), 0, s, U, I, users + (size_t)(i % 16) * B, pos + (size_t)(i % 16) * B, neg + (size_t)(i % 16) * B, B, sink);
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int r = 0; r < 32; ++r) (void)hipGraphLaunch(ge, s);
    (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / (32 * 64);
}
int main() {
    const int nU = 50000, nI = 20000;
    for (int B : {2048, 4096, 16384}) {
        float *U, *I, *sink; int *users, *pos, *neg;
        (void)hipMalloc(&U, (size_t)nU * 64 * 4); (void)hipMalloc(&I, (size_t)nI * 64 * 4); (void)hipMalloc(&sink, 64);
        (void)hipMemset(U, 0, (size_t)nU * 64 * 4); (void)hipMemset(I, 0, (size_t)nI * 64 * 4);
        std::vector<int> hu(16 * B), hp(16 * B), hn(16 * B);
        unsigned x = 12345;
        for (int b = 0; b < 16; ++b)
            for (int i = 0; i < B; ++i) {
                x = x * 1664525u + 1013904223u; hu[b * B + i] = (int)((i * 24 + (x >> 8) % 24) % nU);     // distinct inside a batch (B x 24 <= nU for B = 2048)
                x = x * 1664525u + 1013904223u; hp[b * B + i] = (int)((x >> 8) % nI);
                x = x * 1664525u + 1013904223u; hn[b * B + i] = (int)((x >> 8) % nI);
            }
        (void)hipMalloc(&users, hu.size() * 4); (void)hipMalloc(&pos, hp.size() * 4); (void)hipMalloc(&neg, hn.size() * 4);
        (void)hipMemcpy(users, hu.data(), hu.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(pos, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(neg, hn.data(), hn.size() * 4, hipMemcpyHostToDevice);
        printf("B = %5d: empty %.2f  + indices %.2f  + gathers, dots %.2f  + user-row store %.2f  + item-row atomics %.2f  us per kernel node\n", B,
               run<0>(U, I, users, pos, neg, B, sink), run<1>(U, I, users, pos, neg, B, sink), run<2>(U, I, users, pos, neg, B, sink),
               run<3>(U, I, users, pos, neg, B, sink), run<4>(U, I, users, pos, neg, B, sink));
    }
    return 0;
}
