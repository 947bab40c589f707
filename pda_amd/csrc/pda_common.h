// Shared device helpers for libpda_hip.so (gfx950 only -- wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pda_hip_experimental.h" // (includes pda_hip.h: the stable surface)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PDA_CHECK_LAUNCH()                                  \
    do {                                                    \
        if (hipGetLastError() != hipSuccess) return PDA_ERR_LAUNCH; \
    } while (0)

// Monotone float -> uint32 map (larger float => larger uint); -0.0 is canonicalised by the caller.
__device__ __forceinline__ uint32_t pda_ordf(float v) {
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float pda_unordf(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}
// Packed candidate key: larger = better; ties on the score -> lower item id wins (tf.nn.top_k rule).
__device__ __forceinline__ uint64_t pda_pack_key(float v, uint32_t item) {
    return ((uint64_t)pda_ordf(v + 0.0f) << 32) | (uint64_t)(0xFFFFFFFFu - item);
}
__device__ __forceinline__ float pda_key_val(uint64_t k) { return pda_unordf((uint32_t)(k >> 32)); }
__device__ __forceinline__ int32_t pda_key_item(uint64_t k) { return (int32_t)(0xFFFFFFFFu - (uint32_t)k); }

__device__ __forceinline__ uint64_t pda_readlane_u64(uint64_t v, int src) {
    uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, src);
    uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ float pda_readlane_f32(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
// Four consecutive table elements starting at element index `idx` (a multiple of 4) as fp32: f32 tables -> one 16-byte
// load; bf16 tables (BF) -> one 8-byte load, widened exactly (bf16 is the top half of an fp32).
template <bool BF>
__device__ __forceinline__ f32x4 pda_load4(const void* base, size_t idx) {
    if constexpr (BF) {
        const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + idx);
        f32x4 v = {__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xFFFF0000u), __uint_as_float(w.y << 16),
                   __uint_as_float(w.y & 0xFFFF0000u)};
        return v;
    } else {
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
    }
}

// Orders LDS traffic between lanes of ONE wave (no instruction: compiler-level only; the LDS
// pipeline already executes a wave's DS ops in order).
__device__ __forceinline__ void pda_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
