// Device-side BPR triplet sampler (N1, MF/train_new_api.py:366-412): shared by the stand-alone sampler kernel
// (pda_aux.hip) and by the train step that draws the NEXT batch in spare workgroups of its own launch (pda_bpr_step.hip).
#pragma once
#include "pda_common.h"

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t draw(uint64_t seed, uint64_t step, uint32_t row, uint32_t k) {
    return (uint32_t)(mix64(mix64(seed ^ (step * 0xD1B54A32D192ED03ull)) ^ (((uint64_t)row << 32) | k)) >> 32);
}
__device__ __forceinline__ uint32_t bounded(uint32_t r, uint32_t n) { return (uint32_t)(((uint64_t)r * n) >> 32); }

__device__ uint32_t feistel_perm(uint32_t x, uint32_t n, uint64_t key) {
    int bits = 1;
    while ((1ull << bits) < n) ++bits;
    const int hb = (bits + 1) / 2;
    const uint32_t hm = (1u << hb) - 1u;
    do {
        uint32_t l = x >> hb, r = x & hm;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            const uint32_t f = (uint32_t)mix64(key ^ ((uint64_t)round << 40) ^ r) & hm;
            const uint32_t nl = r;
            r = l ^ f;
            l = nl;
        }
        x = (l << hb) | r;
    } while (x >= n);
    return x;
}

struct SampleArgs {
    int32_t* users;
    const int32_t* user_pool;
    const int64_t* indptr;
    const int32_t* indices;
    const int32_t* slots;
    const float* pop;
    int32_t* pos;
    int32_t* neg;
    float* pos_pop;
    float* neg_pop;
    uint64_t seed, step;
    int B, n_pool, gen_users, neg_lo, neg_hi, n_slots;
    const uint64_t* step_dev;   // optional device-resident step counter added to `step` (HIP-graph replay: pda_counter_add)
    uint64_t* step_next;        // optional: receives *step_dev + 1 (a DIFFERENT location: no launch in between needed)
};

// COH: the batch is handed to other workgroups of the SAME launch (pda_bpr_train_steps_f32): its five words per triplet leave
// as device-scope stores (no cache write-back needed before the grid barrier)
template <bool COH, typename T>
__device__ __forceinline__ void out_store(T* p, T v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// one thread = one triplet r of the batch
template <bool COH = false>
__device__ __forceinline__ void sample_one(SampleArgs a, int r) {
    if (r >= a.B) return;
    if (a.step_dev) {
        const uint64_t cur = *a.step_dev;
        a.step += cur;
        if (a.step_next && r == 0) *a.step_next = cur + 1;
    }
    int u;
    if (a.gen_users) {
        const uint64_t key = mix64(a.seed ^ mix64(a.step));
        // B <= n_pool: distinct users (rd.sample, MF/train_new_api.py:380-381); else with replacement (:383)
        const uint32_t x = a.B <= a.n_pool ? feistel_perm((uint32_t)r, (uint32_t)a.n_pool, key)
                                           : bounded(draw(a.seed, a.step, r, 7), a.n_pool);
        u = a.user_pool ? a.user_pool[x] : (int)x;
        out_store<COH>(&a.users[r], u);
    } else {
        u = a.users[r];
    }
    const int64_t b = a.indptr[u], e = a.indptr[u + 1];
    const int len = (int)(e - b);
    int p = 0, slot = 0;
    if (len == 0) {  // :387-390
        p = 0;
        slot = a.n_slots > 0 ? (int)bounded(draw(a.seed, a.step, r, 1), a.n_slots) : 0;
    } else {         // :392-396
        const int idx = (int)bounded(draw(a.seed, a.step, r, 0), len);
        p = a.indices[b + idx];
        slot = a.slots ? a.slots[b + idx] : 0;
    }
    int n = a.neg_lo;
    const uint32_t span = (uint32_t)(a.neg_hi - a.neg_lo);
    for (uint32_t k = 0; k < 4096; ++k) {  // rejection against the (sorted) train row, :397-401
        n = a.neg_lo + (int)bounded(draw(a.seed, a.step, r, 16 + k), span);
        int64_t lo = b, hi = e;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (a.indices[mid] < n) lo = mid + 1; else hi = mid;
        }
        if (!(lo < e && a.indices[lo] == n)) break;
    }
    out_store<COH>(&a.pos[r], p);
    out_store<COH>(&a.neg[r], n);
    if (a.pop && a.pos_pop) {  // :402-403
        out_store<COH>(&a.pos_pop[r], a.pop[(size_t)p * a.n_slots + slot]);
        out_store<COH>(&a.neg_pop[r], a.pop[(size_t)n * a.n_slots + slot]);
    }
}

}  // namespace
