// Ranking metrics and the device-side triplet sampler (the steps right after / right before the hot path).
#include "pda_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Metrics: one thread per user row.  MF/used_metric.py:4-80, reduction of MF/train_new_api.py:741-778.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) metrics_kernel(const int32_t* topk, int n_rows, int k_cols,
                                                      const int64_t* tgt_indptr, const int32_t* tgt_indices,
                                                      const int32_t* Ks, int n_ks, double* sums) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t hitmask = 0;
    int npos = 0;
    if (r < n_rows) {
        const int64_t b = tgt_indptr[r], e = tgt_indptr[r + 1];
        npos = (int)(e - b);
        for (int k = 0; k < k_cols; ++k) {
            const int it = topk[(size_t)r * k_cols + k];
            bool h = false;
            for (int64_t p = b; p < e; ++p) h |= (tgt_indices[p] == it);   // np.isin(r, target)  used_metric.py:65-67
            if (h) hitmask |= 1ull << k;
        }
    }
    for (int q = 0; q < n_ks; ++q) {
        const int K = Ks[q];
        const int kk = K < k_cols ? K : k_cols;   // r[:K] on a k_cols-long vector
        double prec = 0, rec = 0, ndcg = 0, hit = 0;
        if (r < n_rows && npos > 0) {
            const uint64_t msk = kk >= 64 ? ~0ull : ((1ull << kk) - 1ull);
            const int hits = __popcll(hitmask & msk);
            double dcg = 0, idcg = 0;
            for (int k = 0; k < kk; ++k) {
                const double w = 1.0 / log2((double)k + 2.0);
                if ((hitmask >> k) & 1ull) dcg += w;
                if (k < npos) idcg += w;                                   // tp[:min(maxlen,k)]  :46-47
            }
            prec = (double)hits / kk;                                      // np.mean(r[:k])      :4-18
            rec = (double)hits / npos;                                     // :55-57
            ndcg = idcg > 0 ? dcg / idcg : 0.0;                            // :39-52
            hit = hits > 0 ? 1.0 : 0.0;                                    // :60-62
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            prec += __shfl_xor(prec, o, 64);
            rec += __shfl_xor(rec, o, 64);
            ndcg += __shfl_xor(ndcg, o, 64);
            hit += __shfl_xor(hit, o, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(sums + 0 * n_ks + q, prec);
            atomicAdd(sums + 1 * n_ks + q, rec);
            atomicAdd(sums + 2 * n_ks + q, ndcg);
            atomicAdd(sums + 3 * n_ks + q, hit);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Sampler.  Counter-based RNG (splitmix64 of (seed, step, row, draw)); a keyed Feistel permutation
// with cycle walking draws `B` distinct users per batch (rd.sample semantics) in O(B).
// ------------------------------------------------------------------------------------------------
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

__global__ void __launch_bounds__(256) sample_kernel(SampleArgs a) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
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
        a.users[r] = u;
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
    a.pos[r] = p;
    a.neg[r] = n;
    if (a.pop && a.pos_pop) {  // :402-403
        a.pos_pop[r] = a.pop[(size_t)p * a.n_slots + slot];
        a.neg_pop[r] = a.pop[(size_t)n * a.n_slots + slot];
    }
}

}  // namespace

extern "C" int pda_abi_version(void) { return PDA_ABI_VERSION; }

extern "C" const char* pda_error_string(int code) {
    switch (code) {
        case PDA_OK: return "ok";
        case PDA_ERR_ARG: return "invalid argument";
        case PDA_ERR_UNSUPPORTED: return "unsupported embed dim / K / mode";
        case PDA_ERR_LAUNCH: return "HIP launch failure";
        case PDA_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}

extern "C" int pda_metrics(const int32_t* topk, int n_rows, int k_cols, const int64_t* tgt_indptr,
                           const int32_t* tgt_indices, const int32_t* Ks, int n_ks, double* sums, void* stream) {
    if (!topk || !tgt_indptr || !tgt_indices || !Ks || !sums || n_rows <= 0 || n_ks <= 0) return PDA_ERR_ARG;
    if (k_cols < 1 || k_cols > 64) return PDA_ERR_ARG;
    hipLaunchKernelGGL(metrics_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), topk, n_rows, k_cols, tgt_indptr, tgt_indices, Ks, n_ks, sums);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

extern "C" int pda_sample_triplets(int32_t* users, int gen_users, const int32_t* user_pool, int n_pool, int B,
                                   const int64_t* train_indptr, const int32_t* train_indices,
                                   const int32_t* train_slots, int neg_lo, int neg_hi, const float* pop_matrix,
                                   int n_slots, uint64_t seed, uint64_t step, int32_t* pos, int32_t* neg,
                                   float* pos_pop, float* neg_pop, void* stream) {
    if (!users || !train_indptr || !train_indices || !pos || !neg || B <= 0 || neg_hi <= neg_lo) return PDA_ERR_ARG;
    if (gen_users && n_pool <= 0) return PDA_ERR_ARG;
    if (pop_matrix && (!pos_pop || !neg_pop || n_slots <= 0)) return PDA_ERR_ARG;
    SampleArgs a{users, user_pool, train_indptr, train_indices, train_slots, pop_matrix, pos, neg, pos_pop, neg_pop,
                 seed, step, B, n_pool, gen_users, neg_lo, neg_hi, n_slots, nullptr, nullptr};
    hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

namespace {
__global__ void counter_add_kernel(uint64_t* c, uint64_t inc) { *c += inc; }
}  // namespace

extern "C" int pda_sample_triplets_dev(int32_t* users, int gen_users, const int32_t* user_pool, int n_pool, int B,
                                       const int64_t* train_indptr, const int32_t* train_indices,
                                       const int32_t* train_slots, int neg_lo, int neg_hi, const float* pop_matrix,
                                       int n_slots, uint64_t seed, const uint64_t* step_dev, uint64_t* step_next, int32_t* pos,
                                       int32_t* neg, float* pos_pop, float* neg_pop, void* stream) {
    if (!users || !train_indptr || !train_indices || !pos || !neg || !step_dev || B <= 0 || neg_hi <= neg_lo) return PDA_ERR_ARG;
    if (step_next == step_dev) return PDA_ERR_ARG;      // other workgroups may still be reading *step_dev
    if (gen_users && n_pool <= 0) return PDA_ERR_ARG;
    if (pop_matrix && (!pos_pop || !neg_pop || n_slots <= 0)) return PDA_ERR_ARG;
    SampleArgs a{users, user_pool, train_indptr, train_indices, train_slots, pop_matrix, pos, neg, pos_pop, neg_pop,
                 seed, 0, B, n_pool, gen_users, neg_lo, neg_hi, n_slots, step_dev, step_next};
    hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

extern "C" int pda_counter_add(uint64_t* counter, uint64_t inc, void* stream) {
    if (!counter) return PDA_ERR_ARG;
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), counter, inc);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}
