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
}  // namespace
#include "pda_sample.h"
namespace {

// one wave per workgroup: every thread is a chain of dependent loads, spreading the waves over CUs beats packing them (13 -> 9 us)
__global__ void __launch_bounds__(64) sample_kernel(SampleArgs a) { sample_one(a, (int)(blockIdx.x * blockDim.x + threadIdx.x)); }

// n batches in one launch (blockIdx.y = batch j, drawn with step + j into row j of [n][B] buffers): the sampler never reads
// the tables, so a training loop can have its next 64 batches drawn by ONE launch instead of one per step
__global__ void __launch_bounds__(64) sample_batches_kernel(SampleArgs a, int n_batches) {
    const int j = (int)blockIdx.y, r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const uint64_t cur = a.step_dev ? *a.step_dev : 0ull;
    if (a.step_next && j == 0 && r == 0) *a.step_next = cur + (uint64_t)n_batches;
    a.step += cur + (uint64_t)j;
    a.step_dev = nullptr;
    a.step_next = nullptr;
    const size_t off = (size_t)j * a.B;
    a.users += off;
    a.pos += off;
    a.neg += off;
    if (a.pos_pop) { a.pos_pop += off; a.neg_pop += off; }
    sample_one(a, r);
}

}  // namespace

namespace {
__global__ void __launch_bounds__(256) remap_items_kernel(uint64_t* __restrict__ keys, size_t n, const int32_t* __restrict__ gid, int n_gid) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys[i];
    if (k == 0ull) return;
    const uint32_t local = 0xFFFFFFFFu - (uint32_t)k;
    if (local >= (uint32_t)n_gid) return;                     // (not a row of the table: left as it is)
    keys[i] = (k & 0xFFFFFFFF00000000ull) | (uint64_t)(0xFFFFFFFFu - (uint32_t)gid[local]);
}
}  // namespace

extern "C" int pda_topk_remap_items(uint64_t* keys, size_t n_keys, const int32_t* gid, int n_gid, void* stream) {
    if (!keys || !gid || n_gid <= 0) return PDA_ERR_ARG;
    if (n_keys == 0) return PDA_OK;
    hipLaunchKernelGGL(remap_items_kernel, dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), keys, n_keys, gid, n_gid);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

namespace {
// ------------------------------------------------------------------------------------------------
// Dense ratings: batch_ratings / condition_ratings as a matrix (MF/model_api.py:62,113 fetched by DatasetApi_Model.testing,
// MF/train_new_api.py:642-669: the NeuRec evaluators' protocol -- the reference imports and never calls it).  Every score is the
// k-ordered fmaf chain of the top-K kernels (oracle/pda_oracle.c:dot_chain), so out[r][j] equals the value a top-K call returns
// for that pair bit for bit.  A 16-user x 64-item tile per workgroup, the users' rows in LDS; not a hot path: the point of the
// top-K entry points is that this matrix is never formed.
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(256) score_dense_kernel(const float* __restrict__ U, const float* __restrict__ I, const float* __restrict__ pop,
                                                          const int32_t* __restrict__ users, int n_users_blk, const int32_t* __restrict__ items,
                                                          int n_items, int head, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float su[16][D + 4];
    const int tid = threadIdx.x;
    const int u0 = blockIdx.y * 16, j0 = blockIdx.x * 64;
    for (int q = tid; q < 16 * (D / 4); q += 256) {
        const int r = q / (D / 4), c = q % (D / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (u0 + r < n_users_blk) v = *reinterpret_cast<const f32x4*>(U + (size_t)users[u0 + r] * D + 4 * c);
        *reinterpret_cast<f32x4*>(&su[r][4 * c]) = v;
    }
    __syncthreads();
    const int r = tid >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + (tid & 15) + 16 * q;
        if (u0 + r >= n_users_blk || j >= n_items) continue;
        const int item = items ? items[j] : j;
        const float* v = I + (size_t)item * D;
        float acc[2] = {0.f, 0.f};
#pragma unroll
        for (int c = 0; c < D / 8; ++c) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(v + 8 * c), b = *reinterpret_cast<const f32x4*>(v + 8 * c + 4);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                acc[c & 1] = fmaf(su[r][8 * c + s2], a[s2], acc[c & 1]);
                acc[c & 1] = fmaf(su[r][8 * c + 4 + s2], b[s2], acc[c & 1]);
            }
        }
        float sc = acc[0] + acc[1];
        if (head == PDA_HEAD_POP) sc = (sc > 0.0f ? sc + 1.0f : __expf(sc)) * pop[j];
        out[(size_t)(u0 + r) * n_items + j] = sc;
    }
}
}  // namespace

extern "C" int pda_score_dense_f32(const float* U, const float* I, const float* pop, const int32_t* users, int n_users_blk, const int32_t* items,
                                   int n_items, int d, int head, float* out, void* stream) {
    if (!U || !I || !users || !out || n_users_blk <= 0 || n_items <= 0) return PDA_ERR_ARG;
    if (head != PDA_HEAD_RAW && head != PDA_HEAD_POP) return PDA_ERR_ARG;
    if (head == PDA_HEAD_POP && !pop) return PDA_ERR_ARG;
    const dim3 grid((unsigned)((n_items + 63) / 64), (unsigned)((n_users_blk + 15) / 16));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (d) {
        case 32: hipLaunchKernelGGL(score_dense_kernel<32>, grid, dim3(256), 0, s, U, I, pop, users, n_users_blk, items, n_items, head, out); break;
        case 64: hipLaunchKernelGGL(score_dense_kernel<64>, grid, dim3(256), 0, s, U, I, pop, users, n_users_blk, items, n_items, head, out); break;
        case 128: hipLaunchKernelGGL(score_dense_kernel<128>, grid, dim3(256), 0, s, U, I, pop, users, n_users_blk, items, n_items, head, out); break;
        case 256: hipLaunchKernelGGL(score_dense_kernel<256>, grid, dim3(256), 0, s, U, I, pop, users, n_users_blk, items, n_items, head, out); break;
        default: return PDA_ERR_UNSUPPORTED;
    }
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

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
    hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0,
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
    hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

extern "C" int pda_group_triplets_by_pos_batches(int32_t* users, int32_t* pos, int32_t* neg, float* pos_pop, float* neg_pop, int B,
                                                 int n_batches, void* stream);

extern "C" int pda_sample_batches_dev(int32_t* users, int gen_users, const int32_t* user_pool, int n_pool, int B, int n_batches,
                                      const int64_t* train_indptr, const int32_t* train_indices, const int32_t* train_slots, int neg_lo,
                                      int neg_hi, const float* pop_matrix, int n_slots, uint64_t seed, const uint64_t* step_dev,
                                      uint64_t* step_next, int32_t* pos, int32_t* neg, float* pos_pop, float* neg_pop, int group_by_pos,
                                      void* stream) {
    if (!users || !train_indptr || !train_indices || !pos || !neg || !step_dev || B <= 0 || n_batches <= 0 || n_batches > 65535 ||
        neg_hi <= neg_lo)
        return PDA_ERR_ARG;
    if (step_next == step_dev) return PDA_ERR_ARG;
    if (gen_users && n_pool <= 0) return PDA_ERR_ARG;
    if (pop_matrix && (!pos_pop || !neg_pop || n_slots <= 0)) return PDA_ERR_ARG;
    SampleArgs a{users, user_pool, train_indptr, train_indices, train_slots, pop_matrix, pos, neg, pos_pop, neg_pop,
                 seed, 0, B, n_pool, gen_users, neg_lo, neg_hi, n_slots, step_dev, step_next};
    hipLaunchKernelGGL(sample_batches_kernel, dim3((unsigned)((B + 63) / 64), (unsigned)n_batches), dim3(64), 0,
                       reinterpret_cast<hipStream_t>(stream), a, n_batches);
    PDA_CHECK_LAUNCH();
    if (group_by_pos) return pda_group_triplets_by_pos_batches(users, pos, neg, pos_pop, neg_pop, B, n_batches, stream);
    return PDA_OK;
}

extern "C" int pda_counter_add(uint64_t* counter, uint64_t inc, void* stream) {
    if (!counter) return PDA_ERR_ARG;
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), counter, inc);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// Peaks measured on the box ("Peaks must be measured"; bench.py reports them beside the datasheet
// figures).  The matrix peak: 2 waves per SIMD x 4 independent accumulator chains of v_mfma_f32_32x32x16_bf16 on operands
// held in registers -- the configuration that saturates the pipe (tools/ubench/mfma_lds.hip); operands are pseudo-random
// bit patterns (zero-filled operands clock ~20 % higher: guides/MI355X_MICROARCH.md, DVFS).  The memory peak: a float4
// copy, one pass, grid-stride.
// ------------------------------------------------------------------------------------------------------------------------
namespace {
typedef __bf16 peak_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned peak_u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) peak_mfma_kernel(float* sink, int iters, unsigned seed, int constant_operands) {
    peak_u32x4 a[4], b[4];
    unsigned x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    for (int q = 0; q < 4; ++q)
        for (int k = 0; k < 4; ++k) {
            x = x * 1664525u + 1013904223u;
            a[q][k] = (x & 0x7FFF7FFFu) | 0x3C003C00u;        // bf16 pairs of moderate magnitude, random mantissas and signs cleared
            a[q][k] ^= (x >> 3) & 0x80008000u;
            x = x * 1664525u + 1013904223u;
            b[q][k] = ((x & 0x7FFF7FFFu) | 0x3C003C00u) ^ ((x >> 5) & 0x80008000u);
            if (constant_operands) a[q][k] = b[q][k] = 0x3F803F80u;      // every operand 1.0: no toggling in the multipliers (the guide's 2 495 TF)
        }
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(peak_bf16x8, a[(q + m) & 3]), __builtin_bit_cast(peak_bf16x8, b[q]),
                                                                 acc[q], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    if (s == 12345.678f) sink[0] = s;                          // keeps the chains alive
}

__global__ void __launch_bounds__(256) peak_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n4) {
    // four 16-byte loads in flight per lane and pass; a block walks one contiguous 16 KiB chunk at a time
    const size_t stride = (size_t)gridDim.x * 1024;
    for (size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x; base < n4; base += stride) {
        f32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = base + 256 * q < n4 ? __builtin_nontemporal_load(&src[base + 256 * q]) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) if (base + 256 * q < n4) __builtin_nontemporal_store(v[q], &dst[base + 256 * q]);
    }
}
}  // namespace

extern "C" double pda_peak_mfma_flops_per_launch(int iters) { return 256.0 * 4 * 8 * (double)iters * 16.0 * 2.0 * 32 * 32 * 16; }

extern "C" int pda_peak_mfma_bf16(float* sink, int iters, void* stream) {
    if (!sink || iters <= 0) return PDA_ERR_ARG;
    hipLaunchKernelGGL(peak_mfma_kernel, dim3(256 * 4), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), sink, iters, 2021u, 0);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}
extern "C" int pda_peak_mfma_bf16_const(float* sink, int iters, void* stream) {
    if (!sink || iters <= 0) return PDA_ERR_ARG;
    hipLaunchKernelGGL(peak_mfma_kernel, dim3(256 * 4), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), sink, iters, 2021u, 1);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

// The roof of the sweep's own KIND of loop (round 3): v_mfma_f32_32x32x16_bf16 whose B operand comes from the LDS -- one
// ds_read_b128 per MFMA, two MFMA waves per SIMD x 32 user rows, 18 MFMAs per 64-item block through pda_v4_block_asm.h (the very
// statement the sweep executes), then a VALU read of the block's accumulators -- while four loader waves stream the tiles into two
// LDS slots by LDS-DMA at the sweep's rate, with NO hand-over, no lists, no candidates.  What this reaches on the caller's data
// (random bf16 rows: the chip is power-limited, constant operands clock a third higher) is what is left for the sweep to lose.
#include "pda_topk_common.h"
namespace {
using pda_topk::u32x4;
#include "pda_v4_block_asm.h"
__global__ void __launch_bounds__(1024) peak_mfma_lds_kernel(const unsigned char* __restrict__ rows, size_t n_tiles, unsigned* out, int n_blk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_peak[];
    constexpr int RB = 304, HB = 32 * RB, BB = 2 * HB, NP = 19, LW = 4, MYP = (NP + LW - 1) / LW;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * BB / 4; i += 1024) reinterpret_cast<unsigned*>(smem_peak)[i] = 0x3c003c00u;
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_peak;
    if (wave < 8) {
        const int j = lane & 31, h = lane >> 5;
        u32x4 a[8], aex;
        for (int m = 0; m < 8; ++m) {
            unsigned x = (977u * tid + 131u * m + blockIdx.x) * 2654435761u;
            for (int q = 0; q < 4; ++q) {
                x = x * 1664525u + 1013904223u;
                a[m][q] = (x & 0x807f807fu) | 0x3c003c80u;
            }
        }
        aex = u32x4{0x3c003c00u + (unsigned)tid, 0xbc00bc00u, 0u, 0u};
        f32x16 acc0, acc1;
        u32x4 pi;
        unsigned sink = 0;
        const unsigned base = lds0 + (unsigned)(j * RB + 16 * h);
        for (int b = 0; b < n_blk; ++b) {
            const unsigned addr = base + (unsigned)((b & 1) * BB);
            BlockAsm<128>::run(acc0, acc1, pi, a, aex, addr, addr - 16u * h + 288u);
#pragma unroll
            for (int r = 0; r < 16; ++r) sink |= __float_as_uint(acc0[r]) | __float_as_uint(acc1[r]);
            sink |= pi[0];
            if (__builtin_expect(__any((int)sink < 0 && (sink & 0x7fffffffu) == 0x12345u), 0)) out[1] = sink;
        }
        if (sink == 0x7654321u) out[2] = sink;
    } else if (wave < 12) {
        const int l = wave - 8;
        const size_t tile0 = ((size_t)blockIdx.x * 977) % (n_tiles - (size_t)n_blk - 2);
        for (int b = 0; b < n_blk; ++b) {
            [[maybe_unused]] const unsigned char* src = rows + (tile0 + (size_t)b) * BB + lane * 16;
            [[maybe_unused]] const unsigned dst = lds0 + (unsigned)((b & 1) * BB);
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
            __builtin_amdgcn_s_sleep(6);          // the loaders keep to the MFMA waves' pace (they would run ahead otherwise)
#endif
        }
    } else {
        for (int b = 0; b < n_blk; ++b) __builtin_amdgcn_s_sleep(16);     // four idle waves, like rescoring waves without candidates
    }
    if (tid == 0 && blockIdx.x == 0) out[0] = (unsigned)n_blk;
}
}  // namespace

extern "C" double pda_peak_mfma_lds_flops_per_launch(int n_blk) { return 1024.0 * 8 * 18 * (double)n_blk * 2.0 * 32 * 32 * 16; }

// rows: at least (n_blk + 1024) * 19 456 bytes of bf16 data the caller considers typical; out: 4 words of scratch
extern "C" int pda_peak_mfma_lds_bf16(const void* rows, size_t n_bytes, void* out, int n_blk, void* stream) {
    if (!rows || !out || n_blk <= 0 || n_bytes < ((size_t)n_blk + 1024) * 19456) return PDA_ERR_ARG;
    constexpr int lds = 156 * 1024;               // one workgroup per CU, like the sweep (tiles + lists)
    static int attr_set = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&peak_mfma_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return PDA_ERR_LAUNCH;
        attr_set = 1;
    }
    hipLaunchKernelGGL(peak_mfma_lds_kernel, dim3(1024), dim3(1024), lds, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const unsigned char*>(rows), n_bytes / 19456, reinterpret_cast<unsigned*>(out), n_blk);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

extern "C" int pda_peak_copy(const float* src, float* dst, size_t n_floats, void* stream) {
    if (!src || !dst || n_floats < 4) return PDA_ERR_ARG;
    hipLaunchKernelGGL(peak_copy_kernel, dim3(256 * 16), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const f32x4*>(src), reinterpret_cast<f32x4*>(dst), n_floats / 4);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}
