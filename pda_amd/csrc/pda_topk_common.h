// Shared pieces of the score + top-K kernels (v1: exact fp32 MFMA; v2: bf16x3 MFMA pre-filter + exact fp32 rescoring).
#pragma once
#include "pda_common.h"

namespace pda_topk {

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kUserTile = 128;  // users per workgroup
constexpr int kCap = PDA_TOPK_CAP;

struct ScoreArgs {
    const float* U;   // f32 tables; bf16 tables (uint16 bits) behind the same pointers in the *_bf16 entry points
    const float* I;
    const float* pop;
    const int32_t* users;
    const int64_t* hist_indptr;
    const int32_t* hist_indices;
    uint64_t* out_keys;
    int n_users_blk;
    int item_offset;
    int n_items_local;
    int hist_row_mode;
    int K;
    int n_splits;
    const int* tile_flags;   // optional [n_user_tiles]: only flagged tiles are computed (v2's exact fallback)
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// arguments of the pre-filtered kernels (v2: approximate lists; v3: candidate ring + exact lists)
struct ScoreArgs2 {
    ScoreArgs a;
    const uint16_t* I_hi;   // bf16 [n_items_local, d]
    const uint16_t* I_lo;   // bf16 [n_items_local, d]
    const float* I_norm;    // f32  [n_items_local]   ||i||_2 * (1+2^-10), followed (256-B aligned) by the max over the shard
    const float* I_norm_max;
    const float* pop_max;   // workspace: max |pop| over the shard (PDA_HEAD_POP)
    int* tile_flags;        // workspace: [n_user_tiles], set when a row's near-tie band overflowed
    // ordered sweep (pda_score_topk_ordered_f32): the planes / norms above are stored in VISITING order
    const int* order;       // [n_items_local] visiting position -> local item id
    const float* pop_p;     // [n_items_local] pop in visiting order (PDA_HEAD_POP)
    const float* sufA;      // [n_tiles] max over positions >= 32 t of |pop|            (1 for PDA_HEAD_RAW -> unused, 0)
    const float* sufB;      // [n_tiles] max over positions >= 32 t of |pop| * ||i||    (||i|| for PDA_HEAD_RAW)
    unsigned long long* visited;   // workspace: item tiles actually scored, summed over workgroups (statistics)
    // v3 "folded test": 16 bf16 per item (one extra MFMA k-step) that subtract threshold/pop + 1 + eps inside the matrix pipe
    const uint16_t* I_bex;  // [n_items_local][16]: k 0..7 pieces of 1/pop (1 for PDA_HEAD_RAW), k 8..10 constants and ||i||
    const int32_t* hist_nat;   // the caller's history (item ids ascending per row) also for ordered sweeps: v3 masks at the candidate stage
};

__device__ __forceinline__ uint32_t bf16_rne(float x) {
    uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// x = t1 + t2 + t3 exactly, every piece a bf16 (truncation split; x finite).  Returned as bf16 bit patterns.
__device__ __forceinline__ void bf16_split3(float x, uint32_t& t1, uint32_t& t2, uint32_t& t3) {
    const uint32_t b1 = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(b1);                  // exact: <= 16 significant bits
    const uint32_t b2 = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(b2);                 // exact: <= 8 significant bits, a bf16
    t1 = b1 >> 16;
    t2 = b2 >> 16;
    t3 = __float_as_uint(r2) >> 16;
}
// smallest bf16 >= x  (x >= 0, finite)
__device__ __forceinline__ uint32_t bf16_up(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u >> 16) + ((u & 0xFFFFu) ? 1u : 0u);
}
// split 8 floats into packed bf16 hi / lo words
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    uint32_t h[8], l[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        h[k] = bf16_rne(v[k]);
        l[k] = bf16_rne(v[k] - __uint_as_float(h[k] << 16));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        hi[k] = h[2 * k] | (h[2 * k + 1] << 16);
        lo[k] = l[2 * k] | (l[2 * k + 1] << 16);
    }
}

template <int D>
__device__ __forceinline__ int swzb(int row) {   // bf16 tile: D/8 16-byte chunks per row
    constexpr int CPR = D / 8;
    if constexpr (CPR >= 16) return row & 15;
    else if constexpr (CPR == 8) return (row >> 1) & 7;
    else return (row >> 2) & 3;
}

// defined in pda_score_topk_v3.hip: 1-MFMA pre-filter + candidate ring + exact rescoring (no fallback needed)
int launch_score_v3(const ScoreArgs2& aa, int d, int head, bool ordered, bool bf16_tables, hipStream_t stream);

// defined in pda_score_topk.hip
int launch_score_v1(const ScoreArgs& a, int d, int head, hipStream_t stream, bool bf16_tables = false);

template <int D>
__device__ __forceinline__ int swz(int row) {
    // chunks (16 B) per row = D/4.  The B read of one 16-lane group touches 16 different rows at the
    // same logical chunk; XOR with a row-derived value spreads them over all sixteen 16-B bank slots.
    if constexpr (D / 4 >= 16) return row & 15;
    else return (row >> 1) & 7;  // D == 32: 8 chunks per 128-B row, two rows per 256-B bank line
}

// Compaction of one user's candidate list (<= 60 keys, one per lane): keep the best K at buf[0..K) in
// descending order, update the row's LDS count and threshold.  Whole-wave call.
// After the first compaction buf[0..K) is already sorted, so only the (<= kCap-K) appended keys need ranking:
// rank(old i) = i + #new greater; rank(new) = #old greater (one ballot) + #new greater.  ~4x cheaper than the
// generic all-pairs rank sort, which remains for the first compaction and the final sort.
// GLB: the list lives in HBM (generation 4, 512-user workgroups): the wave's own stores must have reached the cache
// before other lanes read them back -- a workgroup-scope fence (waits for the outstanding stores; the L1 is the CU's).
template <bool GLB>
__device__ __forceinline__ void list_sync() {
    if constexpr (GLB) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    } else {
        pda_wave_sync();
    }
}

template <int CAP = kCap, bool GLB = false>
// flag_word / flag_bit (optional): a set bit says "this list came in UNSORTED" (generation 4 hands its warm-up lists over
// unsorted): the incremental path is off until the list has been through one full compaction, which clears the bit.
// Returns the row's threshold behind the compaction (-inf while it holds fewer than K keys).
__device__ __forceinline__ float compact_list(uint64_t* buf, int* cnt_slot, float* tau_slot, int K, int lane, unsigned* flag_word = nullptr,
                                              unsigned flag_bit = 0u) {
    list_sync<GLB>();
    // ONE round trip for the count, the threshold, the flag and the keys: all four requested before the first is looked at (as four
    // dependent trips beside eight MFMA waves streaming B fragments they were most of a compaction's 2 200 cycles; lanes at and
    // behind CAP read slot CAP - 1 -- a row's CAP slots are its own)
    const int cnt_v = *cnt_slot;
    const int tau_v = __float_as_int(*tau_slot);
    const unsigned flag_v = flag_word != nullptr ? *flag_word : 0u;
    const uint64_t kraw = buf[lane < CAP ? lane : CAP - 1];
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(cnt_v), "v"(tau_v), "v"(flag_v), "v"(kraw));
#endif
    const int c = min(__builtin_amdgcn_readfirstlane(cnt_v), CAP);        // failed appends may have pushed it past kCap
    bool sorted_prefix = __builtin_amdgcn_readfirstlane(tau_v) != (int)0xff800000 && c >= K;
    if (flag_word != nullptr) {
        const unsigned fw = __builtin_amdgcn_readfirstlane(flag_v);
        if (fw & flag_bit) {
            sorted_prefix = false;
            if (lane == 0) *flag_word = fw & ~flag_bit;
        }
    }
    uint64_t key = lane < c ? kraw : (uint64_t)(63 - lane);      // fillers: unique, below any real key
    int rank;
    // First compaction of a row (all <= 59 keys against each other): the keys to rank against come from LDS as broadcast
    // reads, four in flight -- no SGPR round trip per key.  32 of these per wave were 40 % of the exact warm-up of a sweep.
    if (sorted_prefix) {
        rank = lane < K ? lane : 0;
        const uint64_t oldmask = K >= 64 ? ~0ull : ((1ull << K) - 1ull);
        // the first eight new keys without a loop: their broadcasts, compares and ballots do not depend on one another (as a loop of
        // c - K dependent rounds this was ~1 000 of a compaction's 2 200 cycles, and a compaction every seven insertions is half
        // of what a candidate costs its rescoring wave)
        constexpr int NU = 8;
        const int nnew = c - K;
        uint64_t kn[NU];
#pragma unroll
        for (int t = 0; t < NU; ++t) {
            const uint64_t kt = pda_readlane_u64(key, K + t < 63 ? K + t : 63);
            kn[t] = t < nnew ? kt : 0ull;                       // (0: below every key)
        }
#pragma unroll
        for (int t = 0; t < NU; ++t) rank += (kn[t] > key) ? 1 : 0;
#pragma unroll
        for (int t = 0; t < NU; ++t) {
            const int olds_above = __popcll(__ballot(key > kn[t]) & oldmask);
            rank += (lane == K + t && t < nnew) ? olds_above : 0;
        }
        for (int jj = K + NU; jj < c; ++jj) {
            const uint64_t kj = pda_readlane_u64(key, jj);      // (<= 9 keys: an LDS read per key would only add latency)
            rank += (kj > key) ? 1 : 0;
            const int olds_above = __popcll(__ballot(key > kj) & oldmask);
            rank += (lane == jj) ? olds_above : 0;
        }
    } else {
        // (the compares are what a sort costs -- four waves per SIMD at it behind an early-terminating sweep --: whole groups of four
        // keys without masks, two instructions per key; the masks of the last, partial group were 1.5 more per key everywhere)
        rank = 0;
        const int c4 = c & ~3;
        for (int jj = 0; jj < c4; jj += 4) {
            const uint64_t k0 = buf[jj], k1 = buf[jj + 1], k2 = buf[jj + 2], k3 = buf[jj + 3];
            rank += ((k0 > key) ? 1 : 0) + ((k1 > key) ? 1 : 0) + ((k2 > key) ? 1 : 0) + ((k3 > key) ? 1 : 0);
        }
        if (c4 < c) {                             // reads past c stay inside the workgroup's LDS and are masked
            uint64_t k0 = buf[c4], k1 = buf[c4 + 1], k2 = buf[c4 + 2];
            k1 = c4 + 1 < c ? k1 : 0ull;
            k2 = c4 + 2 < c ? k2 : 0ull;
            rank += ((k0 > key) ? 1 : 0) + ((k1 > key) ? 1 : 0) + ((k2 > key) ? 1 : 0);
        }
    }
    list_sync<GLB>();
    if (lane < c && rank < K) buf[rank] = key;
    float tau = -INFINITY;
    if (c >= K) {
        uint64_t mk = __ballot(lane < c && rank == K - 1);
        int src = __builtin_ctzll(mk);
        tau = pda_unordf((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), src));
        if (lane == 0) {
            *tau_slot = tau;
            *cnt_slot = K;
        }
    } else if (lane == 0) {
        *cnt_slot = c;
    }
    list_sync<GLB>();
    return tau;
}


}  // namespace pda_topk
