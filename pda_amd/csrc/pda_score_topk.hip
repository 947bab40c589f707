// Full-catalogue score + history mask + top-K for MI355X (gfx950).
//
// Replaces the TF graph  Gather -> MatMul[Bu,I] -> Elu,+1,*pop -> SparseAdd(-inf) -> TopKV2
// (MF/model_api.py:62,113; MF/train_new_api.py:594-612, driven from :780-794) without ever
// materialising the [Bu, I] rating matrix.
//
// Shape of the computation: a dense fp32 contraction [users, d] x [d, items] (compute-bound: the
// item shard is re-used by every user tile out of L2/Infinity Cache) with a streaming top-K epilogue.
//
//   workgroup  = 256 threads = 4 waves, owns a tile of 128 users; 2 workgroups per CU
//                (2 waves / SIMD, <=256 VGPR, <=80 KB LDS each) so that one workgroup's epilogue
//                overlaps the other's MFMA phase without any hand-written skew.
//   wave       = 32 user rows.  Their embedding rows live in VGPRs for the whole sweep
//                (A operand of v_mfma_f32_32x32x2_f32: d/2 registers).
//   item tile  = 32 items x d, staged once per workgroup into LDS with coalesced 16 B loads
//                (whole 4*d-byte rows), XOR-swizzled so that the B-operand ds_read_b128 of the
//                four 16-lane groups are bank-conflict free; next tile prefetched into registers
//                while the MFMAs of the current one run.
//   k order    = lane-half h supplies k = 8c+4h+s for MFMA (c, s): both operands become one
//                16-byte load per 8 k's, and the result is a fixed fmaf chain (oracle order 1).
//   epilogue   = per accumulator register: head transform ((elu+1)*pop), one v_cmp against the
//                per-user running threshold, wave ballot.  Only when some lane passes (rare after
//                warm-up: ~K/i of the scores at item i) does the wave enter the slow path, which
//                appends (key) candidates to the user's 60-slot LDS list and, when it fills,
//                compacts it with an in-register wave-wide rank sort and raises the threshold.
//   mask       = history CSR rows sorted by item id; each user row keeps a cursor, producing a
//                32-bit mask per (user, tile) that is consulted only on the slow path.
#include "pda_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kUserTile = 128;  // users per workgroup
constexpr int kCap = PDA_TOPK_CAP;

struct ScoreArgs {
    const float* U;
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
};

template <int D>
__device__ __forceinline__ int swz(int row) {
    // chunks (16 B) per row = D/4.  The B read of one 16-lane group touches 16 different rows at the
    // same logical chunk; XOR with a row-derived value spreads them over all sixteen 16-B bank slots.
    if constexpr (D / 4 >= 16) return row & 15;
    else return (row >> 1) & 7;  // D == 32: 8 chunks per 128-B row, two rows per 256-B bank line
}

// In-register rank sort of one user's candidate list (<= 60 keys, one per lane), keeping the best K
// at buf[0..K) in descending order.  Returns the new count and threshold.
__device__ __forceinline__ void compact_list(uint64_t* buf, int& c, float& tau, int K, int lane) {
    pda_wave_sync();
    uint64_t key = lane < c ? buf[lane] : (uint64_t)(63 - lane);  // fillers: unique, below any real key
    int rank = 0;
    for (int jj = 0; jj < c; ++jj) {
        uint64_t kj = pda_readlane_u64(key, jj);
        rank += (kj > key) ? 1 : 0;
    }
    pda_wave_sync();
    if (lane < c && rank < K) buf[rank] = key;
    if (c >= K) {
        uint64_t mk = __ballot(lane < c && rank == K - 1);
        int src = __builtin_ctzll(mk);
        tau = pda_unordf((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), src));
        c = K;
    }
    pda_wave_sync();
}

template <int D, int HEAD>
__global__ void __launch_bounds__(kThreads, (D <= 128 ? 2 : 1)) score_topk_kernel(ScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Bt = reinterpret_cast<float*>(smem);                                   // [32][D] swizzled
    uint64_t* lists = reinterpret_cast<uint64_t*>(smem + 32 * D * sizeof(float));  // [128][kCap]

    constexpr int CPR = D / 4;            // 16-B chunks per item row
    constexpr int NLD = (32 * CPR) / kThreads > 0 ? (32 * CPR) / kThreads : 1;  // float4 loads / thread / tile
    constexpr int NC = D / 8;             // k-chunks of 8

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int split = blockIdx.x % a.n_splits, utile = blockIdx.x / a.n_splits;
    const int K = a.K;

    const int tiles_total = (a.n_items_local + 31) >> 5;
    const int tiles_per = (tiles_total + a.n_splits - 1) / a.n_splits;
    const int t0 = split * tiles_per;
    const int t1 = min(t0 + tiles_per, tiles_total);

    // ---- this lane's user row (rows are indexed by lane&31 in both halves) -------------------
    const int row_blk = utile * kUserTile + wave * 32 + j;
    const bool row_ok = row_blk < a.n_users_blk;
    const int uid = row_ok ? a.users[row_blk] : 0;

    f32x4 areg[NC];
    {
        const float* up = a.U + (size_t)uid * D + 4 * h;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row_ok) v = *reinterpret_cast<const f32x4*>(up + 8 * c);
            areg[c] = v;
        }
    }

    // ---- history cursor ---------------------------------------------------------------------
    int64_t hp = 0, he = 0;
    int nxt = 0x7fffffff;
    if (a.hist_indptr != nullptr && row_ok) {
        const int64_t hr = a.hist_row_mode == PDA_HIST_BY_USER_ID ? (int64_t)uid : (int64_t)row_blk;
        hp = a.hist_indptr[hr];
        he = a.hist_indptr[hr + 1];
        const int lo_item = a.item_offset + t0 * 32;
        int64_t lo = hp, hi = he;  // lower_bound(lo_item)
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (a.hist_indices[mid] < lo_item) lo = mid + 1; else hi = mid;
        }
        hp = lo;
        if (hp < he) nxt = a.hist_indices[hp];
    }

    // ---- per-row running state: count + threshold (lane l <-> row l&31) ------------------------
    int cnt = 0;
    float tau = row_ok ? -INFINITY : INFINITY;  // rows past the end never accept anything
    f32x16 thr;
#pragma unroll
    for (int r = 0; r < 16; ++r) thr[r] = __shfl(tau, (r & 3) + 8 * (r >> 2) + 4 * h, 64);

    // ---- item tile staging -------------------------------------------------------------------
    f32x4 pre[NLD];
    auto tile_load = [&](int t) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int id = tid + kThreads * q;
            const int jj = id / CPR, ch = id % CPR;
            const int it = t * 32 + jj;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (jj < 32 && it < a.n_items_local) v = *reinterpret_cast<const f32x4*>(a.I + (size_t)it * D + 4 * ch);
            pre[q] = v;
        }
    };
    auto tile_store = [&]() {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int id = tid + kThreads * q;
            const int jj = id / CPR, ch = id % CPR;
            if (jj < 32) *reinterpret_cast<f32x4*>(Bt + jj * D + 4 * (ch ^ swz<D>(jj))) = pre[q];
        }
    };

    if (t0 < t1) {
        tile_load(t0);
        tile_store();
    }
    __syncthreads();

    uint64_t* my_lists = lists + (size_t)(wave * 32) * kCap;
    const float* brow = Bt + j * D;
    const int bswz = swz<D>(j);

    for (int t = t0; t < t1; ++t) {
        const int j0 = t * 32;
        const bool has_next = (t + 1) < t1;
        if (has_next) tile_load(t + 1);

        float popj = 1.0f;
        if constexpr (HEAD == PDA_HEAD_POP) popj = (j0 + j < a.n_items_local) ? a.pop[j0 + j] : 0.0f;

        // history bits of this tile for my row
        const int jg0 = a.item_offset + j0, jg1 = jg0 + 32;
        uint32_t hb = 0;
        while (__any(nxt < jg1)) {
            if (nxt < jg1) {
                hb |= 1u << (nxt - jg0);
                ++hp;
                nxt = hp < he ? a.hist_indices[hp] : 0x7fffffff;
            }
        }

        // ---- contraction: 32 users x 32 items x D on the matrix cores -------------------------
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(brow + 4 * ((2 * c + h) ^ bswz));
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[c][0], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[c][1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[c][2], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[c][3], b[3], acc, 0, 0, 0);
        }

        __syncthreads();  // every wave is done reading Bt
        if (has_next) tile_store();

        // ---- epilogue: head transform + threshold test (fast path) ----------------------------
        const int nvalid = min(32, a.n_items_local - j0);
        const uint64_t vmask = nvalid >= 32 ? ~0ull : (((1ull << nvalid) - 1ull) * 0x100000001ull);
        f32x16 tv;
        uint64_t any = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = acc[r];
            if constexpr (HEAD == PDA_HEAD_POP) s = (s > 0.0f ? s + 1.0f : __expf(s)) * popj;
            tv[r] = s;
            any |= __ballot(s > thr[r]);
        }
        any &= vmask;

        if (any) {
            // ---- slow path: some (user, item) beats the user's current threshold ---------------
            const uint32_t my_item = (uint32_t)(jg0 + j);
#pragma nounroll
            for (int r = 0; r < 16; ++r) {
                const float tt = tv[r];
                const uint64_t m = __ballot(tt > thr[r]) & vmask;
                if (!m) continue;
                const uint64_t key = pda_pack_key(tt, my_item);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    uint32_t mh = (uint32_t)(m >> (32 * half));
                    if (!mh) continue;
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    mh &= ~(uint32_t)__builtin_amdgcn_readlane((int)hb, row);  // train items never enter
                    if (!mh) continue;
                    int c = __builtin_amdgcn_readlane(cnt, row);
                    float ta = pda_readlane_f32(tau, row);
                    uint64_t* buf = my_lists + row * kCap;
                    const bool mine = (h == half);
                    for (;;) {
                        const int n = __popc(mh);
                        const int room = kCap - c;
                        if (n > room && c > K) {
                            compact_list(buf, c, ta, K, lane);
                            mh &= (uint32_t)(__ballot(tt > ta) >> (32 * half));  // re-filter with the new threshold
                            if (!mh) break;
                            continue;
                        }
                        const int take = min(n, room);
                        const int rank = __popc(mh & ((1u << j) - 1u));
                        const bool doit = mine && ((mh >> j) & 1u) && rank < take;
                        if (doit) buf[c + rank] = key;
                        c += take;
                        if (take == n) break;
                        // list full (c == kCap > K): drop the lanes just stored, compact on the next turn
                        mh &= ~(uint32_t)(__ballot(doit) >> (32 * half));
                    }
                    if (j == row) {
                        cnt = c;
                        tau = ta;
                    }
                    thr[r] = mine ? ta : thr[r];
                }
            }
        }
        __syncthreads();  // next tile visible in Bt
    }

    // ---- finalise: sort every row's list, emit K packed keys (0 = empty) -------------------------
    for (int rr = 0; rr < 32; ++rr) {
        int c = __builtin_amdgcn_readlane(cnt, rr);
        float ta = 0.f;
        uint64_t* buf = my_lists + rr * kCap;
        compact_list(buf, c, ta, K, lane);  // c <- min(c, K), buf sorted best-first
        const int rb = utile * kUserTile + wave * 32 + rr;
        if (rb < a.n_users_blk && lane < K) {
            const uint64_t k = lane < c ? buf[lane] : 0ull;
            a.out_keys[((size_t)split * a.n_users_blk + rb) * K + lane] = k;
        }
    }
}

template <int D, int HEAD>
int launch_score(const ScoreArgs& a, hipStream_t stream) {
    const size_t smem = 32 * D * sizeof(float) + (size_t)kUserTile * kCap * sizeof(uint64_t);
    static int attr_set = 0;  // idempotent attribute; benign if raced
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&score_topk_kernel<D, HEAD>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return PDA_ERR_LAUNCH;
        attr_set = 1;
    }
    const int utiles = (a.n_users_blk + kUserTile - 1) / kUserTile;
    dim3 grid((unsigned)(utiles * a.n_splits));
    hipLaunchKernelGGL((score_topk_kernel<D, HEAD>), grid, dim3(kThreads), smem, stream, a);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

// ------------------------------------------------------------------------------------------------
// Merge of R sorted partial lists per user.  One wave per user.  rank(key) = sum over lists of
// (#keys greater than key) found by binary search in LDS; rank < K => final position.
// ------------------------------------------------------------------------------------------------
struct MergeArgs {
    const uint64_t* in_keys;
    uint64_t* out_keys;
    int32_t* out_idx;
    float* out_val;
    const int32_t* users;
    const int64_t* hist_indptr;
    const int32_t* hist_indices;
    int hist_row_mode;
    int R, n_users_blk, K;
};

__global__ void __launch_bounds__(256) topk_merge_kernel(MergeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = a.R * a.K, K = a.K;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem) + (size_t)wave * (n + PDA_MAX_K);
    uint64_t* outl = keys + n;
    const int waves_total = gridDim.x * 4;
    for (int u = blockIdx.x * 4 + wave; u < a.n_users_blk; u += waves_total) {
        pda_wave_sync();
        for (int e = lane; e < n; e += 64) {
            const int r = e / K, p = e - r * K;
            keys[e] = a.in_keys[((size_t)r * a.n_users_blk + u) * K + p];
        }
        if (lane < K) outl[lane] = 0ull;
        pda_wave_sync();
        for (int e = lane; e < n; e += 64) {
            const uint64_t key = keys[e];
            if (key == 0ull) continue;
            const int r = e / K, p = e - r * K;
            int rank = p;  // own list is sorted and keys are unique
            for (int r2 = 0; r2 < a.R; ++r2) {
                if (r2 == r) continue;
                const uint64_t* l = keys + r2 * K;
                int lo = 0, hi = K;  // first position whose key is <= mine (descending list)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (l[mid] > key) lo = mid + 1; else hi = mid;
                }
                rank += lo;
            }
            if (rank < K) outl[rank] = key;
        }
        pda_wave_sync();
        uint64_t k = lane < K ? outl[lane] : 0ull;
        const int nreal = __popcll(__ballot(lane < K && k != 0ull));
        if (lane < K) {
            if (a.out_keys) a.out_keys[(size_t)u * K + lane] = k;
            if (a.out_idx) {
                a.out_idx[(size_t)u * K + lane] = k ? pda_key_item(k) : -1;
                if (a.out_val) a.out_val[(size_t)u * K + lane] = k ? pda_key_val(k) : -INFINITY;
            }
        }
        // fewer than K unmasked items: tf.nn.top_k returns the -inf (masked) items, lowest id first
        if (nreal < K && a.out_idx && a.hist_indptr && lane == 0) {
            const int64_t hr = a.hist_row_mode == PDA_HIST_BY_USER_ID ? (int64_t)a.users[u] : (int64_t)u;
            int fill = nreal, prev = -1;
            for (int64_t p = a.hist_indptr[hr]; p < a.hist_indptr[hr + 1] && fill < K; ++p) {
                const int it = a.hist_indices[p];
                if (it != prev) a.out_idx[(size_t)u * K + fill++] = it;
                prev = it;
            }
        }
    }
}

}  // namespace

extern "C" int pda_score_topk_auto_splits(int n_users_blk, int n_items_local) {
    if (n_users_blk <= 0 || n_items_local <= 0) return 1;
    const int utiles = (n_users_blk + kUserTile - 1) / kUserTile;
    const int tiles = (n_items_local + 31) / 32;
    int s = 1;
    while (utiles * s < 512 && s < 64 && tiles / (2 * s) >= 16) s *= 2;
    return s;
}

extern "C" int pda_score_topk_f32(const float* U, const float* I_shard, const float* pop_shard, const int32_t* users,
                                  int n_users_blk, int item_offset, int n_items_local, int d,
                                  const int64_t* hist_indptr, const int32_t* hist_indices, int hist_row_mode, int K,
                                  int head, int n_splits, uint64_t* out_keys, void* stream) {
    if (!U || !I_shard || !users || !out_keys) return PDA_ERR_ARG;
    if (n_users_blk <= 0 || n_items_local <= 0 || item_offset < 0) return PDA_ERR_ARG;
    if (K < 1 || K > PDA_MAX_K || K > PDA_TOPK_CAP - 1) return PDA_ERR_ARG;
    if (head != PDA_HEAD_RAW && head != PDA_HEAD_POP) return PDA_ERR_ARG;
    if (head == PDA_HEAD_POP && !pop_shard) return PDA_ERR_ARG;
    if (hist_indptr && !hist_indices) return PDA_ERR_ARG;
    if (n_splits <= 0) n_splits = pda_score_topk_auto_splits(n_users_blk, n_items_local);
    ScoreArgs a{U, I_shard, pop_shard, users, hist_indptr, hist_indices, out_keys,
                n_users_blk, item_offset, n_items_local, hist_row_mode, K, n_splits};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define PDA_DISPATCH(DD)                                                  \
    case DD:                                                              \
        return head == PDA_HEAD_POP ? launch_score<DD, PDA_HEAD_POP>(a, s) \
                                    : launch_score<DD, PDA_HEAD_RAW>(a, s);
    switch (d) {
        PDA_DISPATCH(32)
        PDA_DISPATCH(64)
        PDA_DISPATCH(128)
        PDA_DISPATCH(256)
        default:
            return PDA_ERR_UNSUPPORTED;
    }
#undef PDA_DISPATCH
}

extern "C" int pda_topk_merge(const uint64_t* in_keys, int R, int n_users_blk, int K, uint64_t* out_keys,
                              int32_t* out_idx, float* out_val, const int32_t* users, const int64_t* hist_indptr,
                              const int32_t* hist_indices, int hist_row_mode, void* stream) {
    if (!in_keys || R < 1 || n_users_blk <= 0 || K < 1 || K > PDA_MAX_K) return PDA_ERR_ARG;
    if (!out_keys && !out_idx) return PDA_ERR_ARG;
    if (hist_indptr && (!hist_indices || (hist_row_mode == PDA_HIST_BY_USER_ID && !users))) return PDA_ERR_ARG;
    const size_t smem = 4 * ((size_t)R * K + PDA_MAX_K) * sizeof(uint64_t);
    if (smem > 160 * 1024) return PDA_ERR_UNSUPPORTED;
    static int attr_set = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_merge_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return PDA_ERR_LAUNCH;
        attr_set = 1;
    }
    MergeArgs a{in_keys, out_keys, out_idx, out_val, users, hist_indptr, hist_indices, hist_row_mode, R, n_users_blk, K};
    const int blocks = min((n_users_blk + 3) / 4, 256 * 8);
    hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)blocks), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), a);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}
