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
//                16-byte load per 8 k's.  Even and odd k-chunks accumulate in two independent fmaf
//                chains that are added once at the end (oracle order 1 restates exactly this).
//   epilogue   = per accumulator register: head transform ((elu+1)*pop), one v_cmp against the
//                per-user running threshold, wave ballot.  Only when some lane passes (rare after
//                warm-up: ~K/i of the scores at item i) does the wave enter the slow path, which
//                appends (key) candidates to the user's 60-slot LDS list and, when it fills,
//                compacts it with an in-register wave-wide rank sort and raises the threshold.
//   mask       = history CSR rows sorted by item id; each user row keeps a cursor, producing a
//                32-bit mask per (user, tile) that is consulted only on the slow path.
#include "pda_topk_common.h"
#include <cstdlib>

namespace {

using namespace pda_topk;

// ABL: profiling-only ablation bits (0 in the shipped instantiations): 1 skip slow path, 2 skip threshold test,
// 4 skip history cursor, 8 skip item-tile global loads.  Enabled by building with -DPDA_ABLATION.
// BF: the tables are bf16 (pda_score_topk_bf16's exact fallback); rows are widened to fp32 on load -- same arithmetic.
template <int D, int HEAD, int ABL = 0, bool BF = false>
__global__ void __launch_bounds__(kThreads, (D <= 128 ? 2 : 1)) score_topk_kernel(ScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Bt = reinterpret_cast<float*>(smem);                                   // [32][D] swizzled
    uint64_t* lists = reinterpret_cast<uint64_t*>(smem + 32 * D * sizeof(float));  // [128][kCap]

    constexpr int CPR = D / 4;            // 16-B chunks per item row
    constexpr int NLD = (32 * CPR) / kThreads;  // float4 loads / thread / tile (D >= 32)
    static_assert(NLD >= 1, "embed dim too small for the 256-thread staging pattern");
    constexpr int NC = D / 8;             // k-chunks of 8 (even: two accumulator chains)
    static_assert(NC % 2 == 0, "embed dim must be a multiple of 16");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int split = blockIdx.x % a.n_splits, utile = blockIdx.x / a.n_splits;
    const int K = a.K;
    if (a.tile_flags != nullptr && a.tile_flags[utile] == 0) return;   // v2 fallback mode: only flagged user tiles

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
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row_ok) v = pda_load4<BF>(a.U, (size_t)uid * D + 4 * h + 8 * c);
            areg[c] = v;
        }
    }

    // ---- history cursor: head `nxt`, look-ahead `nxt2`, and one refill load in flight (`pend_v`).  All loads of
    // the steady-state loop are unconditional and straight-line so the compiler's vmcnt counting stays exact ---
    int64_t hp = 0, he = 0;
    int nxt = 0x7fffffff, nxt2 = 0x7fffffff, pend_v = 0x7fffffff;
    bool pend_flag = false, pend_ok = false;
    const bool hist_on = (a.hist_indptr != nullptr) && !(ABL & 4);
    if (hist_on && row_ok) {
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
        if (hp + 1 < he) nxt2 = a.hist_indices[hp + 1];
    }

    // ---- per-row running state in LDS: candidate count + threshold (rows are private to their wave) ------
    int* cntl = reinterpret_cast<int*>(lists + (size_t)kUserTile * kCap);   // [128]
    float* taul = reinterpret_cast<float*>(cntl + kUserTile);               // [128]
    if (lane < 32) {
        cntl[wave * 32 + lane] = 0;
        taul[wave * 32 + lane] = row_ok ? -INFINITY : INFINITY;  // rows past the end never accept anything
    }
    pda_wave_sync();
    f32x16 thr;
    auto refresh_thr = [&]() {
        int hv = h;
        asm volatile("" : "+v"(hv));   // opaque: keeps the 16 LDS addresses from being hoisted into live registers
#pragma unroll
        for (int r = 0; r < 16; ++r) thr[r] = taul[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hv];
    };
    refresh_thr();

    // ---- item tile staging -------------------------------------------------------------------
    f32x4 pre[NLD];
    // Unconditional loads (row index clamped, never predicated): a load hidden behind a branch cannot be counted
    // by the compiler's vmcnt bookkeeping and turns every later wait into a full drain.  Rows past the end of the
    // shard produce garbage scores that `vmask` discards.
    auto tile_load = [&](int t) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int id = tid + kThreads * q;
            const int jj = id / CPR, ch = id % CPR;
            const int it = min(t * 32 + jj, a.n_items_local - 1);
            pre[q] = pda_load4<BF>(a.I, (size_t)it * D + 4 * ch);
        }
    };
    auto tile_store = [&]() {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int id = tid + kThreads * q;
            const int jj = id / CPR, ch = id % CPR;
            *reinterpret_cast<f32x4*>(Bt + jj * D + 4 * (ch ^ swz<D>(jj))) = pre[q];
        }
    };
    auto pop_load = [&](int t) -> float {
        if constexpr (HEAD == PDA_HEAD_POP) return a.pop[min(t * 32 + j, a.n_items_local - 1)];
        return 1.0f;
    };
    // History bits of tile t for my row.  Branch-free for the common case (at most one train item of the row
    // inside the tile): commit the refill issued by the previous call, advance once, issue the next refill.
    auto hist_bits = [&](int t) -> uint32_t {
        if (!hist_on) return 0u;                                   // wave-uniform
        const int jg0 = a.item_offset + t * 32, jg1 = jg0 + 32;
        nxt2 = pend_flag ? (pend_ok ? pend_v : 0x7fffffff) : nxt2; // value loaded one tile ago: no stall
        const bool adv = nxt < jg1;
        uint32_t hb = adv ? (1u << ((nxt - jg0) & 31)) : 0u;
        hp += adv ? 1 : 0;
        nxt = adv ? nxt2 : nxt;
        const int64_t idx = hp + 1;
        pend_ok = idx < he;
        pend_flag = adv;
        // never predicated (exact vmcnt bookkeeping), but lanes that did not advance all read element 0: one cache
        // line for the wave instead of a 64-line gather on every tile
        const int64_t idc = adv ? max((int64_t)0, min(idx, he - 1)) : (int64_t)0;
        pend_v = a.hist_indices[idc];
        if (__builtin_expect(__any(nxt < jg1), 0)) {               // rare: several train items in one tile
            do {
                if (nxt < jg1) {
                    const int nn2 = pend_flag ? (pend_ok ? pend_v : 0x7fffffff) : nxt2;
                    hb |= 1u << ((nxt - jg0) & 31);
                    ++hp;
                    nxt = nn2;
                    nxt2 = (hp + 1 < he) ? a.hist_indices[hp + 1] : 0x7fffffff;
                    pend_flag = false;
                }
            } while (__any(nxt < jg1));
        }
        return hb;
    };
    auto valid_mask = [&](int t) -> uint64_t {
        const int nvalid = min(32, a.n_items_local - t * 32);
        return nvalid >= 32 ? ~0ull : (((1ull << nvalid) - 1ull) * 0x100000001ull);
    };

    uint64_t* my_lists = lists + (size_t)(wave * 32) * kCap;
    const float* brow = Bt + j * D;
    const int bswz = swz<D>(j);

    // Slow path for one finished tile.  `regmask`: accumulator registers in which some lane beat its threshold.
    // Every passing lane appends its own candidate with an LDS atomic on the row's counter -- no serialisation
    // over lanes or rows; a full list is compacted lazily and only the failed lanes retry.  The loop over
    // registers is a run-time loop (dynamic VGPR index) on purpose: unrolled 16x it costs ~70 live registers.
    auto slow_path = [&](uint32_t regmask, const f32x16& accv, float popv, uint64_t vmask, uint32_t hb, int jg0) {
        const uint32_t my_item = (uint32_t)(jg0 + j);
        const bool lane_ok = (vmask >> lane) & 1ull;
        const bool any_hb = __any(hb != 0);   // some row of this wave has train items inside the tile
        uint32_t still = 0xFFFFu;             // per-lane: registers this lane may (still) append
        uint32_t rm = regmask;
        for (;;) {
            uint32_t ovf_regs = 0, next_still = 0;
#pragma nounroll
            while (rm) {
                const int r = __builtin_ctz(rm);
                rm &= rm - 1u;
                float tt = accv[r];   // exact head value (the fast test only saw an upper bound)
                if constexpr (HEAD == PDA_HEAD_POP) tt = (tt > 0.0f ? tt + 1.0f : __expf(tt)) * popv;
                bool p = lane_ok && ((still >> r) & 1u) && (tt > thr[r]);
                const int rowb = (r & 3) + 8 * (r >> 2);
                if (any_hb) {   // train items never enter
                    const uint32_t h0 = (uint32_t)__builtin_amdgcn_readlane((int)hb, rowb);
                    const uint32_t h1 = (uint32_t)__builtin_amdgcn_readlane((int)hb, rowb + 4);
                    if (((h ? h1 : h0) >> j) & 1u) p = false;
                }
                bool ov = false;
                if (p) {
                    const int row = wave * 32 + rowb + 4 * h;
                    const int slot = atomicAdd(&cntl[row], 1);   // ds_add_rtn_u32: distinct slots for concurrent lanes
                    if (slot < kCap) lists[(size_t)row * kCap + slot] = pda_pack_key(tt, my_item);
                    else ov = true;
                }
                if (ov) next_still |= 1u << r;
                if (__any(ov)) ovf_regs |= 1u << r;
            }
            if (!ovf_regs) break;
            // some list overflowed: compact every full list of this wave, then retry what is still above threshold
            pda_wave_sync();
            uint64_t full = __ballot(lane < 32 && cntl[wave * 32 + (lane & 31)] >= kCap);
            while (full) {
                const int row = __builtin_ctzll(full);
                full &= full - 1ull;
                compact_list(my_lists + row * kCap, &cntl[wave * 32 + row], &taul[wave * 32 + row], K, lane);
            }
            refresh_thr();
            still = next_still;
            rm = ovf_regs;
        }
    };

    // Threshold test on an UPPER BOUND of the head: ub = (max(s,0)+1)*pop equals the exact (elu(s)+1)*pop for
    // s > 0 (bitwise) and is >= it for s <= 0, so no candidate is missed and the exp is only paid on the slow path.
    auto head_ub = [&](float sc, float popv) -> float {
        if constexpr (HEAD == PDA_HEAD_POP) return (fmaxf(sc, 0.0f) + 1.0f) * popv;
        return sc;
    };
    auto fast_test = [&](const f32x16& accv, float popv, uint64_t vmask) -> uint32_t {
        uint32_t regmask = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            regmask |= (__ballot(head_ub(accv[r], popv) > thr[r]) & vmask) ? (1u << r) : 0u;
        return regmask;
    };

    // Software pipeline.  Iteration t:  issue the global loads of tile t+1;  MFMA chain of tile t with the
    // threshold test of tile t-1 interleaved in the matrix instructions' shadow;  barrier;  stage tile t+1 into
    // LDS (the only vmcnt wait of the iteration);  advance the history cursor to tile t+1;  slow path of t-1.
    f32x16 acc_prev = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint64_t vmask_prev = 0, vmask_cur = 0;
    uint32_t hb_prev = 0, hb_cur = 0;
    float popj_prev = 0.f, popj_cur = 0.f;

    if (t0 < t1) {
        tile_load(t0);
        popj_cur = pop_load(t0);
        tile_store();
        hb_cur = hist_bits(t0);
        vmask_cur = valid_mask(t0);
    }
    __syncthreads();

    for (int t = t0; t < t1; ++t) {
        const bool has_next = (t + 1) < t1;
        const int tn = has_next ? t + 1 : t;          // last iteration re-loads its own tile: keeps the loads unconditional
        float popj_next = 0.f;
        if constexpr (!(ABL & 8)) {
            tile_load(tn);
            popj_next = pop_load(tn);
        }
        __builtin_amdgcn_sched_barrier(0);   // pin the prefetch ahead of the MFMA chain (the scheduler otherwise sinks it)

        // ---- contraction of tile t on the matrix cores  ||  threshold test of tile t-1 ------------
        // Two accumulator chains (even / odd k-chunks): a dependent v_mfma chain loses its SrcC forwarding (+43
        // cycles) as soon as anything is issued between two links, while instructions between MFMAs on DIFFERENT
        // accumulators are nearly free -- so the test of the previous tile is interleaved here.
        f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 acc1 = acc0;
        uint32_t regmask = 0;
#pragma unroll
        for (int c = 0; c < NC; c += 2) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(brow + 4 * ((2 * c + h) ^ bswz));
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(brow + 4 * ((2 * c + 2 + h) ^ bswz));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[c][q], b0[q], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[c + 1][q], b1[q], acc1, 0, 0, 0);
            }
            if constexpr (!(ABL & 2)) {
#pragma unroll
                for (int r = (16 * c) / NC; r < (16 * (c + 2)) / NC; ++r)
                    regmask |= (__ballot(head_ub(acc_prev[r], popj_prev) > thr[r]) & vmask_prev) ? (1u << r) : 0u;
            }
        }
        if constexpr (ABL & 2) asm volatile("" ::"v"(acc_prev[0]), "v"(acc_prev[7]), "v"(acc_prev[15]));
        // Scheduling recipe for the block above: B fragments ahead of their MFMAs, the test spread over the gaps.
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (HEAD == PDA_HEAD_POP ? 64 : 16) / (4 * NC) + 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x004, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        const f32x16 acc = acc0 + acc1;

        __syncthreads();  // every wave is done reading Bt
        uint32_t hb_next = 0;
        if (has_next) {
            tile_store();
            hb_next = hist_bits(t + 1);
        }
        if constexpr (ABL & 1) {
            asm volatile("" ::"s"(regmask));
        } else {
            if (regmask) slow_path(regmask, acc_prev, popj_prev, vmask_prev, hb_prev, a.item_offset + (t - 1) * 32);
        }
        __syncthreads();  // next tile visible in Bt

        acc_prev = acc;
        vmask_prev = vmask_cur;
        hb_prev = hb_cur;
        popj_prev = popj_cur;
        vmask_cur = has_next ? valid_mask(t + 1) : 0ull;
        hb_cur = hb_next;
        popj_cur = popj_next;
    }
    if (t0 < t1) {   // drain: threshold test + slow path of the last tile
        const uint32_t regmask = fast_test(acc_prev, popj_prev, vmask_prev);
        if (!(ABL & 1) && regmask) slow_path(regmask, acc_prev, popj_prev, vmask_prev, hb_prev, a.item_offset + (t1 - 1) * 32);
    }

    // ---- finalise: sort every row's list, emit K packed keys (0 = empty) -------------------------
    for (int rr = 0; rr < 32; ++rr) {
        uint64_t* buf = my_lists + rr * kCap;
        compact_list(buf, &cntl[wave * 32 + rr], &taul[wave * 32 + rr], K, lane);  // count <- min(count, K), buf sorted
        const int c = cntl[wave * 32 + rr];
        const int rb = utile * kUserTile + wave * 32 + rr;
        if (rb < a.n_users_blk && lane < K) {
            const uint64_t k = lane < c ? buf[lane] : 0ull;
            a.out_keys[((size_t)split * a.n_users_blk + rb) * K + lane] = k;
        }
    }
}

template <int D, int HEAD, int ABL = 0, bool BF = false>
int launch_score(const ScoreArgs& a, hipStream_t stream) {
    const size_t smem = 32 * D * sizeof(float) + (size_t)kUserTile * (kCap * sizeof(uint64_t) + 8);
    static int attr_set = 0;  // idempotent attribute; benign if raced
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&score_topk_kernel<D, HEAD, ABL, BF>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return PDA_ERR_LAUNCH;
        attr_set = 1;
    }
    const int utiles = (a.n_users_blk + kUserTile - 1) / kUserTile;
    dim3 grid((unsigned)(utiles * a.n_splits));
    hipLaunchKernelGGL((score_topk_kernel<D, HEAD, ABL, BF>), grid, dim3(kThreads), smem, stream, a);
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
    // per wave: the R lists | the merged list | the lists' lengths (lists behind a shared warm-up are all but empty: they are skipped)
    const size_t per_wave = (size_t)(n + PDA_MAX_K) * 8 + (((size_t)a.R * 4 + 7) & ~(size_t)7);
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem + (size_t)wave * per_wave);
    uint64_t* outl = keys + n;
    int* lens = reinterpret_cast<int*>(outl + PDA_MAX_K);
    const int waves_total = gridDim.x * 4;
    for (int u = blockIdx.x * 4 + wave; u < a.n_users_blk; u += waves_total) {
        pda_wave_sync();
        for (int e = lane; e < n; e += 64) {
            const int r = e / K, p = e - r * K;
            keys[e] = a.in_keys[((size_t)r * a.n_users_blk + u) * K + p];
        }
        if (lane < K) outl[lane] = 0ull;
        pda_wave_sync();
        // the largest K-th key of any list bounds the answer from below: K keys of that list are at least as large, so a smaller
        // key has rank >= K and needs no search (32 full lists: 50 + a few survivors of 1 600 keys)
        uint64_t floor_key = 0ull;
        for (int r = lane; r < a.R; r += 64) {
            floor_key = max(floor_key, keys[r * K + K - 1]);
            int lo = 0, hi = K;                      // the list's length: its first empty slot (sorted, zeros behind the keys)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (keys[r * K + mid] != 0ull) lo = mid + 1; else hi = mid;
            }
            lens[r] = lo;
        }
        pda_wave_sync();
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) floor_key = max(floor_key, (uint64_t)__shfl_xor((unsigned long long)floor_key, o, 64));
        for (int e = lane; e < n; e += 64) {
            const uint64_t key = keys[e];
            if (key == 0ull || key < floor_key) continue;
            const int r = e / K, p = e - r * K;
            int rank = p;  // own list is sorted and keys are unique
            for (int r2 = 0; r2 < a.R; ++r2) {
                if (r2 == r) continue;
                const uint64_t* l = keys + r2 * K;
                int lo = 0, hi = lens[r2];  // first position whose key is <= mine (descending list)
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

int pda_topk::launch_score_v1(const ScoreArgs& a, int d, int head, hipStream_t s, bool bf16) {
    if (bf16) {
#define PDA_DISPATCH_BF(DD)                                                             \
    case DD:                                                                            \
        return head == PDA_HEAD_POP ? launch_score<DD, PDA_HEAD_POP, 0, true>(a, s)    \
                                    : launch_score<DD, PDA_HEAD_RAW, 0, true>(a, s);
        switch (d) {
            PDA_DISPATCH_BF(64)
            PDA_DISPATCH_BF(128)
            PDA_DISPATCH_BF(256)
            default:
                return PDA_ERR_UNSUPPORTED;
        }
#undef PDA_DISPATCH_BF
    }
#ifdef PDA_ABLATION
    if (const char* e = getenv("PDA_ABLATE")) {
        if (d == 128 && head == PDA_HEAD_POP && a.tile_flags == nullptr) switch (atoi(e)) {
            case 1: return launch_score<128, PDA_HEAD_POP, 1>(a, s);
            case 3: return launch_score<128, PDA_HEAD_POP, 3>(a, s);
            case 7: return launch_score<128, PDA_HEAD_POP, 7>(a, s);
            case 15: return launch_score<128, PDA_HEAD_POP, 15>(a, s);
            case 4: return launch_score<128, PDA_HEAD_POP, 4>(a, s);
            default: break;
        }
    }
#endif
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
                n_users_blk, item_offset, n_items_local, hist_row_mode, K, n_splits, nullptr};
    return pda_topk::launch_score_v1(a, d, head, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pda_topk_merge(const uint64_t* in_keys, int R, int n_users_blk, int K, uint64_t* out_keys,
                              int32_t* out_idx, float* out_val, const int32_t* users, const int64_t* hist_indptr,
                              const int32_t* hist_indices, int hist_row_mode, void* stream) {
    if (!in_keys || R < 1 || n_users_blk <= 0 || K < 1 || K > PDA_MAX_K) return PDA_ERR_ARG;
    if (!out_keys && !out_idx) return PDA_ERR_ARG;
    if (hist_indptr && (!hist_indices || (hist_row_mode == PDA_HIST_BY_USER_ID && !users))) return PDA_ERR_ARG;
    const size_t smem = 4 * (((size_t)R * K + PDA_MAX_K) * sizeof(uint64_t) + (((size_t)R * 4 + 7) & ~(size_t)7));
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
