// score + mask + top-K, generation 3: ONE bf16 MFMA per k-step as pre-filter + candidate ring + exact fp32 rescoring.
// Same packed keys as v1 / v2, bit for bit (tests/test_gpu_score_topk.py runs every case through all of them).
//
// v2 spends 3 bf16 MFMAs per k-step (hi/lo split of both operands) because its per-user lists are keyed by the
// APPROXIMATE head and have to stay within a narrow band.  Measured (C3, visiting order): the pure MFMA loop is 6.2 ms
// of the 10.0 ms, at a power-throttled 1.8 GHz -- the matrix pipe, not the epilogue, is what is left to cut.  v3 uses
//     s~ = bf16(u) . bf16(i)                    one v_mfma_f32_32x32x16_bf16 per 16 k:  3x less pipe time again
//     |s~ - s_exact| <= eps(u,i) = 2^-7 (1.01) ||u|| ||i||       (fp32 tables; bf16 tables: 2^-14, only the add order differs)
// as a FILTER only: pairs whose head upper bound beats the user's exact running threshold go into a per-wave LDS ring
// (4 bytes each: row, item id) and are rescored with the exact fp32 fmaf chain of v1 when the ring fills -- 16 candidates
// per pass, four lanes per candidate, all four waves of the workgroup draining in the same iteration.  The per-user
// lists therefore hold EXACT keys: thresholds are exact, there is no band, no end-of-sweep rescoring and no fallback
// kernel.  The coarser eps lets ~15 % more candidates through than an exact test would; what makes the design pay is the
// visiting order (popular first: a few hundred candidates per user instead of K ln(I/K)).
//
// Error bound, fp32 tables: u_k = uh_k + du_k, |du_k| <= 2^-8 |u_k| (RNE to 8 significant bits: the spacing of bf16 in [1, 2) is 2^-7, the
// unit roundoff HALF of it -- 1 + 2^-8 rounds to 1; rounds 1 - 4 used 2^-9, half the worst case: dense rows never showed it, rows with one
// or a few non-zero elements can, tests/test_gpu_score_topk.py::test_filter_bound_worst_case_rounding), same for i:
//   u_k i_k - uh_k ih_k = du_k i_k + uh_k di_k,  |.| <= 2^-8 (2 + 2^-8) |u_k i_k|;  summed, Cauchy-Schwarz: 2^-7 (1 + 2^-9) ||u|| ||i||.
//   bf16 products are exact in fp32; fp32 accumulation of d terms (any order) and the rounding of the exact chain add
//   <= 2 d 2^-24 sum|u_k i_k| <= 2^-15 ||u|| ||i|| for d <= 256.  Norms are padded by (1+2^-10)(1+1e-4).  Used: 2^-7 * 1.01.
#include "pda_topk_common.h"
#include <cstdlib>
#ifndef PDA_KWARM
#define PDA_KWARM 4
#endif
#ifndef PDA_KWARM_NAT
#define PDA_KWARM_NAT 16
#endif
#ifndef PDA_VOTE
#define PDA_VOTE 3
#endif

using namespace pda_topk;

namespace {

constexpr int kCap3 = PDA_TOPK_CAP - 1;   // 59 slots per user list
constexpr int kRing = 192;                // ring entries per wave (u32 each); a push needs kRing - 64 free
constexpr int kRingTrig = 64;             // a wave above this asks the whole workgroup to drain

__device__ __forceinline__ f32x16 zero16w() {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return z;
}

template <int D, int HEAD, bool ORD, bool BF>
__global__ void __launch_bounds__(kThreads, 2) score_topk_v3_kernel(ScoreArgs2 aa) {
    const ScoreArgs& a = aa.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr float kEps = BF ? 6.103515625e-5f : 7.890625e-3f;   // 2^-14  |  2^-7 * 1.01
    constexpr int CPR = D / 8;                 // 16-byte chunks per bf16 row
    constexpr int NM = D / 16;                 // MFMA k-steps
    // Item tile = NB blocks of 32 columns.  The loop is instruction-issue bound (PMC: ~260 instructions per wave and 32
    // items around 8 MFMAs), and roughly 60 % of them do not depend on the tile width (history cursor, staging addresses,
    // barriers, drain / stop flags, loop control): two blocks per tile halve that share.  d = 256: one block (LDS).
    // Natural item order with the VALU test keeps one block: there the candidate-rich slow path dominates and the staler
    // thresholds of a wider tile cost more than the bookkeeping saves (measured 12.2 vs 11.5 ms at C3); with the folded test
    // (raw head) the wider tile wins again: 11.2 -> 10.1 ms.
    constexpr int NB = D <= 128 ? 2 : 1;
    constexpr int TW = 32 * NB;
    // Folded test: one extra MFMA k-step subtracts  thr / pop - 1 - eps  inside the matrix pipe (bf16 pieces prepared per
    // item in I_bex, per row in `aex`), so that "candidate" is "accumulator > 0" and the 16 rows of a lane reduce with
    // v_max3 before a single compare: 9 VALU per 32x32 block instead of 48 (max, fma, cmp per register).  The item side needs
    // 1/pop: ordered preps carry it; natural-order PDA-head sweeps rebuild the pieces per tile (~15 VALU per block).
    constexpr bool kHistAtCand = ORD;   // where the train-item mask is applied: see process_ring
    constexpr int NLD = (TW * CPR) / kThreads; // 16-byte loads per thread per tile
    static_assert(NLD >= 1, "v3 needs embed dim >= 64");
    uint16_t* Bh = reinterpret_cast<uint16_t*>(smem);                                       // [TW][D] bf16, swizzled
    constexpr int kTileBytes = D <= 128 ? 64 * D * 2 : 32 * D * 2;   // d <= 128: room for the fp32 warm-up block (32 x D x 4)
    uint64_t* lists = reinterpret_cast<uint64_t*>(smem + kTileBytes);                       // [128][kCap3] EXACT keys
    int* cntl = reinterpret_cast<int*>(lists + (size_t)kUserTile * kCap3);                  // [128]
    float* taul = reinterpret_cast<float*>(cntl + kUserTile);                               // [128] exact K-th value (-inf until K entries)
    uint32_t* rings = reinterpret_cast<uint32_t*>(taul + kUserTile);                        // [4][kRing]
    int* wgflag = reinterpret_cast<int*>(rings + 4 * kRing);                                // [2] "some wave wants to drain its ring"
    int* votes = wgflag + 2;                                                                // [4] ORD: wave w sees no use in going on

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int split = blockIdx.x % a.n_splits, utile = blockIdx.x / a.n_splits;
    const int K = a.K;
    const int tiles_total = (a.n_items_local + TW - 1) / TW;
    int t0, stride, nt;                        // this workgroup's tiles: t0, t0 + stride, ... (nt of them)
    if constexpr (ORD) {
        t0 = split;
        stride = a.n_splits;
        nt = t0 < tiles_total ? (tiles_total - t0 + stride - 1) / stride : 0;
    } else {
        const int tiles_per = (tiles_total + a.n_splits - 1) / a.n_splits;
        t0 = split * tiles_per;
        stride = 1;
        nt = max(0, min(t0 + tiles_per, tiles_total) - t0);
    }
    auto tile_of = [&](int k) __attribute__((always_inline)) { return t0 + k * stride; };   // (every lambda is force-inlined: a closure
    // that the inliner leaves out of line lives in scratch together with everything it captures -- measured 3x slower)

    const int row_blk = utile * kUserTile + wave * 32 + j;
    const bool row_ok = row_blk < a.n_users_blk;
    const int uid = row_ok ? a.users[row_blk] : 0;

    // ---- A operand: this lane's user row (k = 16m + 8h .. +7) rounded to bf16; padded row norm --------------------
    u32x4 ah[NM];
    float nu_row;
    {
        float ss = 0.f;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = x;
            if (row_ok) {
                x = pda_load4<BF>(a.U, (size_t)uid * D + 8 * h + 16 * m);
                y = pda_load4<BF>(a.U, (size_t)uid * D + 8 * h + 16 * m + 4);
            }
            u32x4 lo_unused;
            split8(x, y, ah[m], lo_unused);    // bf16 tables: the RNE of a bf16 value is itself
#pragma unroll
            for (int k = 0; k < 4; ++k) ss += x[k] * x[k] + y[k] * y[k];
        }
        ss += __shfl_xor(ss, 32, 64);
        nu_row = sqrtf(ss) * 1.0009765625f * 1.0001f;              // padded ||u||
    }

    // ---- history cursor (as v1 / v2) --------------------------------------------------------------------------------
    int64_t hp = 0, he = 0, hbeg = 0;
    int nxt = 0x7fffffff, nxt2 = 0x7fffffff, pend_v = 0x7fffffff;
    bool pend_flag = false, pend_ok = false;
    const bool hist_on = a.hist_indptr != nullptr;
    if (hist_on && row_ok) {
        const int64_t hr = a.hist_row_mode == PDA_HIST_BY_USER_ID ? (int64_t)uid : (int64_t)row_blk;
        hp = a.hist_indptr[hr];
        he = a.hist_indptr[hr + 1];
        hbeg = hp;
        const int lo_item = a.item_offset + t0 * TW;
        int64_t lo = hp, hi = he;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (a.hist_indices[mid] < lo_item) lo = mid + 1; else hi = mid;
        }
        hp = lo;
        if (hp < he) nxt = a.hist_indices[hp];
        if (hp + 1 < he) nxt2 = a.hist_indices[hp + 1];
    }
    auto hist_bits = [&](int t) __attribute__((always_inline)) -> uint64_t {
        if (!hist_on) return 0ull;
        const int jg0 = a.item_offset + t * TW, jg1 = jg0 + TW;
        // most tiles hold no history entry of any of the wave's 32 rows (always so a few tiles into an ordered sweep): leave
        // before the cursor bookkeeping.  A refill still in flight stays pending (flag and register untouched).  The loads
        // below become conditional, which costs nothing here: the tile's own loads were consumed before this call.
        if (!__any(nxt < jg1)) return 0ull;
        nxt2 = pend_flag ? (pend_ok ? pend_v : 0x7fffffff) : nxt2;
        const bool adv = nxt < jg1;
        uint64_t hb = (adv && (!ORD || nxt >= jg0)) ? (1ull << ((nxt - jg0) & (TW - 1))) : 0ull;
        hp += adv ? 1 : 0;
        nxt = adv ? nxt2 : nxt;
        const int64_t idx = hp + 1;
        pend_ok = idx < he;
        pend_flag = adv;
        // never predicated (exact vmcnt bookkeeping); lanes that did not advance all read element 0: one cache line
        const int64_t idc = adv ? max((int64_t)0, min(idx, he - 1)) : (int64_t)0;
        pend_v = a.hist_indices[idc];
        if (__builtin_expect(__any(nxt < jg1), 0)) {
            do {
                if (nxt < jg1) {
                    const int nn2 = pend_flag ? (pend_ok ? pend_v : 0x7fffffff) : nxt2;
                    if (!ORD || nxt >= jg0) hb |= 1ull << ((nxt - jg0) & (TW - 1));
                    ++hp;
                    nxt = nn2;
                    nxt2 = (hp + 1 < he) ? a.hist_indices[hp + 1] : 0x7fffffff;
                    pend_flag = false;
                }
            } while (__any(nxt < jg1));
        }
        return hb;
    };

    // ---- per-row state in LDS -----------------------------------------------------------------------------------------
    if (lane < 32) {
        cntl[wave * 32 + lane] = 0;
        taul[wave * 32 + lane] = row_ok ? -INFINITY : INFINITY;
    }
    if (tid < 2) wgflag[tid] = 0;
    if (tid < 4) votes[tid] = 0;
    pda_wave_sync();
    float nu_max = nu_row;                     // ONE norm per wave (the largest): eps scale of the filter, and the
#pragma unroll                                 // termination bound of the ordered sweep
    for (int o = 32; o > 0; o >>= 1) nu_max = fmaxf(nu_max, __shfl_xor(nu_max, o, 64));
    nu_max *= kEps * 1.001f;
    // lowered threshold of row (r in accumulator layout of lane half hv); strictly below tau (also for tau == 0): an item
    // that TIES the K-th value must get through -- in visiting order it may carry the lower id and win
    auto thr_of = [&](int r, int hv) __attribute__((always_inline)) -> float {
        const float tq = taul[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hv];
        return (tq == INFINITY || tq == -INFINITY) ? tq : tq - fabsf(tq) * 9.5367431640625e-7f - 1e-30f;
    };
    // The threshold of the lane's OWN row j (finite: +-1e30 stand for +-inf), lowered by 2^-16 relative -- the last
    // MFMA adds 11 products of magnitude up to |thr / pop| + 1 + eps to s~ in fp32 (<= 17 roundings of 2^-23 at that
    // magnitude: 2^-18.9; the constant slot carries +5e-6, the eps slot +8 %) -- the wave's minimum, and the A operand of the
    // extra k-step:  k 0..7 (lanes < 32): -(t1,t1,t2,t2,t1,t3,t2,t3), thr = t1 + t2 + t3 exactly;  k 8..10: +1, +1, +eps scale.
    float thr_own = 0.f, thr_min = 0.f;
    u32x4 aex = {0u, 0u, 0u, 0u};
    auto refresh_thr = [&]() __attribute__((always_inline)) {
        {
            const float tq = taul[wave * 32 + j];
            float tf = (tq == INFINITY || tq == -INFINITY) ? tq : tq - fabsf(tq) * 1.52587890625e-5f - 1e-30f;
            tf = fminf(fmaxf(tf, -1.0e30f), 1.0e30f);
            thr_own = tf;
            float m = tf;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
            thr_min = m;
            uint32_t t1, t2, t3;
            bf16_split3(tf, t1, t2, t3);
            t1 ^= 0x8000u;
            t2 ^= 0x8000u;
            t3 ^= 0x8000u;
            const uint32_t nnu = bf16_up(nu_max * 1.08f);
            aex[0] = h ? 0x3F803F80u : (t1 | (t1 << 16));
            aex[1] = h ? nnu : (t2 | (t2 << 16));
            aex[2] = h ? 0u : (t1 | (t3 << 16));
            aex[3] = h ? 0u : (t2 | (t3 << 16));
        }
    };
    refresh_thr();

    uint64_t* my_lists = lists + (size_t)(wave * 32) * kCap3;
    uint32_t* ring = rings + wave * kRing;
    int ring_cnt = 0;   // wave-uniform
    unsigned n_cand = 0;   // statistics: pairs this wave rescored exactly

    // ---- item tile staging ---------------------------------------------------------------------------------------------
    u32x4 pA_h[NLD];
    auto tile_load = [&](int t, u32x4 (&ph)[NLD]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int id = tid + kThreads * q;
            const int jj = id / CPR, ch = id % CPR;
            const uint32_t it = (uint32_t)min(t * TW + jj, a.n_items_local - 1);
            ph[q] = *reinterpret_cast<const u32x4*>(aa.I_hi + (it * (uint32_t)D + 8u * (uint32_t)ch));   // 32-bit element offset
        }
    };
    auto tile_store = [&](const u32x4 (&ph)[NLD]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int id = tid + kThreads * q;
            const int jj = id / CPR, ch = id % CPR;
            *reinterpret_cast<u32x4*>(Bh + jj * D + 8 * (ch ^ swzb<D>(jj))) = ph[q];
        }
    };
    auto bex_load = [&](int t, u32x4 (&bx)[NB]) __attribute__((always_inline)) {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const uint32_t it = (uint32_t)min(t * TW + 32 * cb + j, a.n_items_local - 1);
            bx[cb] = *reinterpret_cast<const u32x4*>(aa.I_bex + (it * 16u + 8u * (uint32_t)h));
        }
    };
    auto lane_consts = [&](int t, float (&popv)[NB], int (&idv)[NB]) __attribute__((always_inline)) {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const int it = min(t * TW + 32 * cb + j, a.n_items_local - 1);
            popv[cb] = 1.0f;
            if constexpr (HEAD == PDA_HEAD_POP) popv[cb] = ORD ? aa.pop_p[it] : a.pop[it];
            if constexpr (ORD) idv[cb] = a.item_offset + aa.order[it];      // the ring keeps the item's real id
            else idv[cb] = a.item_offset + t * TW + 32 * cb + j;
        }
    };

    // ---- exact rescoring of the ring: D/32 lanes per candidate, 64/(D/32) candidates per pass -------------------------
    // Lane LPC*ci+q loads k = 32q .. 32q+31 of candidate ci's user row and item row (contiguous across the lanes of a
    // candidate), and the two fmaf chains of v1 (even / odd k-chunks, k ascending) are carried from lane to lane: phase
    // p completes the 32 k's of lane p and hands the accumulators to lane p+1.  The last lane ends up with the bit-exact
    // v1 score and appends to the row's list.
    auto process_ring = [&]() __attribute__((always_inline)) {
        constexpr int LPC = D / 32;                 // lanes per candidate: each owns 32 consecutive k (4 chunks of 8)
        constexpr int CPP = 64 / LPC;               // candidates per pass
        const int q = lane % LPC, ci = lane / LPC;
        n_cand += (unsigned)ring_cnt;
        for (int base = 0; base < ring_cnt; base += CPP) {
            const int e = base + ci;
            const bool valid = e < ring_cnt;
            const uint32_t word = valid ? ring[e] : 0u;
            const int row = (int)(word >> 27);
            const int item = valid ? (int)(word & 0x7FFFFFFu) : a.item_offset;     // global item id
            const int urow = __shfl(uid, row, 64);
            const size_t ub = (size_t)urow * D + q * 32, ib = (size_t)(item - a.item_offset) * D + q * 32;
            f32x4 uu[8], ii[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                uu[c] = pda_load4<BF>(a.U, ub + 4 * c);
                ii[c] = pda_load4<BF>(a.I, ib + 4 * c);
            }
            float pv = 1.0f;
            if constexpr (HEAD == PDA_HEAD_POP) pv = a.pop[item - a.item_offset];
            float c0 = 0.f, c1 = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
            for (int ph = 0; ph < LPC; ++ph) {
                o0 = c0;
                o1 = c1;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {      // chunk 4 q + cc of the row: even chunks feed chain 0, odd ones chain 1
#pragma unroll
                    for (int sidx = 0; sidx < 4; ++sidx) {
                        if (cc & 1) {
                            o1 = __builtin_fmaf(uu[2 * cc][sidx], ii[2 * cc][sidx], o1);
                            o1 = __builtin_fmaf(uu[2 * cc + 1][sidx], ii[2 * cc + 1][sidx], o1);
                        } else {
                            o0 = __builtin_fmaf(uu[2 * cc][sidx], ii[2 * cc][sidx], o0);
                            o0 = __builtin_fmaf(uu[2 * cc + 1][sidx], ii[2 * cc + 1][sidx], o0);
                        }
                    }
                }
                if (ph < LPC - 1) {
                    const float r0 = __shfl_up(o0, 1, 64), r1 = __shfl_up(o1, 1, 64);
                    if (q == ph + 1) {
                        c0 = r0;
                        c1 = r1;
                    }
                }
            }
            float sc = o0 + o1;                               // meaningful on the candidate's last lane
            if constexpr (HEAD == PDA_HEAD_POP) sc = (sc > 0.0f ? sc + 1.0f : __expf(sc)) * pv;
            const float tt = (valid && q == LPC - 1) ? sc : -INFINITY;
            const int lrow = wave * 32 + row;
            // ">=": equal scores are decided by the key (lower item id wins) at the next compaction, so ties must get in
            bool p = valid && q == LPC - 1 && (tt >= taul[lrow]);
            if (kHistAtCand && hist_on) {
                // Ordered sweeps mask train items HERE, not in the sweep: behind the warm-up tiles a history entry matters only
                // if it passes the filter and the exact threshold, and then one binary search in the row's (id-sorted) history
                // settles it.  The per-tile history cursor cost 10 % of the sweep for a handful of hits per user.  (Natural
                // order keeps the cursor: hundreds of accepted candidates per user, each search a chain of dependent loads --
                // measured 9.8 -> 11.1 ms.)
                int64_t lo = __shfl(hbeg, row, 64), hi = __shfl(he, row, 64);
                const int64_t hend = hi;
                if (p) {
                    while (lo < hi) {
                        const int64_t mid = (lo + hi) >> 1;
                        if (aa.hist_nat[mid] < item) lo = mid + 1; else hi = mid;
                    }
                    if (lo < hend && aa.hist_nat[lo] == item) p = false;
                }
            }
            const uint64_t key = pda_pack_key(tt, (uint32_t)item);
            for (;;) {
                bool ov = false;
                if (p) {
                    const int slot = atomicAdd(&cntl[lrow], 1);
                    if (slot < kCap3) lists[(size_t)lrow * kCap3 + slot] = key;
                    else ov = true;
                }
                if (!__any(ov)) break;
                pda_wave_sync();
                uint64_t full = __ballot(lane < 32 && cntl[wave * 32 + (lane & 31)] >= kCap3);
                while (full) {
                    const int rr = __builtin_ctzll(full);
                    full &= full - 1ull;
                    compact_list<kCap3>(my_lists + rr * kCap3, &cntl[wave * 32 + rr], &taul[wave * 32 + rr], K, lane);
                }
                p = ov && (tt >= taul[lrow]);
            }
        }
        ring_cnt = 0;
        pda_wave_sync();
        refresh_thr();
    };

    // ---- push the flagged lanes of the previous tile into the ring ------------------------------------------------------
    // m: the lane's flagged accumulator registers, bit 16 cb + 15 - r  <->  register r of column block cb.
    // Lane-parallel: every flagged lane pushes ITS OWN top flagged register per round (usually one round).
    auto push_masks = [&](uint32_t m, uint64_t hb, const int (&item_id)[NB]) __attribute__((always_inline)) {
        const bool any_hb = __any(hb != 0ull);
        while (__any(m != 0)) {
            const bool act = m != 0;
            const int bit = 31 - __builtin_clz(m | 1u);
            const int cb = bit >> 4, r = 15 - (bit & 15);
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            m &= ~(1u << bit);
            bool p = act;
            if (any_hb) {                                                           // train items never enter
                const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)hb, row, 64), hi = (uint32_t)__shfl((int)(uint32_t)(hb >> 32), row, 64);
                if (((cb ? hi : lo) >> j) & 1u) p = false;
            }
            const uint64_t pm = __ballot(p);
            if (!pm) continue;
            if (ring_cnt + 64 > kRing) process_ring();
            const int slot = ring_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0));
            int idsel = item_id[0];
            if constexpr (NB == 2) idsel = cb ? item_id[1] : item_id[0];
            if (p) ring[slot] = ((uint32_t)row << 27) | (uint32_t)idsel;
            ring_cnt += __popcll(pm);
        }
    };
    // ---- main loop ---------------------------------------------------------------------------------------------------------
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint64_t hb_cur = 0;
    float pop_cur[NB];
    int id_cur[NB];
    bool ok_cur[NB];   // lane's item exists
    u32x4 bex_cur[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
        pop_cur[cb] = 0.f;
        id_cur[cb] = 0;
        ok_cur[cb] = false;
    }

    // ---- exact warm-up (ordered sweeps): the first kWarm tiles go through the fp32 matrix cores ---------------------------
    // In visiting order most of a user's final top K sits in the first few tiles, and until a list holds K entries EVERY
    // pair is a candidate: through the ring that is ~1000 exact rescorings per wave and tile, each pass a gather latency.
    // v_mfma_f32_32x32x2_f32 in the k order of v1 gives the same bits as the fmaf chain (that is what v1 is), so these
    // tiles are scored exactly on the matrix cores and their keys go straight into the lists: 64 MFMAs of 64 cycles per
    // 32 items, paid for 2 tiles only.
    // visiting order: 256 items (more costs more than it saves -- the thresholds are already high after them); natural
    // order: the record process K ln(n / K) is slower, 512 items pay (11.05 -> 9.9 ms at C3; 1024 items: the same)
    constexpr int kWarm = D <= 128 ? (ORD ? (PDA_KWARM * 2) / NB : PDA_KWARM_NAT) : 0;
    int k0 = 0;          // first tile of the pre-filtered loop
    int n32 = 0;         // statistics: 32-item tiles scored
    if constexpr (kWarm > 0) {
        constexpr int NC = D / 8;                 // k-chunks of 8
        constexpr int CPR4 = D / 4;               // 16-byte chunks per fp32 row
        constexpr int NLD4 = (32 * CPR4) / kThreads;
        float* Bt = reinterpret_cast<float*>(smem);          // [32][D] fp32, swizzled: exactly the bytes of the [64][D] bf16 tile
        const int nwarm = min(kWarm, nt);
        f32x4 areg[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row_ok) v = pda_load4<BF>(a.U, (size_t)uid * D + 4 * h + 8 * c);
            areg[c] = v;
        }
        const float* brow = Bt + j * D;
        const int bswz = swz<D>(j);
        for (int w = 0; w < nwarm; ++w) {
            const int t = tile_of(w);
            n32 += min(NB, (a.n_items_local - t * TW + 31) >> 5);
            float popw[NB];
            int idw[NB];
            lane_consts(t, popw, idw);
            const uint64_t hbw = hist_bits(t);
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                __syncthreads();                  // the previous block has been read by everyone
#pragma unroll
                for (int q = 0; q < NLD4; ++q) {
                    const int id = tid + kThreads * q;
                    const int jj = id / CPR4, ch = id % CPR4;
                    const int pos = min(t * TW + 32 * cb + jj, a.n_items_local - 1);
                    uint32_t item = (uint32_t)pos;
                    if constexpr (ORD) item = (uint32_t)aa.order[pos];
                    *reinterpret_cast<f32x4*>(Bt + jj * D + 4 * (ch ^ swz<D>(jj))) = pda_load4<BF>(a.I, (size_t)item * D + 4 * ch);
                }
                __syncthreads();
                f32x16 acc0 = zero16w(), acc1 = zero16w();
#pragma unroll
                for (int c = 0; c < NC; c += 2) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(brow + 4 * ((2 * c + h) ^ bswz));
                    const f32x4 b1 = *reinterpret_cast<const f32x4*>(brow + 4 * ((2 * c + 2 + h) ^ bswz));
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[c][q], b0[q], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[c + 1][q], b1[q], acc1, 0, 0, 0);
                    }
                }
                const f32x16 accx = acc0 + acc1;
                const bool okw = (t * TW + 32 * cb + j) < a.n_items_local;
                const uint32_t hbits = cb ? (uint32_t)(hbw >> 32) : (uint32_t)hbw;
                const bool any_hb = __any(hbits != 0u);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int lrow = wave * 32 + row;
                    float sc = accx[r];
                    if constexpr (HEAD == PDA_HEAD_POP) sc = (sc > 0.0f ? sc + 1.0f : __expf(sc)) * popw[cb];
                    bool p = okw && (sc >= taul[lrow]);
                    if (any_hb) {
                        const uint32_t hbr = (uint32_t)__shfl((int)hbits, row, 64);          // train items never enter
                        if ((hbr >> j) & 1u) p = false;
                    }
                    const uint64_t key = pda_pack_key(sc, (uint32_t)idw[cb]);
                    for (;;) {
                        bool ov = false;
                        if (p) {
                            const int slot = atomicAdd(&cntl[lrow], 1);
                            if (slot < kCap3) lists[(size_t)lrow * kCap3 + slot] = key;
                            else ov = true;
                        }
                        if (!__any(ov)) break;
                        pda_wave_sync();
                        uint64_t full = __ballot(lane < 32 && cntl[wave * 32 + (lane & 31)] >= kCap3);
                        while (full) {
                            const int rr = __builtin_ctzll(full);
                            full &= full - 1ull;
                            compact_list<kCap3>(my_lists + rr * kCap3, &cntl[wave * 32 + rr], &taul[wave * 32 + rr], K, lane);
                        }
                        p = ov && (sc >= taul[lrow]);
                    }
                }
            }
        }
        __syncthreads();                          // the last fp32 block has been read: the bf16 tiles may overwrite it
        pda_wave_sync();
        refresh_thr();
        k0 = nwarm;
    }

    if (k0 < nt) {
        const int tk = tile_of(k0);
        tile_load(tk, pA_h);
        lane_consts(tk, pop_cur, id_cur);
        bex_load(tk, bex_cur);
        tile_store(pA_h);
        hb_cur = kHistAtCand ? 0ull : hist_bits(tk);      // ordered main loop: history is masked at the candidate stage
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) ok_cur[cb] = (tk * TW + 32 * cb + j) < a.n_items_local;
    }
    __syncthreads();

    const uint16_t* bhrow = Bh + j * D;
    const int bsw = swzb<D>(j);

    // One iteration = one tile: MFMAs, then the filter on THAT tile's scores (v1 / v2 test the previous tile in the shadow of
    // the MFMA chain; with one MFMA per k-step the chain is ~15 % of the iteration, and carrying a second set of scores,
    // item constants and history bits costs more registers and moves than the overlap returns).
    auto iteration = [&](int k, u32x4 (&cur_h)[NLD]) __attribute__((always_inline)) -> bool {
        const bool has_next = (k + 1) < nt;
        const int tn = tile_of(min(k + 1, nt - 1));
        float pop_next[NB];
        int id_next[NB];
        tile_load(tn, cur_h);
        lane_consts(tn, pop_next, id_next);
        u32x4 bex_next[NB];
        bex_load(tn, bex_next);
        __builtin_amdgcn_sched_barrier(0);

        // One accumulator chain per column block (NB = 2) or per even/odd k-step (NB = 1), and the B operands of the next PF
        // MFMAs always in flight: left to itself hipcc keeps a single B quad live and every MFMA waits a full LDS latency.
        constexpr int S = NB * NM, PF = S < 8 ? S : 8;
        auto b_load = [&](int s_) __attribute__((always_inline)) -> u32x4 {
            const int cb = s_ / NM, mm = s_ % NM;
            return *reinterpret_cast<const u32x4*>(bhrow + cb * (32 * D) + 8 * ((2 * mm + h) ^ bsw));
        };
        f32x16 acc[2] = {zero16, zero16};
        u32x4 bq[PF];
#pragma unroll
        for (int s_ = 0; s_ < PF; ++s_) bq[s_] = b_load(s_);
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
            const int cb = s_ / NM, mm = s_ % NM;
            const int ai = NB == 2 ? cb : (mm & 1);
            acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[mm]), __builtin_bit_cast(bf16x8, bq[s_ % PF]),
                                                              acc[ai], 0, 0, 0);
            if (s_ + PF < S) bq[s_ % PF] = b_load(s_ + PF);
        }
        f32x16 sc[NB];
        if constexpr (NB == 2) {
            sc[0] = acc[0];
            sc[NB - 1] = acc[1];
        } else {
            sc[0] = acc[0] + acc[1];
        }
        uint64_t okm[NB], many = 0;
        bool clampy[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            uint64_t mc = 0;
            clampy[cb] = false;
            {
                u32x4 bx = bex_cur[cb];
                if constexpr (HEAD == PDA_HEAD_POP && !ORD) {
                    // natural order: the prep never saw the popularity (raw-type pieces) -- the 1/pop pieces and the constant
                    // are rebuilt here, ~15 VALU per block (v_rcp is within 1 ulp; the 5e-7 lowering covers it)
                    const float pv = pop_cur[cb];
                    const float ip = (pv > 0.f) ? fminf(__builtin_amdgcn_rcpf(pv) * 0.9999995f, 1.0e6f) : 1.0e6f;
                    uint32_t p1, p2, p3;
                    bf16_split3(ip, p1, p2, p3);
                    const uint32_t kk = ((pv == pv) ? 0x3F80u : 0xFF61u) | (bf16_up(8.0e-6f) << 16);
                    bx[0] = h ? kk : (p1 | (p2 << 16));
                    bx[1] = h ? bx[1] : (p1 | (p2 << 16));
                    bx[2] = h ? 0u : (p3 | (p1 << 16));
                    bx[3] = h ? 0u : (p3 | (p2 << 16));
                } else if constexpr (HEAD == PDA_HEAD_RAW && ORD) {
                    // ordered prep built with a popularity, raw head asked for: 1/pop := 1, constant := +8e-6
                    bx[0] = h ? bf16_up(8.0e-6f) : 0x00003F80u;
                    bx[1] = h ? bx[1] : 0x00003F80u;
                    bx[2] = h ? 0u : 0x3F800000u;
                    bx[3] = 0u;
                }
                sc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aex), __builtin_bit_cast(bf16x8, bx), sc[cb], 0, 0, 0);
                // (compiler-visible maxima, NOT inline asm: the hazard recogniser has to see the VALU read of the MFMA
                // result -- an asm v_max3 right behind the MFMA read the accumulator before it was written)
                // Integer maxima of the bit patterns: "some register is a positive float" == "the signed max is > 0", and
                // v_max3_i32 needs no canonicalising pre-max of every MFMA result (fmaxf: +6 VALU per block).  A NaN with
                // a clear sign bit counts as a candidate, which is the safe side.
                int ma = max(__float_as_int(sc[cb][0]), __float_as_int(sc[cb][1])), mb = max(__float_as_int(sc[cb][2]), __float_as_int(sc[cb][3]));
#pragma unroll
                for (int r = 4; r < 16; r += 4) {
                    ma = max(max(ma, __float_as_int(sc[cb][r])), __float_as_int(sc[cb][r + 1]));
                    mb = max(max(mb, __float_as_int(sc[cb][r + 2])), __float_as_int(sc[cb][r + 3]));
                }
                mc = __ballot(max(ma, mb) > 0);
                if constexpr (HEAD == PDA_HEAD_POP) {
                    // s~ + eps < 0: the head is exp(.) pop <= pop -- such an item can only matter to rows with thr < pop.
                    // Rare once the lists are warm (thr_min is the smallest threshold of the wave's rows).
                    clampy[cb] = __any(pop_cur[cb] > thr_min);
                    if (clampy[cb]) mc = ~0ull;
                }
            }
            okm[cb] = __ballot(ok_cur[cb]);
            many |= mc & okm[cb];
        }

        // All four waves drain their rings in the SAME iteration (flag set by whichever wave is filling up): the
        // latency-bound rescoring of the four waves then overlaps instead of stalling the workgroup four times.
        if (ring_cnt > kRingTrig && lane == 0) wgflag[k & 1] = 1;
        __syncthreads();  // every wave is done reading the tile
        const bool drain = wgflag[k & 1] != 0;
        if (tid == 0) wgflag[(k + 1) & 1] = 0;   // flag of iteration k-1: everyone has read it, nobody sets it before k+1
        uint64_t hb_next = 0;
        if (has_next) {
            tile_store(cur_h);
            if constexpr (!kHistAtCand) hb_next = hist_bits(tn);
        }
        if (many) {
            // The lane's own bit mask, built per lane: 32 wave-wide masks in SGPRs (spilled through VGPR lanes) and their
            // per-lane extraction were most of what a natural-order sweep -- which takes this path on almost every tile --
            // spent outside the MFMAs.
            uint32_t m = 0;
#pragma unroll
            for (int cb = NB - 1; cb >= 0; --cb) {
                uint32_t mcb = 0;
                {
                    // "register is a positive float" = sign bit of (0 - bits); one v_sub + one v_alignbit per register.
                    // (-0.0 counts as positive: a false candidate at worst)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        mcb = __builtin_amdgcn_alignbit(mcb, 0u - (uint32_t)__float_as_int(sc[cb][r]), 31);
                    if (clampy[cb]) {
                        int hv = h;
                        asm volatile("" : "+v"(hv));
#pragma unroll
                        for (int r = 0; r < 16; ++r) mcb |= (pop_cur[cb] > thr_of(r, hv)) ? (1u << (15 - r)) : 0u;
                    }
                }
                if (!ok_cur[cb]) mcb = 0;
                m = (m << 16) | mcb;
            }
            push_masks(m, hb_cur, id_cur);
        }
        bool stop = false;
        if constexpr (ORD) {
            // every 4th tile: can anything at or behind the next tile still reach one of my rows?  (bound: pda_score_topk_v2.hip,
            // row norms from LDS.)  Candidates still waiting in the ring can only raise thresholds.
            if ((k & PDA_VOTE) == PDA_VOTE && has_next && aa.sufA != nullptr) {
                const float sa = aa.sufA[tn * NB], sb = aa.sufB[tn * NB];
                const bool dead = __builtin_fmaf(nu_row, sb, sa) * 1.000002f < thr_own;    // one row per lane (both halves hold row j)
                const bool alldead = __all(dead);
                if (lane == 0) votes[wave] = alldead ? 1 : 0;
            }
        }
        if (drain && ring_cnt > 0) process_ring();
        __syncthreads();  // next tile visible
        if constexpr (ORD) {
            if ((k & PDA_VOTE) == PDA_VOTE && has_next && aa.sufA != nullptr) stop = (votes[0] & votes[1] & votes[2] & votes[3]) != 0;
        }

        hb_cur = hb_next;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            pop_cur[cb] = pop_next[cb];
            bex_cur[cb] = bex_next[cb];
            id_cur[cb] = id_next[cb];
            ok_cur[cb] = has_next && (tn * TW + 32 * cb + j) < a.n_items_local;
        }
        return stop;
    };
    for (int k = k0; k < nt; ++k) {
        n32 += min(NB, (a.n_items_local - tile_of(k) * TW + 31) >> 5);
        if (iteration(k, pA_h)) break;
    }
    if (tid == 0) atomicAdd(aa.visited, (unsigned long long)n32);
    // kernel identity (workspace + 16; tests read it back to prove WHICH kernel a call ran): generation 3 | ORD | HEAD | BF | d / 64
    if (tid == 0 && blockIdx.x == 0)
        reinterpret_cast<unsigned*>(aa.visited)[2] = (3u << 28) | ((ORD ? 1u : 0u) << 12) | ((unsigned)HEAD << 13) | ((BF ? 1u : 0u) << 14) | (unsigned)(D >> 6);
    if (ring_cnt > 0) process_ring();
    if (lane == 0) atomicAdd(reinterpret_cast<unsigned*>(aa.visited) - 1, n_cand);   // workspace + 4: u32 "pairs rescored"

    // ---- finalise: the lists are exact; sort and emit ---------------------------------------------------------------------
    for (int rr = 0; rr < 32; ++rr) {
        uint64_t* buf = my_lists + rr * kCap3;
        compact_list<kCap3>(buf, &cntl[wave * 32 + rr], &taul[wave * 32 + rr], K, lane);
        const int c = cntl[wave * 32 + rr];
        const int rb = utile * kUserTile + wave * 32 + rr;
        if (rb < a.n_users_blk && lane < K) {
            const uint64_t k = lane < c ? buf[lane] : 0ull;
            a.out_keys[((size_t)split * a.n_users_blk + rb) * K + lane] = k;
        }
    }
}

template <int D, int HEAD, bool ORD, bool BF>
int launch_v3(const ScoreArgs2& aa, hipStream_t stream) {
    const size_t smem = (D <= 128 ? 64 : 32) * D * sizeof(uint16_t) + (size_t)kUserTile * (kCap3 * sizeof(uint64_t) + 8) + 4 * kRing * sizeof(uint32_t) + 32;
    static int attr_set = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&score_topk_v3_kernel<D, HEAD, ORD, BF>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return PDA_ERR_LAUNCH;
        attr_set = 1;
    }
    const int utiles = (aa.a.n_users_blk + kUserTile - 1) / kUserTile;
    hipLaunchKernelGGL((score_topk_v3_kernel<D, HEAD, ORD, BF>), dim3((unsigned)(utiles * aa.a.n_splits)), dim3(kThreads), smem, stream, aa);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

}  // namespace

int pda_topk::launch_score_v3(const ScoreArgs2& aa, int d, int head, bool ordered, bool bf16, hipStream_t s) {
    if ((uint64_t)aa.a.item_offset + (uint64_t)aa.a.n_items_local > (1ull << 27)) return PDA_ERR_UNSUPPORTED;   // ring: 27-bit item ids
    if ((uint64_t)aa.a.n_items_local * (uint64_t)d >= (1ull << 32)) return PDA_ERR_UNSUPPORTED;                  // 32-bit plane offsets
#define PDA_V3_(DD, ORDV, BFV) \
    (head == PDA_HEAD_POP ? launch_v3<DD, PDA_HEAD_POP, ORDV, BFV>(aa, s) : launch_v3<DD, PDA_HEAD_RAW, ORDV, BFV>(aa, s))
#define PDA_V3(DD)                                                             \
    case DD:                                                                   \
        if (bf16) return ordered ? PDA_V3_(DD, true, true) : PDA_V3_(DD, false, true);   \
        return ordered ? PDA_V3_(DD, true, false) : PDA_V3_(DD, false, false);
    switch (d) {
        PDA_V3(64) PDA_V3(128) PDA_V3(256)
        default: return PDA_ERR_UNSUPPORTED;
    }
#undef PDA_V3
#undef PDA_V3_
}
