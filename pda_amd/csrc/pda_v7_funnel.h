// The funnel (round 5): exact top-K for sweeps that meet HUNDREDS of list insertions per user -- the raw head (MF/train_new_api.py:597-598,
// the ranking the reference evaluates in every epoch, :1139-1141, and the only one of --train normal, :1160-1165) -- on the huge geometry's
// machine mapping.  Compiled in pda_score_funnel.hip (which holds the schedule, the workspace layout and the host entry points).
//
// What bound generation 4's many-candidates geometry: a running exact list takes K ln(n / n0) true insertions per user
// (359 measured at config 3), each an exact rescoring that gathers a 512-byte item row -- 46 GB per launch against 0.4 GB algorithmic,
// matrix pipe 24 % busy.  The funnel keeps everything approximate until the end:
//   * sweep7_kernel (loops: pda_v7_emit_loop_asm.h, fp16 operands) scores a PART of the catalogue against FIXED per-user thresholds and writes,
//     for every (user, quarter of a half-tile) whose best bound s~ + ct beats the threshold, the lane's eight bounds to the lane's own list: no
//     list maintenance, no gather, no exit from the loop.  The first launch (Loop7M) writes nothing: it keeps maxima, maxthr7_kernel turns them
//     into the first thresholds;
//   * expand7_kernel turns a launch's entries into the row's pool {ub >= the thresholds used so far} (train items masked), threshold7_kernel
//     finds the threshold of the next part -- the r-th largest LOWER bound, r < K chosen so that the threshold stays below the row's final
//     K-th value with probability 1 - g_fail_p (1e-6 per row and launch; the parts grow by 4, then by 2) -- and, after the last part,
//     tk = the K-th largest lower bound (valid by construction: K items reach it);
//   * resolve7_kernel rescores the final pool {ub >= tk} exactly -- the fp32 chain of the oracle --, sorts and writes the K keys.  A row whose
//     bets were lost (fewer than K pairs of the final pool reach a threshold that was used) or whose lists overflowed is served inside the
//     same call by generation 4's exact lists, seeded with the row's tk (pda_score_funnel.hip: fail list, fallback sweep, fail_merge7_kernel).
// Every returned score is an fp32 score of the oracle's chain; the fp16 products decide only what is looked at, under a rigorous bound formed
// from the ACTUAL rounding residuals (see sweep7_kernel).
#pragma once
#include "pda_v7_emit_loop_asm.h"

struct Args7 {
    const unsigned char* rows5;      // the swizzled bf16 image (prep): 32-item half-tiles
    const float* meta5;              // (pmax, nmax, 0, 0) per half-tile
    const unsigned char* ufrag;      // the user block as MFMA operands (uprep5_kernel)
    const float* unorm;              // padded ||u|| per block row
    const float* uerr;               // padded ||u - fp16(u)|| per block row
    const float* thr;                // [n_users_blk] the thresholds of this launch (already lowered)
    unsigned char* elist;            // [workgroup][wave] list regions of cap_e entries per list
    unsigned* ecnt;                  // [workgroup][wave][user block][lane] cursors
    float* eu_wave;                  // [user tile][wave][2] the wave's (A, B) of ct = pmax + A nmax + B rmax (what the launch formed ct with)
    unsigned* stats;
    int n_users_blk, n_splits, n_tiles;
    int tile_lo, tile_hi;            // this launch scores the 64-item tiles [tile_lo, tile_hi) of the visiting order (every split: its own of them)
    int cap_e;                       // entries per list
    float* mrun;                     // [workgroup][wave][(2 NU + 1) 64]: the first launch's maxima (Loop7M): the two largest per (user block, lane), the largest ct
    const int* n_users_dev;          // or NULL: the number of rows that exist (a device-side count; workgroups beyond it leave at once)
    const int* prep_hdr;             // the prep's header: word 3 = 1 <=> its half-tile image is fp16 (pda_item_prep7_*)
};

// the local tiles [i0, i1) of split sp that lie in the global tile range [lo, hi)
__device__ __forceinline__ void stage_tiles7(int n_tiles, int sp, int S, int lo, int hi, int& i0, int& i1) {
    const int nt = split_tiles(n_tiles, sp, S);
    i0 = min(nt, max(0, (lo - sp + S - 1) / S));
    i1 = min(nt, max(0, (hi - sp + S - 1) / S));
    i1 = max(i1, i0);
}

// one launch of the emitting sweep: d = 64 / 128: 1 024-user workgroups (UPW = 256); d = 256: 512-user workgroups (UPW = 128)
// MAXM: the funnel's first launch -- no thresholds yet and nothing written per half-tile: every lane keeps the two largest maxima of each of its user
// blocks (Loop7M), maxthr7_kernel turns them into the first thresholds.
template <int D, bool BF, int UPW, bool MAXM = false>
__global__ void __launch_bounds__(256, 1) sweep7_kernel(Args7 g) {
    constexpr int NK = D / 32, NU = UPW / 16, UT = 4 * UPW, SS = slot_bytes5(D);
    static_assert(SS == Loop7<D, NU>::kSlotBytes, "one LDS image for all loops");
    constexpr unsigned ES = Loop7<D, NU>::kEntryStride;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x % g.n_splits, utile = blockIdx.x / g.n_splits;
    const int n_rows = g.n_users_dev != nullptr ? min(g.n_users_blk, *g.n_users_dev) : g.n_users_blk;
    if (utile * UT >= n_rows) return;
    int i0, i1;
    stage_tiles7(g.n_tiles, split, g.n_splits, g.tile_lo, g.tile_hi, i0, i1);
    // (a prep whose image is bf16 -- pda_item_prep4_*: the products below would be of reinterpreted bits -- is refused: error word 7, nothing is scored, every row
    // ends with an empty pool and is served by the exact fallback)
    const bool prep_ok = g.prep_hdr[3] == 1;
    if (!prep_ok && tid == 0 && blockIdx.x == 0) g.stats[0] = 7u;
    const unsigned hend = prep_ok ? 2u * (unsigned)(i1 - i0) : 0u;
    const int row0 = wave * UPW, j = lane & 15;
    const size_t widx = (size_t)blockIdx.x * 4 + wave;
    unsigned* cnt = g.ecnt + widx * (NU * 64);
    // The bound of the bf16 product, from the ACTUAL rounding residuals (Cauchy-Schwarz on u . i - u~ . i~ = (u - u~) . i + u~ . (i - i~)):
    //     |s~ - s| <= ||u - u~|| ||i|| + ||u~|| ||i - i~|| + 2^-14 ||u|| ||i||          (the last term: fp32 accumulation, pda_score_topk_v3.hip)
    // which is rigorous like the worst case 2^-7 ||u|| ||i|| of the other kernels and ~2.5 x tighter on dense rows (an RNE residual is ~0.4 of
    // its bound in norm).  Per launch the wave's maxima A = max (||u - u~|| + 2^-14 ||u||), B = max ||u~|| and the half-tile's maxima nmax, rmax:
    //     ct = pmax + A nmax + B rmax.
    float ea = 0.f, eb = 0.f, thr[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int rb = utile * UT + row0 + 16 * u + j;
        const float nu_r = rb < n_rows ? g.unorm[rb] : 0.f, ue_r = rb < n_rows ? g.uerr[rb] : 0.f;
        ea = fmaxf(ea, ue_r + nu_r * 6.103515625e-5f);
        eb = fmaxf(eb, nu_r + ue_r);             // ||u~|| <= ||u|| + ||u - u~||: rigorous for any rounding (fp16 subnormals included; round 5 used 1.0039 ||u||)
        thr[u] = rb < n_rows ? fminf(fmaxf(g.thr[rb], -1.0e30f), 1.0e30f) : 1.0e30f;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        ea = fmaxf(ea, __shfl_xor(ea, o, 64));
        eb = fmaxf(eb, __shfl_xor(eb, o, 64));
    }
    const float eu = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ea * 1.001f)));
    const float eu2 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(eb * 1.001f)));
    if (lane == 0 && split == 0) {
        g.eu_wave[2 * (utile * 4 + wave)] = eu;
        g.eu_wave[2 * (utile * 4 + wave) + 1] = eu2;
    }
    if (tid == 0 && blockIdx.x == 0) g.stats[4] = (4u << 28) | (7u << 8) | ((BF ? 1u : 0u) << 14) | (unsigned)(D >> 6);     // kernel identity: geometry 7
    [[maybe_unused]] float* mr = MAXM ? g.mrun + widx * ((2 * NU + 1) * 64) : nullptr;
    if (hend == 0u) {
        if constexpr (MAXM) {
#pragma unroll
            for (int u = 0; u < 2 * NU; ++u) mr[u * 64 + lane] = -INFINITY;
            mr[2 * NU * 64 + lane] = 0.f;
        } else {
#pragma unroll
            for (int u = 0; u < NU; ++u) cnt[u * 64 + lane] = 0u;
        }
        return;
    }
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned char* my_ufrag = g.ufrag + ((size_t)utile * 4 + wave) * (size_t)(NU * NK * 1024);
    const size_t region = (size_t)g.cap_e * ES;
    const size_t base = (size_t)(g.elist + widx * region);
    u32x4 rsrc;
    rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)base);
    rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32) & 0xFFFFu);
    rsrc[2] = __builtin_amdgcn_readfirstlane((unsigned)region);
    rsrc[3] = 0x00020000u;
    const size_t img = (size_t)g.rows5, meta = (size_t)g.meta5;
    if constexpr (MAXM)
        Loop7M<D, NU>::run(0u, 0u, hend, ring_lds, 1024u * (unsigned)wave, (unsigned)(split + i0 * g.n_splits), (unsigned)g.n_splits, (unsigned)img, (unsigned)(img >> 32),
                           (unsigned)meta, (unsigned)(meta >> 32), eu, eu2, my_ufrag, mr, (unsigned)lane * 16u);
    else
        Loop7<D, NU>::run(0u, 0u, hend, ring_lds, 1024u * (unsigned)wave, (unsigned)(split + i0 * g.n_splits), (unsigned)g.n_splits, (unsigned)img, (unsigned)(img >> 32),
                          (unsigned)meta, (unsigned)(meta >> 32), eu, eu2, my_ufrag, rsrc, cnt, thr, (unsigned)lane * 16u);
#ifdef PDA_V7_DUMP_LDS      /* debugging: workgroup 0 leaves its LDS behind the lists */
    __syncthreads();
    if (blockIdx.x == 0) {
        unsigned char* dst = g.elist + (size_t)gridDim.x * 4 * region;
        for (int i = tid * 16; i < kNSlot5 * SS; i += 256 * 16) *reinterpret_cast<u32x4*>(dst + i) = *reinterpret_cast<const u32x4*>(smem + i);
    }
#endif
}


// ---- the state of a block's rows between the launches of a funnel -------------------------------------------------------------
constexpr int kCand7 = 192;           // candidates a row keeps between two launches (K + the pairs inside the bound's band; more = a hard failure)
constexpr int kPool7 = 512;           // a row's pool inside threshold7_kernel: what it kept + what the launch added
struct Rows7 {
    float* thr;                       // [rows] the threshold of the emitting launches: the one in use, then the next one (lowered: strictly below its source)
    float* tk;                        // [rows] the last launch's K-th largest LOWER bound: at least K unmasked items reach it
    float* tmax;                      // [rows] the (un-lowered) source of thr: the thresholds only rise, so it is the largest one any launch has used
    int* ncand;                       // [rows]
    uint64_t* qpool;                  // [rows][4 S][cap_q][2]: what expand7_kernel found in the (quarter, split) lists of the launch just finished (same pairs of words)
    unsigned* qcnt;                   // [rows][4 S]
    int cap_q;
    uint64_t* cand;                   // [rows][kCand7][2]: what the row keeps between two launches: ordered(v) << 32 | visiting position, ordered(ub) << 32 | ordered(lb)  (select7_kernel)
    unsigned* flags;                  // [rows] bit 0: a list or the pool overflowed; bit 1: the last selection found a bet lost
};
struct Sel7 {
    Args7 e;
    Rows7 r;
    const u32x4* pinfo;               // per visiting position: (||i|| padded, ||i' - i~'|| padded, local id, popularity) -- the prep's dense 16-byte records
    const int32_t* users;
    const int64_t* hist_indptr;
    const int32_t* hist_indices;
    const uint32_t* bloom;            // [rows][32] or NULL
    int hist_row_mode, item_offset, n_items_local, K;
    int first_launch;                 // the launch ran against -inf (no maxima launch in front of it): expand7_kernel takes shortcuts, threshold7_kernel masks
    int rank_next;                    // the next launch's threshold: the rank_next-th largest lower bound (0: this was the last launch)
    // resolve7_kernel
    const void* U;
    const void* I;
    uint64_t* out_keys;               // [rows][K]
    int* fail_list;                   // [rows] block rows that need the exact fallback; fail_count[0] of them
    int* fail_count;
};

__device__ __forceinline__ float lowered7(float tq) {     // strictly below tq (ties must pass), +-1e30 for +-inf (as sweep5_kernel's thr_of)
    const float tf = (tq == INFINITY || tq == -INFINITY) ? tq : tq - fabsf(tq) * 1.52587890625e-5f - 1e-30f;
    return fminf(fmaxf(tf, -1.0e30f), 1.0e30f);
}

// The first thresholds, from the maxima of the first launch (sweep7_kernel<.., MAXM>).  A lane of the sweep kept, per user block, the two largest
// of its half-tile maxima: of (row, quarter) -- a fixed quarter of the items, random like the visiting order -- the two best bounds seen among the
// launch's m items.  T = the smallest over the four quarters of the quarter's SECOND largest lower bound: eight distinct items reach it, two per
// quarter, and the items of the whole catalogue that reach it number at least (n / m) x Gamma(8) (four independent Gamma(2) classes): the bet of
// rank_for7 with rank 8.  Lower bounds: a maximum is s~ + ct of its half-tile; minus twice the wave's largest ct (which half-tile is not kept).
// Four lanes per row (a quarter each); the splits' maxima of a quarter are merged first.
template <int D>
__global__ void __launch_bounds__(256) maxthr7_kernel(Sel7 g) {
    // Four lanes per row, one per quarter (round 6; one thread per row walked 12 S dependent-free but serially issued loads: 63 us for a 2 048-user
    // block with its 32 item splits -- eight workgroups on the whole chip); the splits four at a time, their loads in flight together.
    constexpr int UPW = D == 256 ? 128 : 256, UT = 4 * UPW, NU = UPW / 16, MR = (2 * NU + 1) * 64;
    const int n_rows = g.e.n_users_dev != nullptr ? min(g.e.n_users_blk, *g.e.n_users_dev) : g.e.n_users_blk;
    const int rb = blockIdx.x * 64 + (threadIdx.x >> 2), hh = threadIdx.x & 3;
    const bool ok = rb < n_rows;
    const int rbs = ok ? rb : 0;
    const int S = g.e.n_splits;
    const int utile = rbs / UT, w = (rbs % UT) / UPW, u = (rbs % UPW) >> 4, j = rbs & 15;
    float b1 = -INFINITY, b2 = -INFINITY;                        // the quarter's two largest lower bounds over the splits
    auto take = [&](float m, float ctm) __attribute__((always_inline)) {
        const float lb = m == -INFINITY ? -INFINITY : (m - ctm) - ctm - (fabsf(m) + ctm) * 4.8e-7f;
        const float lo = fminf(b1, lb);
        b1 = fmaxf(b1, lb);
        b2 = fmaxf(b2, lo);
    };
    for (int sp0 = 0; sp0 < S; sp0 += 4) {
        float ctm[4], m0[4], m1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int sp = min(sp0 + q, S - 1);
            const float* mr = g.e.mrun + (((size_t)utile * S + sp) * 4 + w) * MR;
            ctm[q] = mr[2 * NU * 64 + j + 16 * hh];
            m0[q] = mr[u * 64 + j + 16 * hh];
            m1[q] = mr[(NU + u) * 64 + j + 16 * hh];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (sp0 + q < S) {                                   // (same order as one thread per row took them: split by split, k = 0 then 1)
                take(m0[q], ctm[q]);
                take(m1[q], ctm[q]);
            }
    }
    float t = b2;                                                // the smallest over the row's four quarters (lanes 4 r .. 4 r + 3)
    t = fminf(t, __shfl_xor(t, 1, 64));
    t = fminf(t, __shfl_xor(t, 2, 64));
    if (ok && hh == 0) {
        g.r.thr[rb] = lowered7(t);
        g.r.tmax[rb] = t;
    }
}

// The selection between two launches, in two kernels (one wave per row doing both was latency-bound: ~15 dependent round trips per row at five
// waves per SIMD, 1 - 2.6 ms per launch at 262 144 rows).  Raw head.  The bounds of a pair come from the value v = s~ + ct the launch wrote:
// s~ = v - ct(half-tile), then +- the PAIR's own bound A_u ||i|| + B_u ||i - i~|| (sweep7_kernel) -- tighter than ct, which took the wave's and the
// half-tile's maxima.
//   * expand7_kernel: one wave per user block of a sweep wave (16 rows x 4 quarters = its 64 lists, a lane per list), entry after entry: every value
//     above the launch's threshold whose own upper bound reaches the thresholds used so far, and that is no train item, goes to the list's OWN
//     slots of the row's pool in the workspace (a lane per (row, quarter, split): no atomics, nothing to wait for behind the stores).  Per entry
//     ONE dependent round trip: the half-tile's meta entry, the values' tails and their Bloom words (keyed by visiting position: known from the
//     entry) travel together; the next entry is requested before this one is worked on.
//   * threshold7_kernel: one wave per row.  Not the last launch: T = the rank_next-th largest LOWER bound of the pool (kept + new).  The bet
//     (schedule7, rank_for7): at least K items of the whole catalogue reach T, i.e. T <= the row's final K-th value -- so whatever lies below T is
//     dropped for good, here and in the next launch.  The last launch: tk = the K-th largest lower bound of the pool -- K items reach it, whatever
//     the bets were; the bets held iff tk >= the thresholds that were used.  Then the pool {ub >= tk} holds the exact best K.  Else: the fallback.
__device__ __forceinline__ unsigned bloom7_h1(unsigned pos) { return (pos * 0x9E3779B1u) >> 22; }
__device__ __forceinline__ unsigned bloom7_h2(unsigned pos) { return (pos * 0x85EBCA6Bu + 0x27D4EB2Fu) >> 22; }
// the rows' train items as 1024-bit Bloom filters over VISITING POSITIONS (as hist_bloom4_kernel over item ids): 32 rows per workgroup, 8 lanes per row
__global__ void __launch_bounds__(256) hist_bloom7_kernel(const int32_t* __restrict__ users, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                          int hist_row_mode, int n_users_blk, const int* __restrict__ pos_of, int item_offset, int n_items_local,
                                                          uint32_t* __restrict__ bloom) {
    __shared__ uint32_t w[32 * 32];
    const int tid = threadIdx.x, r = tid >> 3, sub = tid & 7;
    for (int q = tid; q < 32 * 32; q += 256) w[q] = 0u;
    __syncthreads();
    const int u = (int)blockIdx.x * 32 + r;
    if (u < n_users_blk) {
        const int64_t hr = hist_row_mode == PDA_HIST_BY_USER_ID ? (int64_t)users[u] : (int64_t)u;
        const int64_t b = indptr[hr], e = indptr[hr + 1];
        for (int64_t i0 = b + sub; i0 < e; i0 += 8 * 4) {          // (four ids, then four positions of a lane in flight: two dependent loads per train item)
            int loc[4];
            unsigned pos[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                loc[q] = i0 + 8 * q < e ? indices[i0 + 8 * q] - item_offset : -1;
                if (loc[q] >= n_items_local) loc[q] = -1;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) pos[q] = loc[q] >= 0 ? (unsigned)pos_of[loc[q]] : 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (loc[q] >= 0) {
                    const unsigned h1 = bloom7_h1(pos[q]), h2 = bloom7_h2(pos[q]);
                    atomicOr(&w[r * 32 + (h1 >> 5)], 1u << (h1 & 31u));
                    atomicOr(&w[r * 32 + (h2 >> 5)], 1u << (h2 & 31u));
                }
        }
    }
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * 32 * 32;
    for (int q = tid; q < 32 * 32; q += 256)
        if ((int)blockIdx.x * 32 + (q >> 5) < n_users_blk) bloom[base + q] = w[q];
}

template <int D>
__global__ void __launch_bounds__(256) expand7_kernel(Sel7 g) {
    constexpr int UPW = D == 256 ? 128 : 256, UT = 4 * UPW, NU = UPW / 16;
    constexpr unsigned ES = 64u * NU * 48u, LS = NU * 48u;
    constexpr int EPW = 64 / NU;                                 // entries of a list per step of the wave
    const int lane = threadIdx.x & 63;
    const int n_rows = g.e.n_users_dev != nullptr ? min(g.e.n_users_blk, *g.e.n_users_dev) : g.e.n_users_blk;
    const int S = g.e.n_splits;
    // One wave per LANE of a sweep wave: that lane's NU lists (its user blocks: NU rows, one quarter) lie side by side in every entry slot -- NU x 48
    // contiguous bytes -- so a wave reads EPW entry slots of NU lists per step, coalesced, and a list of 20 entries is 5 steps, not 20.
    // wave id -> (user tile, split, sweep wave, sweep lane); lane -> (list = user block u, entry slot es)
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = wid & 63, w = (wid >> 6) & 3, ws = wid >> 8, sp = ws % S, utile = ws / S;
    if (utile * UT >= n_rows) return;
    const int u = lane % NU, es = lane / NU;
    const int hh = l >> 4;
    const int rb = utile * UT + w * UPW + 16 * u + (l & 15);
    const bool row_ok = rb < n_rows;
    const int rbs = row_ok ? rb : 0;
    const size_t qidx = ((size_t)rbs * 4 + hh) * S + sp;         // this list's slots of the row's pool
    int i0, i1;
    stage_tiles7(g.e.n_tiles, sp, S, g.e.tile_lo, g.e.tile_hi, i0, i1);
    const unsigned hend = 2u * (unsigned)(i1 - i0);
    const size_t widx = ((size_t)utile * S + sp) * 4 + w;
    unsigned c = (row_ok && hend > 0u) ? g.e.ecnt[widx * (NU * 64) + u * 64 + l] / ES : 0u;
    if (c > (unsigned)g.e.cap_e) {
        c = (unsigned)g.e.cap_e;
        if (es == 0) atomicOr(&g.r.flags[rb], 1u);
    }
    unsigned cmax = c;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cmax = max(cmax, (unsigned)__shfl_xor((int)cmax, o, 64));
    if (cmax == 0u) {
        if (row_ok && es == 0) g.r.qcnt[qidx] = 0u;
        return;
    }
    const float thr_used = fminf(fmaxf(g.r.thr[rbs], -1.0e30f), 1.0e30f);      // (as sweep7_kernel clamps it)
    const uint32_t t_old_o = pda_ordf(g.r.tmax[rbs] + 0.0f);                   // the (un-lowered) source of thr_used: -inf in the first launch
    const bool first = g.first_launch != 0;
    const float eu = g.e.eu_wave[2 * (utile * 4 + w)], eu2 = g.e.eu_wave[2 * (utile * 4 + w) + 1];      // what the launch formed ct with
    const float ua = (g.e.uerr[rbs] + g.e.unorm[rbs] * 6.103515625e-5f) * 1.001f, ub2 = (g.e.unorm[rbs] + g.e.uerr[rbs]) * 1.001f;   // the ROW's own A, B
    const bool hist_on = g.hist_indptr != nullptr;
    long long hb = 0, he = 0;
    if (hist_on && row_ok) {
        const int64_t hr = g.hist_row_mode == PDA_HIST_BY_USER_ID ? (int64_t)g.users[rb] : (int64_t)rb;
        hb = g.hist_indptr[hr];
        he = g.hist_indptr[hr + 1];
    }
    const unsigned char* lp = g.e.elist + widx * ((size_t)g.e.cap_e * ES) + (size_t)l * LS + (size_t)u * 48;
    uint64_t* prow = g.r.qpool + qidx * (size_t)(2 * g.r.cap_q);
    int cq = 0;                                                  // pairs found in this list so far (the same in its EPW lanes)
    f32x4 na = {0.f, 0.f, 0.f, 0.f}, nb = na;
    unsigned nhd = 0xFFFFFFFFu;
    if ((unsigned)es < c) {
        const unsigned char* ep = lp + (size_t)es * ES;
        na = *reinterpret_cast<const f32x4*>(ep);
        nb = *reinterpret_cast<const f32x4*>(ep + 16);
        nhd = *reinterpret_cast<const unsigned*>(ep + 32);
    }
    for (unsigned e0 = 0; e0 < cmax; e0 += EPW) {
        const unsigned e = e0 + (unsigned)es;
        const f32x4 a = na, b = nb;
        const unsigned hd = nhd;
        bool have = e < c && hd < hend;                         // (the two half-tiles behind the end write entries too)
        nhd = 0xFFFFFFFFu;
        if (e + EPW < c) {
            const unsigned char* ep = lp + (size_t)(e + EPW) * ES;
            na = *reinterpret_cast<const f32x4*>(ep);
            nb = *reinterpret_cast<const f32x4*>(ep + 16);
            nhd = *reinterpret_cast<const unsigned*>(ep + 32);
        }
        const unsigned pos0 = (unsigned)(sp + (i0 + (int)((have ? hd : 0u) >> 1)) * S) * 64u + (hd & 1u) * 32u + 4u * (unsigned)hh;
        // which of the entry's eight values are above the launch's threshold: one, as a rule (the entry exists because one was) -- the values are
        // worked on one at a time, the first two with their loads issued together (8 x the work for all eight was most of this kernel's issue time)
        float vv[8];
        unsigned m = 0u;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            vv[r] = r < 4 ? a[r & 3] : b[r & 3];
            const bool p = have && vv[r] > thr_used && pos0 + 16u * (unsigned)(r >> 2) + (unsigned)(r & 3) < (unsigned)g.n_items_local;
            m |= p ? 1u << r : 0u;
        }
        if (!__any(m != 0u)) continue;                          // (wave-uniform: the shuffles below are reached by all lanes or none)
        float4 mt = {0.f, 0.f, 0.f, 0.f};
        if (m != 0u) mt = *reinterpret_cast<const float4*>(g.e.meta5 + 4 * (size_t)(pos0 >> 5));
        const float ct = __builtin_fmaf(eu2, mt.z, __builtin_fmaf(eu, mt.y, mt.x));
        auto value_of = [&](int r) __attribute__((always_inline)) -> float {
            float v = vv[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) v = r == q ? vv[q] : v;
            return v;
        };
        // one value of the lane: its position's record (||i||, ||i - i~||, local id, popularity) and the two Bloom words by visiting position travel
        // together.  The first launch against -inf (no maxima launch: everything passes, 256 items per row of which a handful stay) takes the half-tile's
        // maxima instead (looser, as valid) and leaves the train-item check of what stays to threshold7_kernel: its threshold may then be one of the
        // row's train items -- a rank less on a rank that is chosen with margin (rank_for7).
        struct Req { bool on; unsigned pos; float v; f32x4 tl; uint32_t bw1, bw2; };
        auto request = [&](bool on, int r) __attribute__((always_inline)) -> Req {
            Req q;
            q.on = on;
            q.pos = pos0 + 16u * (unsigned)(r >> 2) + (unsigned)(r & 3);
            q.v = value_of(r);
            q.tl = f32x4{mt.y, mt.z, 0.f, 0.f};
            q.bw1 = 0xFFFFFFFFu;
            q.bw2 = 0xFFFFFFFFu;
            if (on && !first) q.tl = __builtin_bit_cast(f32x4, g.pinfo[q.pos]);
            if (hist_on && !first && g.bloom != nullptr && on) {
                q.bw1 = g.bloom[(size_t)rb * 32 + (bloom7_h1(q.pos) >> 5)];
                q.bw2 = g.bloom[(size_t)rb * 32 + (bloom7_h2(q.pos) >> 5)];
            }
            return q;
        };
        auto finish = [&](const Req& q) __attribute__((always_inline)) {
            const float bp = __builtin_fmaf(ua, q.tl[0], ub2 * q.tl[1]);
            const float st = q.v - ct, guard = (fabsf(q.v) + ct) * 4.8e-7f;               // (the roundings of the accumulator and of this subtraction)
            const uint32_t ubo = pda_ordf(st + bp + guard + 0.0f), lbo = pda_ordf(st - bp - guard + 0.0f);
            bool p = q.on && ubo >= t_old_o;                     // (below a threshold already used: dropped for good)
            if (hist_on && !first) {                             // train items never enter: a binary search behind the two Bloom bits
                const bool look = p && (((q.bw1 >> (bloom7_h1(q.pos) & 31u)) & (q.bw2 >> (bloom7_h2(q.pos) & 31u)) & 1u) != 0u);
                if (__any(look)) {
                    if (look) {
                        const int item = g.item_offset + __float_as_int(q.tl[2]);
                        long long lo = hb, hi2 = he;
                        while (lo < hi2) {
                            const long long mid = (lo + hi2) >> 1;
                            if (g.hist_indices[mid] < item) lo = mid + 1; else hi2 = mid;
                        }
                        if (lo < he && g.hist_indices[lo] == item) p = false;
                    }
                }
            }
            // the list's slots: its EPW lanes (NU apart) one after the other
            const int np = p ? 1 : 0;
            int before = 0, total = 0;
#pragma unroll
            for (int k = 0; k < EPW; ++k) {
                const int o = __shfl(np, u + k * NU, 64);
                before += k < es ? o : 0;
                total += o;
            }
            const int slot = cq + before;                       // (past the list's slots: threshold7_kernel sees the count and flags the row)
            if (p && slot < g.r.cap_q)
                *reinterpret_cast<ulonglong2*>(prow + 2 * slot) = make_ulonglong2(((uint64_t)pda_ordf(q.v + 0.0f) << 32) | (uint64_t)q.pos, ((uint64_t)ubo << 32) | lbo);
            cq += total;
        };
        const int r1 = __builtin_ctz(m | 0x100u);
        const unsigned m1 = m & (m - 1u);
        const int r2 = __builtin_ctz(m1 | 0x100u);
        unsigned mr = m1 & (m1 - 1u);
        const Req q1 = request(m != 0u, r1 & 7), q2 = request(m1 != 0u, r2 & 7);
        finish(q1);
        if (__any(m1 != 0u)) finish(q2);
        while (__any(mr != 0u)) {                               // (three and more of the eight: one more round trip each)
            const int r3 = __builtin_ctz(mr | 0x100u);
            const Req q3 = request(mr != 0u, r3 & 7);
            mr &= mr - 1u;
            finish(q3);
        }
    }
    if (row_ok && es == 0) g.r.qcnt[qidx] = (unsigned)cq;
}

template <int D, int NSL>
__global__ void __launch_bounds__(256) threshold7_kernel(Sel7 g) {
    const int lane = threadIdx.x & 63;
    const int n_rows = g.e.n_users_dev != nullptr ? min(g.e.n_users_blk, *g.e.n_users_dev) : g.e.n_users_blk;
    const int rb = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rb >= n_rows) return;
    const int K = g.K;
    const int S = g.e.n_splits, nl = 4 * S;
    const int nkept = g.r.ncand[rb];                             // what the row kept
    const float t_old = g.r.tmax[rb];
    bool over = false;
    int n = nkept;
    constexpr int NKK = kCand7 / 64;
    // NSL > 0: the row's (quarter, split) lists straight into registers -- slot t = lane + 64 k of the flattened [list][cap_q] space is real iff its index
    // inside its list is below the list's count: two round trips (the counts, then every pair).  NSL = 0 (more than eight item splits): through an LDS pool.
    constexpr int NKP = NSL > 0 ? NKK + NSL : kPool7 / 64;
    uint64_t key[NKP], bnd[NKP];
    if constexpr (NSL > 0) {
        const int cap_q = g.r.cap_q, slots = nl * cap_q;
        // Which slot a (lane, group) pair reads.  The lists fill from slot 0 and hold ~10 entries of their 64: with the lists INTERLEAVED -- 64 / nl lanes
        // per list, group k = the k-th run of 64 / nl slots of every list -- the pairs sit in the first one or two groups and the others are dead (skipped by
        // the descent and the keep pass: a third of this kernel's instructions); list after list (group = 64 consecutive slots) where nl does not divide 64.
        const int lpl = 64 / nl;
        const bool inter = nl * lpl == 64 && NSL * lpl >= cap_q;
        int lcn[NSL], slot_t[NSL];
        bool in_list[NSL];
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            const int t = lane + 64 * k;
            const int list = inter ? lane / lpl : t / cap_q, idx = inter ? lane % lpl + k * lpl : t % cap_q;
            in_list[k] = inter ? idx < cap_q : t < slots;
            slot_t[k] = list * cap_q + idx;
            lcn[k] = (in_list[k] || inter) ? (int)g.r.qcnt[(size_t)rb * nl + list] : 0;
            in_list[k] = in_list[k] && idx < lcn[k];
        }
#pragma unroll
        for (int k = 0; k < NKK; ++k) {
            const int i = lane + 64 * k;
            const ulonglong2 kv = i < nkept ? *reinterpret_cast<const ulonglong2*>(g.r.cand + ((size_t)rb * kCand7 + i) * 2) : make_ulonglong2(0ull, 0ull);
            key[k] = kv.x;
            bnd[k] = kv.y;
        }
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            const bool real = in_list[k];
            over = over || lcn[k] > cap_q;
            const ulonglong2 kv = real ? *reinterpret_cast<const ulonglong2*>(g.r.qpool + ((size_t)rb * slots + slot_t[k]) * 2) : make_ulonglong2(0ull, 0ull);
            key[NKK + k] = kv.x;
            bnd[NKK + k] = kv.y;                                 // (ordered bounds of real pairs are > 0)
            n += __popcll(__ballot(real));
        }
        over = __any(over);
    } else {
        __shared__ ulonglong2 pool_s[4][kPool7];
        ulonglong2* pool = pool_s[threadIdx.x >> 6];
        for (int i = lane; i < n; i += 64) pool[i] = *reinterpret_cast<const ulonglong2*>(g.r.cand + ((size_t)rb * kCand7 + i) * 2);
        for (int l0 = 0; l0 < nl; l0 += 64) {
            // the launch's lists, 64 per round, a LANE per list: its count, its place in the pool (a prefix over the lanes: the same order as list after list),
            // then every lane copies its own list -- the rounds are the longest list's entries (a handful), not the number of lists.  (Round 5 walked the
            // lists one after the other: 128 dependent little loads per row at 32 item splits, 80 us per launch on the reference's 2 048-user blocks.)
            const int lcnt = l0 + lane < nl ? (int)g.r.qcnt[(size_t)rb * nl + l0 + lane] : 0;
            over = over || lcnt > g.r.cap_q;
            const int cq = min(lcnt, g.r.cap_q);
            int inc = cq, cmax = cq;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(inc, o, 64);
                if (lane >= o) inc += v;
                cmax = max(cmax, __shfl_xor(cmax, o, 64));
            }
            const int off = n + inc - cq, tot = __shfl(inc, 63, 64);
            const uint64_t* src = g.r.qpool + ((size_t)rb * nl + min(l0 + lane, nl - 1)) * (size_t)(2 * g.r.cap_q);
            for (int i = 0; i < cmax; ++i)
                if (i < cq && off + i < kPool7) pool[off + i] = *reinterpret_cast<const ulonglong2*>(src + 2 * i);
            n += tot;
        }
        over = __any(over) || n > kPool7;
        n = min(n, kPool7);
        pda_wave_sync();
#pragma unroll
        for (int k = 0; k < NKP; ++k) {
            const int i = lane + 64 * k;
            const ulonglong2 kv = i < n ? pool[i] : make_ulonglong2(0ull, 0ull);
            key[k] = kv.x;
            bnd[k] = kv.y;
        }
    }
    // which register groups hold anything (wave-uniform)
    unsigned live = 0u;
#pragma unroll
    for (int k = 0; k < NKP; ++k) live |= __ballot(bnd[k] != 0ull) != 0ull ? 1u << k : 0u;
    // the q-th largest ordered lower bound (q <= n): bitwise descent over ballots -- from the highest bit in which the pool's bounds differ at all (they
    // share sign, exponent and more: ~24 of the 32 steps are left) down to bit `low`.  low > 0 leaves the low bits of the result zero: a value BELOW the
    // q-th largest (in the ordered domain clearing low bits lowers positive and negative floats alike), which q pairs reach all the same.
    uint32_t omx = 0u, omn = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < NKP; ++k) {
        const uint32_t lb = (uint32_t)bnd[k];
        omx = max(omx, lb);
        omn = min(omn, bnd[k] != 0ull ? lb : 0xFFFFFFFFu);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        omx = max(omx, (uint32_t)__shfl_xor((int)omx, o, 64));
        omn = min(omn, (uint32_t)__shfl_xor((int)omn, o, 64));
    }
    omx = (uint32_t)__builtin_amdgcn_readfirstlane((int)omx);
    omn = (uint32_t)__builtin_amdgcn_readfirstlane((int)omn);
    auto qth = [&](int q, int low) __attribute__((always_inline)) -> uint32_t {
        const uint32_t diff = omx ^ omn;                          // (n >= q >= 1: the pool holds a real pair, omn <= omx)
        if (diff == 0u) return omx;
        const int top = 31 - __builtin_clz(diff);
        uint32_t t = top >= 31 ? 0u : (omx & ~((2u << top) - 1u));        // the bits above `top`: common to every real bound
        for (int bit = top; bit >= low; --bit) {
            const uint32_t trial = t | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int k = 0; k < NKP; ++k)
                if (live & (1u << k)) cnt += __popcll(__ballot((uint32_t)bnd[k] >= trial));
            if (cnt >= q) t = trial;
        }
        return t;
    };
    const bool last = g.rank_next <= 0;
    bool failed = false;
    uint32_t keep_o;
    if (!last) {
        const int q = min(g.rank_next, K);
        const float tr = n >= q ? fmaxf(pda_unordf(qth(q, 12)), t_old) : t_old;       // (too few pairs above the old threshold: it stays)
        keep_o = pda_ordf(tr + 0.0f);
        if (lane == 0) {
            g.r.thr[rb] = lowered7(tr);
            g.r.tmax[rb] = tr;
        }
    } else {
        // tk: K pairs reach it (the descent's low bits cleared: a little below the K-th largest lower bound); the bets held iff K pairs reach the
        // thresholds that were used -- counted exactly, not read off the truncated tk
        float tk = -INFINITY;
        if (n >= K) tk = pda_unordf(qth(K, 8));
        const uint32_t t_old_o = pda_ordf(t_old + 0.0f);
        int c_old = 0;
#pragma unroll
        for (int k = 0; k < NKP; ++k)
            if (live & (1u << k)) c_old += __popcll(__ballot(bnd[k] != 0ull && (uint32_t)bnd[k] >= t_old_o));
        failed = n < K || c_old < K;                             // fewer than K pairs reach the thresholds that were used: a bet was lost
        keep_o = pda_ordf(tk + 0.0f);
        if (lane == 0) g.r.tk[rb] = tk;
    }
    const bool mask_here = g.first_launch != 0 && g.hist_indptr != nullptr;       // (a launch against -inf: expand7_kernel left the train items in)
    long long hb = 0, he = 0;
    if (mask_here) {
        const int64_t hr = g.hist_row_mode == PDA_HIST_BY_USER_ID ? (int64_t)g.users[rb] : (int64_t)rb;
        hb = g.hist_indptr[hr];
        he = g.hist_indptr[hr + 1];
    }
    int kept = 0;
#pragma unroll
    for (int k = 0; k < NKP; ++k) {
        if (!(live & (1u << k))) continue;
        bool p = bnd[k] != 0ull && (uint32_t)(bnd[k] >> 32) >= keep_o;
        if (mask_here && p) {
            const int item = g.item_offset + (int)g.pinfo[(unsigned)key[k]][2];
            long long lo = hb, hi2 = he;
            while (lo < hi2) {
                const long long mid = (lo + hi2) >> 1;
                if (g.hist_indices[mid] < item) lo = mid + 1; else hi2 = mid;
            }
            if (lo < he && g.hist_indices[lo] == item) p = false;
        }
        const uint64_t bm = __ballot(p);
        const int slot = kept + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0));
        if (p && slot < kCand7) *reinterpret_cast<ulonglong2*>(g.r.cand + ((size_t)rb * kCand7 + slot) * 2) = make_ulonglong2(key[k], bnd[k]);
        kept += __popcll(bm);
    }
    if (kept > kCand7) { over = true; kept = kCand7; }
    if (lane == 0) {
        g.r.ncand[rb] = kept;
        if (over || failed) g.r.flags[rb] |= over ? 1u : 2u;     // bit 0: a list or the pool overflowed; bit 1: a bet was lost
    }
}

// One wave per block row: the final pool rescored exactly (the fp32 chain of the oracle, oracle/pda_oracle.c dot_chain, as sweep5_kernel's
// rescore_ring), sorted, the best K written.  A row whose lists overflowed, or whose valid bound tk ended below a threshold that was used, goes
// on the fallback list instead.
template <int D, bool BF>
__global__ void __launch_bounds__(256) resolve7_kernel(Sel7 g) {
    constexpr int LPC = D / 32, CPP = 64 / LPC;
    __shared__ uint64_t keys_s[4][kCand7];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_rows = g.e.n_users_dev != nullptr ? min(g.e.n_users_blk, *g.e.n_users_dev) : g.e.n_users_blk;
    const int rb = blockIdx.x * 4 + wave;
    if (rb >= n_rows) return;
    const int K = g.K;
    uint64_t* orow = g.out_keys + (size_t)rb * K;
    const bool failed = g.r.flags[rb] != 0u;
    if (failed) {
        if (lane == 0) {
            const int slot = atomicAdd(g.fail_count, 1);
            g.fail_list[slot] = rb;
        }
        if (lane < K) orow[lane] = 0ull;
        return;
    }
    uint64_t* ks = keys_s[wave];
    const int n = g.r.ncand[rb];
    const int uid = g.users[rb];
    const int q = lane % LPC, ci = lane / LPC;
    f32x4 uu[8];
    const size_t ub = (size_t)uid * D + q * (LPC == 8 ? 32 : 8);
#pragma unroll
    for (int c = 0; c < 8; ++c) uu[c] = pda_load4<BF>(g.U, ub + (LPC == 8 ? 4 * c : 8 * LPC * (c >> 1) + 4 * (c & 1)));
    // the pool's items up front, a lane per candidate (two round trips for the whole row: visiting positions, then their local ids) -- a pass below
    // then starts with its row loads: one round trip per pass instead of three dependent ones
    constexpr int NKC0 = kCand7 / 64;
    int locs[NKC0];
#pragma unroll
    for (int k = 0; k < NKC0; ++k) {
        const int i = lane + 64 * k;
        const unsigned pos = i < n ? (unsigned)g.r.cand[((size_t)rb * kCand7 + i) * 2] : 0u;
        locs[k] = i < n ? (int)g.pinfo[pos][2] : 0;
    }
    for (int base = 0; base < n; base += CPP) {
        const bool have = base + ci < n;
        const int kk = base >> 6;                                // (wave-uniform: CPP divides 64)
        int lsel = locs[0];
#pragma unroll
        for (int k = 1; k < NKC0; ++k) lsel = kk == k ? locs[k] : lsel;
        const int loc = __shfl(lsel, (base & 63) + ci, 64);
        f32x4 ii[8];
        const size_t ib = (size_t)loc * D + q * (LPC == 8 ? 32 : 8);
#pragma unroll
        for (int c = 0; c < 8; ++c) ii[c] = pda_load4<BF>(g.I, ib + (LPC == 8 ? 4 * c : 8 * LPC * (c >> 1) + 4 * (c & 1)));
        auto fma8 = [&](float acc, int cq) __attribute__((always_inline)) -> float {
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                acc = __builtin_fmaf(uu[2 * cq][sidx], ii[2 * cq][sidx], acc);
                acc = __builtin_fmaf(uu[2 * cq + 1][sidx], ii[2 * cq + 1][sidx], acc);
            }
            return acc;
        };
        float o = 0.f, o_other = 0.f;
        if constexpr (LPC == 8) {
            float c0 = 0.f, c1 = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
            for (int ph = 0; ph < LPC; ++ph) {
                o0 = c0;
                o1 = c1;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
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
            o = o1;
            o_other = o0;
        } else if constexpr (LPC == 4) {
            const bool hi = q >= 2;
            auto swap2 = [](float x) __attribute__((always_inline)) -> float {
                return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
            };
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) {
                const float a = fma8(o, cq);
                const float a_sw = swap2(a);
                const float b = fma8(hi ? a_sw : o, cq);
                const float b_sw = swap2(b);
                o = hi ? b : b_sw;
            }
        } else {
            static_assert(LPC == 2, "d = 64");
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) o = fma8(o, cq);
        }
        const float o0 = LPC == 8 ? o_other : __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(o), 0xB1, 0xF, 0xF, true));            // quad_perm [1,0,3,2]
        const float sc = o0 + o;                                 // meaningful on the candidate's last lane
        if (have && q == LPC - 1) ks[base + ci] = (sc >= -INFINITY) ? pda_pack_key(sc, (uint32_t)(g.item_offset + loc)) : 0ull;       // (NaN never ranks)
    }
    pda_wave_sync();
    // rank every key among the n (distinct: the item field) and write the best K; the rest of the row: 0 = empty
    constexpr int NKC = kCand7 / 64;
    uint64_t mine[NKC];
    int rank[NKC];
#pragma unroll
    for (int k = 0; k < NKC; ++k) {
        mine[k] = lane + 64 * k < n ? ks[lane + 64 * k] : 0ull;
        rank[k] = 0;
    }
    for (int i = 0; i < n; ++i) {
        const uint64_t o2 = ks[i];
#pragma unroll
        for (int k = 0; k < NKC; ++k) rank[k] += o2 > mine[k] ? 1 : 0;
    }
    int n_valid = 0;
#pragma unroll
    for (int k = 0; k < NKC; ++k) {
        if (mine[k] != 0ull && rank[k] < K) orow[rank[k]] = mine[k];
        n_valid += __popcll(__ballot(mine[k] != 0ull));
    }
    if (lane >= n_valid && lane < K) orow[lane] = 0ull;
}
// the statistic "pairs rescored exactly" (workspace + 4): the rows' final pool sizes summed -- one atomic per 1 024 rows.  (One per row, from
// resolve7_kernel itself, serialised 262 144 atomics on one address: 3.3 ms of a 17.9 ms call.)
__global__ void __launch_bounds__(1024) stat7_kernel(Rows7 r, int n, const int* __restrict__ n_users_dev, unsigned* __restrict__ stats) {
    __shared__ unsigned part[16];
    const int n_rows = n_users_dev != nullptr ? min(n, *n_users_dev) : n;
    const int i = blockIdx.x * 1024 + threadIdx.x;
    unsigned v = (i < n_rows && r.flags[i] == 0u) ? (unsigned)r.ncand[i] : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += (unsigned)__shfl_xor((int)v, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int k = 0; k < 16; ++k) t += part[k];
        if (t != 0u) atomicAdd(stats + 1, t);
    }
}

// ---- the exact fallback's plumbing: the failed rows' users as a block of their own (padded with the block's first user), and their merged
// lists back into the rows they belong to
__global__ void __launch_bounds__(256) fail_users7_kernel(const int32_t* __restrict__ users, const int* __restrict__ fail_list, const int* __restrict__ fail_count,
                                                          int n, const float* __restrict__ tk, const unsigned* __restrict__ flags, int by_row,
                                                          int32_t* __restrict__ users2, float* __restrict__ seed2, int* __restrict__ n_dev2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *n_dev2 = by_row ? (*fail_count > 0 ? n : 0) : *fail_count;
    if (i >= n) return;
    // The seed of generation 4's sweep: the failed row's tk -- K unmasked items reach it whatever became of the bets (-inf: fewer than K pairs in its pool)
    // -- so the fallback rescans the catalogue against an almost final threshold instead of building a list from nothing; rows that are only there to fill
    // the block: +1e30, nothing qualifies.
    //   by_row = 0 (history by user id): the failed rows are RE-BLOCKED -- row i of the fallback's block is row fail_list[i] -- and the block is as large as the count;
    //   by_row = 1 (history by block row: the reference's per-block COO mask, MF/train_new_api.py:791): the rows keep their places (their mask rows with them); the
    //   whole block is swept again iff anybody failed, the rows that did not fail against +1e30.
    int src = i;
    bool real;
    if (by_row) {
        real = flags[i] != 0u;
    } else {
        real = i < *fail_count;
        src = real ? fail_list[i] : 0;
    }
    users2[i] = users[src];
    float sd = 1.0e30f;
    if (real) {
        const float t = tk[src];
        sd = (t > -1.0e30f && t < 1.0e30f) ? lowered7(t) : -INFINITY;
    }
    seed2[i] = sd;
}
// one wave per failed row: the best K of its S sorted partial lists (K rounds of "the largest head"), written to the row it came from
__global__ void __launch_bounds__(256) fail_merge7_kernel(const uint64_t* __restrict__ keys, int S, int n, int K, const int* __restrict__ fail_list,
                                                          const int* __restrict__ fail_count, int by_row, uint64_t* __restrict__ out_keys, unsigned* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) stats[6] = (unsigned)*fail_count;      // workspace + 24: rows served by the exact fallback
    const int nf = min(n, *fail_count);
    // (a bounded grid striding over the failed rows -- a handful as a rule; one workgroup per four ROWS OF THE BLOCK was 12 500 workgroups that left at once: 26 us at config 2)
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < nf; i += gridDim.x * 4) {
    int cur = 0;                                                                  // lane s < S: the head of list s
    uint64_t* orow = out_keys + (size_t)fail_list[i] * K;
    const size_t krow = by_row ? (size_t)fail_list[i] : (size_t)i;                // (by_row: the fallback's rows kept their places)
    for (int k = 0; k < K; ++k) {
        uint64_t h = 0ull;
        for (int s0 = 0; s0 < S; s0 += 64) {                                      // (S <= 64 in practice: one round)
            const int s = s0 + lane;
            const uint64_t v = (s < S && cur < K) ? keys[((size_t)s * n + krow) * K + cur] : 0ull;
            h = v > h ? v : h;
        }
        uint64_t best = h;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const uint64_t other = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)best, o, 64);
            best = other > best ? other : best;
        }
        if (best == 0ull) {
            if (lane + k < K) orow[lane + k] = 0ull;                              // (all lists exhausted; K <= 64)
            break;
        }
        if (h == best) ++cur;                                                     // (keys are distinct: one lane advances)
        if (lane == 0) orow[k] = best;
    }
    }
}

template <int D, bool BF, int UPW, bool MAXM = false>
int launch_sweep7(const Args7& g, hipStream_t stream) {
    constexpr int UT = 4 * UPW;
    constexpr size_t lds = (size_t)kNSlot5 * slot_bytes5(D) + 64;
    static int attr_set = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep7_kernel<D, BF, UPW, MAXM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PDA_ERR_LAUNCH;
        attr_set = 1;
    }
    const int utiles = (g.n_users_blk + UT - 1) / UT;
    hipLaunchKernelGGL((sweep7_kernel<D, BF, UPW, MAXM>), dim3((unsigned)(utiles * g.n_splits)), dim3(256), lds, stream, g);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}
