// score + mask + top-K, generation 2: bf16x3 MFMA pre-filter + exact fp32 rescoring (same results as v1, bit for bit).
//
// v1 (pda_score_topk.hip) is bound by the fp32 matrix pipe: d/2 v_mfma_f32_32x32x2_f32 of 64 cycles per 32x32 tile,
// and every non-MFMA instruction steals pipe time.  The top-K only needs EXACT scores for the few (user, item) pairs
// that can beat the user's running threshold (~K*ln(I/K) of I).  v2 therefore
//   1. scores every pair approximately with bf16 MFMAs on a hi/lo split of both operands,
//        s~ = uh.ih + uh.il + ul.ih                       (3 x d/16 v_mfma_f32_32x32x16_bf16 of 32 cycles: 5.3x less pipe time)
//   2. bounds the error rigorously:  |s~ - s_exact| <= eps(u,i) = 2^-13 * ||u||_2 * ||i||_2   (derivation below)
//   3. tests the UPPER BOUND  head_ub(s~ + eps) > threshold  in the MFMA shadow,
//   4. pushes the rare survivors (row, item) into a per-wave LDS ring (a few instructions, no serialisation), and
//   5. when the ring fills, rescoring happens lane-parallel: each lane recomputes ONE candidate's score with the
//      exact fp32 fmaf chain of v1 (two chains over even/odd k-chunks -- bitwise what v_mfma_f32_32x32x2_f32 gives),
//      applies the exact head, compares with the exact threshold and appends to the user's LDS list (v1 machinery).
// Because step 5 is exact and step 3 never rejects a pair whose exact head beats the threshold, v2 returns the
// same packed keys as v1 (tests/test_gpu_score_topk.py checks equality with v1 and with the oracle).
//
// Revision (approximate lists): measured, 35 % of v2's time was steps 4-5 -- ~550 exact rescorings per user, each a
// 1 KB gather plus a 128-long dependent fmaf chain.  Only ~K of them matter at the end.  The sweep now keeps the
// per-user lists keyed by the APPROXIMATE head h~ = head(s~) and a per-row half-width W >= |head(s_exact) - h~|
// (head is pop-Lipschitz in s:  W = nu_row * max||i|| * max(pop), plus 2^-16 |h~| for the fp32 evaluation):
//   * threshold of the fast test:  T = h~_(K) - W      (a lower bound of the exact K-th best so far)
//   * a list entry is dropped at compaction only if   h~ < h~_(K) - 2W   (it can no longer reach the exact top K)
//   * at the end of the sweep each row's <= 58 survivors are rescored exactly (the v1 fmaf chains) and sorted.
// If more than 58-K entries ever sit inside a row's 2W band (massive near-ties), the row's user tile is flagged in
// `workspace` and recomputed by the exact v1 kernel, launched right behind on the same stream (it exits immediately
// for unflagged tiles).  The returned keys are therefore still exactly v1's.
//
// Error bound.  bf16 keeps 8 significant bits (RNE): x = xh + xl + xr with |x-xh| <= 2^-8|x|, |xr| <= 2^-16|x|.
//   u.i - (uh.ih + uh.il + ul.ih) = ul.il + ur.i + (uh+ul).ir   =>  |.| <= (2^-16 + 2^-16 + 2^-16(1+2^-8)) sum|u_k i_k|
//   the bf16 products are exact in fp32; their fp32 accumulation (3d/16*16 adds, any order) errs by <= 3d*2^-24 sum|.|
//   the exact chain itself differs from the real dot by <= d*2^-24 sum|u_k i_k|.
//   With d <= 256: total <= (3.1*2^-16 + 1024*2^-24) sum|u_k i_k| < 2^-13.6 ||u|| ||i||  (Cauchy-Schwarz); we use 2^-13 and
//   norms rounded up by (1+2^-10).  The fp32 evaluation of s~+eps and of the head bound is covered by comparing against
//   threshold*(1 -/+ 2^-20).
#include "pda_topk_common.h"
#include <cstdlib>

using namespace pda_topk;

namespace {

constexpr int kCap2 = PDA_TOPK_CAP - 2;  // 58 slots per user list
constexpr float kEpsScale = 1.220703125e-4f;  // 2^-13

#ifdef PDA_ABLATION
__device__ unsigned long long pda_dbg[8];   // per-wave sums: 0 ring entries, 1 cycles in process_ring, 2 cycles in compactions, 3 cycles in push, 4 total cycles, 5 waves
#define PDA_T0(v) const long long v = __builtin_readcyclecounter()
#define PDA_T1(v, acc) acc += __builtin_readcyclecounter() - v
#else
#define PDA_T0(v) do {} while (0)
#define PDA_T1(v, acc) do {} while (0)
#endif

// one row per D/8 threads: fp32 -> bf16 hi, bf16 lo, padded norm
// BF: the table is bf16 already -- there is nothing to split; `hi` (may be NULL) receives the row when it has to be
// gathered into visiting order, `lo` is unused.
template <int D, bool BF>
__global__ void __launch_bounds__(256) item_prep_kernel(const void* __restrict__ I, int n, uint16_t* __restrict__ hi,
                                                        uint16_t* __restrict__ lo, float* __restrict__ nrm, int* __restrict__ nrm_max_bits,
                                                        const int* __restrict__ order, const float* __restrict__ pop,
                                                        float* __restrict__ pop_p, int* __restrict__ pos_of, int* __restrict__ bad,
                                                        uint16_t* __restrict__ bex) {
    constexpr int TPR = D / 8;
    const int row = blockIdx.x * (256 / TPR) + threadIdx.x / TPR, e = threadIdx.x % TPR;
    float ss = 0.f, popv = 1.0f;
    if (row < n) {
        int src = row;
        if (pop) popv = pop[order ? min(max(order[row], 0), n - 1) : row];
        if (order) {                      // ordered prep: position `row` holds item order[row]
            src = order[row];
            if (src < 0 || src >= n) { if (e == 0) atomicOr(bad, 1); src = 0; }
            else if (e == 0) {
                if (atomicExch(&pos_of[src], row) != -1) atomicOr(bad, 1);   // not a permutation
                if (pop) pop_p[row] = pop[src];
            }
        }
        const f32x4 a = pda_load4<BF>(I, (size_t)src * D + 8 * e);
        const f32x4 b = pda_load4<BF>(I, (size_t)src * D + 8 * e + 4);
        if constexpr (BF) {
            if (hi) *reinterpret_cast<u32x4*>(hi + (size_t)row * D + 8 * e) =
                        *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(I) + (size_t)src * D + 8 * e);
        } else {
            u32x4 h, l;
            split8(a, b, h, l);
            *reinterpret_cast<u32x4*>(hi + (size_t)row * D + 8 * e) = h;
            *reinterpret_cast<u32x4*>(lo + (size_t)row * D + 8 * e) = l;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) ss += a[k] * a[k] + b[k] * b[k];
    }
#pragma unroll
    for (int o = TPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (row < n && e == 0) {
        const float v = sqrtf(ss) * 1.0009765625f * 1.0001f;
        nrm[row] = v;
        atomicMax(nrm_max_bits, __float_as_int(v));   // v >= 0: integer order == float order
        // The extra k-step of v3's folded test (pda_score_topk_v3.hip): with the A side holding the pieces of -threshold, -1
        // and +eps_scale, one more MFMA turns the accumulator into  s~ - thr * (1/pop)' + (1 + eps)  -- "candidate" is then
        // "accumulator > 0".  (1/pop)' is rounded DOWN and capped (conservative for thr > 0; thr <= 0 takes the kernel's
        // pop > thr path), split into three bf16 pieces so that the eight products carry thr/pop to 2^-30.
        uint32_t p1 = 0x3F80u, p2 = 0, p3 = 0, k1 = 0, k2 = 0;   // PDA_HEAD_RAW: 1/pop := 1, constant := +8e-6 slack
        if (pop) {
            float ip = (popv > 0.f) ? fminf((1.0f / popv) * 0.9999995f, 1.0e6f) : 1.0e6f;
            bf16_split3(ip, p1, p2, p3);
            k1 = 0x3F80u;
            k2 = bf16_up(8.0e-6f);
            if (!(popv == popv)) k1 = 0xFF61u;   // NaN popularity: never a candidate (the exact kernels' comparisons are false too)
        } else {
            k1 = bf16_up(8.0e-6f);
        }
        u32x4 lo4, hi4;
        lo4[0] = p1 | (p2 << 16);
        lo4[1] = p1 | (p2 << 16);
        lo4[2] = p3 | (p1 << 16);
        lo4[3] = p3 | (p2 << 16);
        hi4[0] = k1 | (k2 << 16);
        hi4[1] = bf16_up(v);
        hi4[2] = 0;
        hi4[3] = 0;
        *reinterpret_cast<u32x4*>(bex + (size_t)row * 16) = lo4;
        *reinterpret_cast<u32x4*>(bex + (size_t)row * 16 + 8) = hi4;
    }
}

// per-tile maxima of |pop| and |pop|*||i|| in visiting order, then their suffix maxima (single workgroup)
__global__ void __launch_bounds__(256) tile_bound_kernel(const float* __restrict__ pop_p, const float* __restrict__ nrm, int n, int n_tiles,
                                                         float* __restrict__ tA, float* __restrict__ tB) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    float ma = 0.f, mb = 0.f;
    for (int q = 0; q < 32; ++q) {
        const int i = t * 32 + q;
        if (i < n) {
            const float pa = pop_p ? fabsf(pop_p[i]) : 0.f, pb = pop_p ? pa * nrm[i] * 1.000001f : nrm[i];
            ma = fmaxf(ma, pa);
            mb = fmaxf(mb, pb);
        }
    }
    tA[t] = ma;
    tB[t] = mb;
}
__global__ void __launch_bounds__(1024) suffix_max_kernel(float* __restrict__ tA, float* __restrict__ tB, int n_tiles) {
    __shared__ float sa[1024], sb[1024];
    const int per = (n_tiles + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, n_tiles);
    float ma = 0.f, mb = 0.f;
    for (int t = lo; t < hi; ++t) { ma = fmaxf(ma, tA[t]); mb = fmaxf(mb, tB[t]); }
    sa[threadIdx.x] = ma;
    sb[threadIdx.x] = mb;
    __syncthreads();
    float ra = 0.f, rb = 0.f;                      // max over the chunks behind mine
    for (int q = threadIdx.x + 1; q < 1024; ++q) { ra = fmaxf(ra, sa[q]); rb = fmaxf(rb, sb[q]); }
    for (int t = hi - 1; t >= lo; --t) {
        ra = fmaxf(ra, tA[t]);
        rb = fmaxf(rb, tB[t]);
        tA[t] = ra;
        tB[t] = rb;
    }
}

// History rows rewritten in visiting positions (in-shard item ids -> item_offset + pos_of[id - item_offset]) and sorted
// again.  One wave per row: bitonic sort in LDS up to 2048 entries, rank sort in global memory beyond.
__global__ void __launch_bounds__(64) hist_reorder_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                          const int* __restrict__ pos_of, int item_offset, int n_items_local,
                                                          int32_t* __restrict__ out) {
    __shared__ int buf[2048];
    const int lane = threadIdx.x;
    const int64_t b = indptr[blockIdx.x], e = indptr[blockIdx.x + 1];
    const int64_t L = e - b;
    if (L <= 0) return;
    auto mapped = [&](int64_t i) __attribute__((always_inline)) {
        const int v = indices[b + i];
        const int loc = v - item_offset;
        return (loc >= 0 && loc < n_items_local) ? item_offset + pos_of[loc] : v;
    };
    if (L <= 2048) {
        int P = 64;
        while (P < L) P <<= 1;
        for (int i = lane; i < P; i += 64) buf[i] = i < L ? mapped(i) : 0x7fffffff;
        __syncthreads();
        for (int k = 2; k <= P; k <<= 1)
            for (int jst = k >> 1; jst > 0; jst >>= 1) {
                for (int q = lane; q < P / 2; q += 64) {
                    const int i = ((q & ~(jst - 1)) << 1) | (q & (jst - 1)), p = i | jst;
                    const int x = buf[i], y = buf[p];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { buf[i] = y; buf[p] = x; }
                }
                __syncthreads();
            }
        for (int i = lane; i < L; i += 64) out[b + i] = buf[i];
    } else {
        for (int64_t i = lane; i < L; i += 64) {
            const int v = mapped(i);
            int64_t rank = 0;
            for (int64_t q = 0; q < L; ++q) {
                const int w = mapped(q);
                rank += (w < v || (w == v && q < i)) ? 1 : 0;
            }
            out[b + rank] = v;
        }
    }
}

__global__ void __launch_bounds__(256) pop_max_kernel(const float* __restrict__ pop, int n, int* __restrict__ out_bits) {
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = fmaxf(m, fabsf(pop[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_int(m));
}

// Compaction of one row's APPROXIMATE list: sort (sorted prefix + new tail), keep everything inside the 2W band under the
// K-th entry, publish the lower-bound threshold T = h~_(K) - W.  Returns true if the band does not fit (tile must be redone).
template <int CAP>
__device__ __forceinline__ void compact_band(uint64_t* buf, int* cnt_slot, float* tau_slot, int* sorted_slot, float* drop_slot, float wabs, int K, int lane) {
    pda_wave_sync();
    const int c = min(__builtin_amdgcn_readfirstlane(*cnt_slot), CAP);
    const int n0 = min(__builtin_amdgcn_readfirstlane(*sorted_slot), c);
    uint64_t key = lane < c ? buf[lane] : (uint64_t)(63 - lane);
    int rank;
    if (n0 > 0) {
        rank = lane < n0 ? lane : 0;
        const uint64_t oldmask = n0 >= 64 ? ~0ull : ((1ull << n0) - 1ull);
        for (int jj = n0; jj < c; ++jj) {
            const uint64_t kj = pda_readlane_u64(key, jj);
            rank += (kj > key) ? 1 : 0;
            const int olds_above = __popcll(__ballot(key > kj) & oldmask);
            rank += (lane == jj) ? olds_above : 0;
        }
    } else {
        rank = 0;
        for (int jj = 0; jj < c; ++jj) {
            const uint64_t kj = pda_readlane_u64(key, jj);
            rank += (kj > key) ? 1 : 0;
        }
    }
    pda_wave_sync();
    int keep = c;
    if (c >= K) {
        const uint64_t mk = __ballot(lane < c && rank == K - 1);
#ifdef PDA_ABLATION
        if (lane == 0 && __popcll(mk) != 1) atomicAdd(&pda_dbg[6], 1ull);
#endif
        const float vK = pda_unordf((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), __builtin_ctzll(mk)));
        const float w = (wabs + fabsf(vK) * 1.52587890625e-5f) * 1.001f;
        const float band = vK - 2.0f * w;
        keep = __popcll(__ballot(lane < c && pda_key_val(key) >= band));   // ranks 0..keep-1: the list is ordered by value
        if (keep > CAP - 1) {
            // The band does not fit: keep the best CAP-1 and remember how high the dropped ones could still reach.
            // Whether that matters is decided at the END against the exact K-th value (early-sweep clusters, when the
            // running threshold is still low, never do).
#ifdef PDA_ABLATION
            if (lane == 0) atomicAdd(&pda_dbg[7], 1ull);
#endif
            keep = CAP - 1;
            const uint64_t md = __ballot(lane < c && rank == keep);          // best dropped entry
            const float vd = pda_unordf((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), __builtin_ctzll(md)));
            if (lane == 0) *drop_slot = fmaxf(*drop_slot, vd + (wabs + fabsf(vd) * 1.52587890625e-5f) * 1.001f);
        }
        if (lane == 0) *tau_slot = vK - w;
    }
    if (lane < c && rank < keep) buf[rank] = key;
    if (lane == 0) {
        *cnt_slot = keep;
        *sorted_slot = keep;
    }
    pda_wave_sync();
}

// ORD (ordered sweep with exact early termination): items are visited in the caller's `order` (planes, norms, pop and the
// history CSR are all stored in visiting positions).  For a row with padded norm ||u|| every item at position >= 32 t has
//     head(s) <= (1 + max(s,0)) |pop| <= |pop| + ||u|| (|pop| ||i||) <= sufA[t] + ||u|| sufB[t]        (Cauchy-Schwarz)
// (PDA_HEAD_RAW: s <= ||u|| sufB[t]).  Once that bound is below the row's lower-bound threshold T for all 128 rows of
// the workgroup, nothing behind t can enter any list and the sweep stops.  Any `order` is correct; popular-first makes
// the suffix bounds fall quickly, which is how PDA's popularity-weighted head lets most of the catalogue go unscored.
// Item splits take interleaved tiles (t = split, split + n_splits, ...) so that every split sees the strong items early.
// NP = MFMAs per k-step: 3 for fp32 tables (bf16 hi/lo split, above), 1 for bf16 tables (pda_score_topk_bf16): the products
// of two bf16 are exact in fp32, so one v_mfma_f32_32x32x16_bf16 differs from the exact fmaf chain only by the order of the
// fp32 additions: |s~ - s_chain| <= 2 d 2^-24 sum|u_k i_k| <= 2^-15 ||u|| ||i|| for d <= 256; 2^-14 is used.
template <int D, int HEAD, bool ORD, int ABL = 0, int NP = 3>   // ABL: profiling-only (-DPDA_ABLATION): 1 drop candidates, 2 skip the test, 4 no history
__global__ void __launch_bounds__(kThreads, 2) score_topk_v2_kernel(ScoreArgs2 aa) {
    constexpr bool BF = NP == 1;
    constexpr float kEps = BF ? 6.103515625e-5f : kEpsScale;
    const ScoreArgs& a = aa.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CPR = D / 8;                 // 16-byte chunks per bf16 row
    constexpr int NM = D / 16;                 // MFMA k-steps
    constexpr int NLD = (32 * CPR) / kThreads; // 16-byte loads per thread per tile, per plane (hi / lo)
    static_assert(NLD >= 1, "v2 needs embed dim >= 64");
    uint16_t* Bh = reinterpret_cast<uint16_t*>(smem);              // [32][D] bf16 hi, swizzled
    uint16_t* Bl = Bh + 32 * D;                                    // [32][D] bf16 lo (NP == 3 only)
    uint64_t* lists = reinterpret_cast<uint64_t*>(smem + (BF ? 1 : 2) * 32 * D * sizeof(uint16_t));   // [128][kCap2]
    int* cntl = reinterpret_cast<int*>(lists + (size_t)kUserTile * kCap2);                 // [128]
    float* taul = reinterpret_cast<float*>(cntl + kUserTile);                              // [128]
    int* sortl = reinterpret_cast<int*>(taul + kUserTile);                                 // [128] length of the sorted prefix
    float* dropl = reinterpret_cast<float*>(sortl + kUserTile);                            // [128] highest reach of entries dropped by a band overflow
    int* votes = reinterpret_cast<int*>(dropl + kUserTile);                                // [4] ORD: wave w sees no use in going on

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int split = blockIdx.x % a.n_splits, utile = blockIdx.x / a.n_splits;
    const int K = a.K;
    const int tiles_total = (a.n_items_local + 31) >> 5;
    // this workgroup's tiles: t0, t0 + stride, ... (nt of them)
    int t0, stride, nt;
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
    auto tile_of = [&](int k) __attribute__((always_inline)) { return t0 + k * stride; };

    const int row_blk = utile * kUserTile + wave * 32 + j;
    const bool row_ok = row_blk < a.n_users_blk;
    const int uid = row_ok ? a.users[row_blk] : 0;

    // ---- A operand: this lane's user row (k = 16m + 8h .. +7), split into bf16 hi / lo; padded row norm ----------
    u32x4 ah[NM], al[NM];
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
            if constexpr (BF) {          // the row IS bf16: keep its bits (the truncating repack is exact)
                const u32x4 raw = {(__float_as_uint(x[0]) >> 16) | (__float_as_uint(x[1]) & 0xFFFF0000u),
                                   (__float_as_uint(x[2]) >> 16) | (__float_as_uint(x[3]) & 0xFFFF0000u),
                                   (__float_as_uint(y[0]) >> 16) | (__float_as_uint(y[1]) & 0xFFFF0000u),
                                   (__float_as_uint(y[2]) >> 16) | (__float_as_uint(y[3]) & 0xFFFF0000u)};
                ah[m] = raw;
                al[m] = raw;             // unused
            } else {
                split8(x, y, ah[m], al[m]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) ss += x[k] * x[k] + y[k] * y[k];
        }
        ss += __shfl_xor(ss, 32, 64);
        nu_row = sqrtf(ss) * 1.0009765625f * 1.0001f * kEps;   // eps(u,i) = nu_row * I_norm[i]
    }

    // ---- history cursor (as v1) ------------------------------------------------------------------------------
    int64_t hp = 0, he = 0;
    int nxt = 0x7fffffff, nxt2 = 0x7fffffff, pend_v = 0x7fffffff;
    bool pend_flag = false, pend_ok = false;
    const bool hist_on = a.hist_indptr != nullptr && !(ABL & 4);
    if (hist_on && row_ok) {
        const int64_t hr = a.hist_row_mode == PDA_HIST_BY_USER_ID ? (int64_t)uid : (int64_t)row_blk;
        hp = a.hist_indptr[hr];
        he = a.hist_indptr[hr + 1];
        const int lo_item = a.item_offset + t0 * 32;
        int64_t lo = hp, hi = he;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (a.hist_indices[mid] < lo_item) lo = mid + 1; else hi = mid;
        }
        hp = lo;
        if (hp < he) nxt = a.hist_indices[hp];
        if (hp + 1 < he) nxt2 = a.hist_indices[hp + 1];
    }
    auto hist_bits = [&](int t) __attribute__((always_inline)) -> uint32_t {
        if (!hist_on) return 0u;
        const int jg0 = a.item_offset + t * 32, jg1 = jg0 + 32;
        if (!__any(nxt < jg1)) return 0u;      // no row of the wave has an entry in this tile (see pda_score_topk_v3.hip)
        nxt2 = pend_flag ? (pend_ok ? pend_v : 0x7fffffff) : nxt2;
        const bool adv = nxt < jg1;
        uint32_t hb = (adv && (!ORD || nxt >= jg0)) ? (1u << ((nxt - jg0) & 31)) : 0u;
        hp += adv ? 1 : 0;
        nxt = adv ? nxt2 : nxt;
        const int64_t idx = hp + 1;
        pend_ok = idx < he;
        pend_flag = adv;
        // never predicated (exact vmcnt bookkeeping), but lanes that did not advance all read element 0: one cache
        // line for the wave instead of a 64-line gather on every tile
        const int64_t idc = adv ? max((int64_t)0, min(idx, he - 1)) : (int64_t)0;
        pend_v = a.hist_indices[idc];
        if (__builtin_expect(__any(nxt < jg1), 0)) {
            do {
                if (nxt < jg1) {
                    const int nn2 = pend_flag ? (pend_ok ? pend_v : 0x7fffffff) : nxt2;
                    if (!ORD || nxt >= jg0) hb |= 1u << ((nxt - jg0) & 31);
                    ++hp;
                    nxt = nn2;
                    nxt2 = (hp + 1 < he) ? a.hist_indices[hp + 1] : 0x7fffffff;
                    pend_flag = false;
                }
            } while (__any(nxt < jg1));
        }
        return hb;
    };

    // ---- per-row state in LDS -------------------------------------------------------------------------------
    if (lane < 32) {
        cntl[wave * 32 + lane] = 0;
        sortl[wave * 32 + lane] = 0;
        dropl[wave * 32 + lane] = -INFINITY;
        taul[wave * 32 + lane] = row_ok ? -INFINITY : INFINITY;
    }
    pda_wave_sync();
    // half-width of the head interval of this lane's row (absolute part):  |head(s_exact) - head(s~)| <= W
    float w_abs = nu_row * aa.I_norm_max[0] * 1.004f;
    if constexpr (HEAD == PDA_HEAD_POP) w_abs *= aa.pop_max[0];
    bool ovf_tile = false;   // wave-uniform: some row's near-tie band overflowed -> tile recomputed by v1
    f32x16 thr;    // lower-bound thresholds T, lowered by a 2^-20 relative margin (fp32 evaluation of the bound)
    f32x16 nu;     // eps scale of the row behind each accumulator register
    auto refresh_thr = [&]() __attribute__((always_inline)) {
        int hv = h;
        asm volatile("" : "+v"(hv));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float tq = taul[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hv];
            thr[r] = (tq == INFINITY || tq == -INFINITY) ? tq : tq - fabsf(tq) * 9.5367431640625e-7f;
        }
    };
    refresh_thr();
    if constexpr (ORD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) nu[r] = __shfl(nu_row, (r & 3) + 8 * (r >> 2) + 4 * h, 64);
    }
    // the fast test uses ONE eps scale per wave (the largest row norm): eps is ~1e-4 of the score spread, so the few
    // per cent more candidates are free, and the test needs no per-register eps any more
    float nu_max = nu_row;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nu_max = fmaxf(nu_max, __shfl_xor(nu_max, o, 64));
    nu_max *= 1.001f;

    uint64_t* my_lists = lists + (size_t)(wave * 32) * kCap2;
    long long dbg_entries = 0, dbg_proc = 0, dbg_comp = 0, dbg_push = 0;
    PDA_T0(t_all);
    (void)dbg_entries; (void)dbg_proc; (void)dbg_comp; (void)dbg_push;

    // ---- item tile staging (register prefetch, unconditional clamped loads) ------------------------------------
    // (a second register set, i.e. prefetching two tiles ahead, was measured: no gain -- the loads are not what waves wait for)
    u32x4 pA_h[NLD], pA_l[NLD];
    auto tile_load = [&](int t, u32x4 (&ph)[NLD], u32x4 (&pl)[NLD]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int id = tid + kThreads * q;
            const int jj = id / CPR, ch = id % CPR;
            const int it = min(t * 32 + jj, a.n_items_local - 1);
            ph[q] = *reinterpret_cast<const u32x4*>(aa.I_hi + (size_t)it * D + 8 * ch);
            if constexpr (!BF) pl[q] = *reinterpret_cast<const u32x4*>(aa.I_lo + (size_t)it * D + 8 * ch);
        }
    };
    auto tile_store = [&](const u32x4 (&ph)[NLD], const u32x4 (&pl)[NLD]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int id = tid + kThreads * q;
            const int jj = id / CPR, ch = id % CPR;
            const int off = jj * D + 8 * (ch ^ swzb<D>(jj));
            *reinterpret_cast<u32x4*>(Bh + off) = ph[q];
            if constexpr (!BF) *reinterpret_cast<u32x4*>(Bl + off) = pl[q];
        }
    };
    auto lane_consts = [&](int t, float& popv, float& niv, int& idv) __attribute__((always_inline)) {
        const int it = min(t * 32 + j, a.n_items_local - 1);
        niv = aa.I_norm[it];
        popv = 1.0f;
        if constexpr (HEAD == PDA_HEAD_POP) popv = ORD ? aa.pop_p[it] : a.pop[it];
        if constexpr (ORD) idv = a.item_offset + aa.order[it];      // the list keeps the item's real id
        else idv = a.item_offset + t * 32 + j;
    };

    // ---- slow path: flagged lanes append (h~, item) to their row's list, lane-parallel --------------------------------
    // `m`: bit 15-r <-> accumulator register r of the previous tile.  Every flagged lane handles its own top flagged
    // register per round (usually one round); row, history bits and s~ are per-lane values, so there is no
    // wave-uniform loop over registers.
    auto push_flagged = [&](uint32_t m, uint32_t hb, int item_id, float popv, const f32x16& accv) __attribute__((always_inline)) {
        PDA_T0(tq);
        const bool any_hb = __any(hb != 0);
        bool compacted = false;
        while (__any(m != 0)) {
            const bool act = m != 0;
            const int bit = 31 - __builtin_clz(m | 1u);
            const int r = 15 - bit;
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            m &= ~(1u << bit);
            bool p = act;
            if (any_hb) {
                const uint32_t hbr = (uint32_t)__shfl((int)hb, row, 64);           // train items never enter
                if ((hbr >> j) & 1u) p = false;
            }
            // s~ of MY register r.  Common case: one flagged lane in the wave -> r is wave-uniform -> one indexed read.
            float sv;
            const uint64_t actm = __ballot(act);
            if ((actm & (actm - 1ull)) == 0ull) {
                sv = accv[__builtin_amdgcn_readlane(r, __builtin_ctzll(actm))];
            } else {
                sv = accv[0];
#pragma unroll
                for (int k = 1; k < 16; ++k) sv = (r == k) ? accv[k] : sv;
            }
            float hv = sv;
            if constexpr (HEAD == PDA_HEAD_POP) hv = (sv > 0.0f ? sv + 1.0f : __expf(sv)) * popv;
            const int lrow = wave * 32 + row;
            const uint64_t key = pda_pack_key(hv, (uint32_t)item_id);
            for (;;) {
                bool ov = false;
                if (p) {
                    const int slot = atomicAdd(&cntl[lrow], 1);
                    if (slot < kCap2) lists[(size_t)lrow * kCap2 + slot] = key;
                    else ov = true;
                }
                if (!__any(ov)) break;
                pda_wave_sync();
                uint64_t full = __ballot(lane < 32 && cntl[wave * 32 + (lane & 31)] >= kCap2);
                while (full) {
                    const int rr = __builtin_ctzll(full);
                    full &= full - 1ull;
                    PDA_T0(tc);
                    compact_band<kCap2>(my_lists + rr * kCap2, &cntl[wave * 32 + rr], &taul[wave * 32 + rr], &sortl[wave * 32 + rr],
                                        &dropl[wave * 32 + rr], pda_readlane_f32(w_abs, rr), K, lane);
                    PDA_T1(tc, dbg_comp);
                }
                compacted = true;
                const float hv_hi = hv + __shfl(w_abs, row, 64) * 1.001f + fabsf(hv) * 1.52587890625e-5f;
                p = ov && (hv_hi >= taul[lrow]);        // still able to reach the top K of its row?
            }
        }
        if (compacted) refresh_thr();
        PDA_T1(tq, dbg_push);
    };

    // The fast test leaves one 64-bit lane mask per accumulator register in SGPRs (v_cmp writing an SGPR pair: 3 VALU per
    // register instead of 5, the OR over registers is SALU).  The slow path rebuilds the per-lane bit mask from them.
    auto push_masks = [&](const uint64_t (&M)[16], uint64_t okm, uint32_t hb, int item_id, float popv, const f32x16& accv) __attribute__((always_inline)) {
        uint32_t m = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) m |= ((M[r] >> lane) & 1ull) ? (1u << (15 - r)) : 0u;
        m = ((okm >> lane) & 1ull) ? m : 0u;
        push_flagged(m, hb, item_id, popv, accv);
    };
    // the fast test of one accumulator register: lanes whose head upper bound can beat the row's threshold.
    //   PDA head:  (max(s~ + eps, 0) + 1) pop > T   <=>   max(s~, -eps) > T / pop - 1 - eps        (pop > 0; pop = 0 never passes)
    //   raw head:  s~ + eps > T
    // per tile and lane: neg_eps = -eps, ipop <= 1/pop, cc = -1 - eps;  per register: v_max, v_fma, v_cmp (to SGPRs)
    auto test_reg = [&](float sacc, float t, float neg_eps, float ipop, float cc) __attribute__((always_inline)) -> uint64_t {
        if constexpr (HEAD == PDA_HEAD_POP) return __ballot(fmaxf(sacc, neg_eps) > __builtin_fmaf(t, ipop, cc));
        else return __ballot(sacc > t + neg_eps);
    };
    auto test_consts = [&](float popv, float niv, float& neg_eps, float& ipop, float& cc) __attribute__((always_inline)) {
        neg_eps = -(nu_max * niv * 1.001f + 3e-6f);        // 3e-6: the roundings of v_fma / v_rcp on values of order <= 10
        ipop = 0.f;
        cc = 0.f;
        if constexpr (HEAD == PDA_HEAD_POP) {
            ipop = __builtin_amdgcn_rcpf(popv) * 0.9999995f;   // <= 1/pop: a smaller right-hand side only lets more through
            cc = -1.0f + neg_eps;
        }
    };

    // ---- main loop (software pipeline as v1: test of tile t-1 in the shadow of the MFMAs of tile t) ----------------
    f32x16 acc_prev = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint32_t hb_prev = 0, hb_cur = 0;
    float pop_prev = 0.f, pop_cur = 0.f, ni_prev = 0.f, ni_cur = 0.f;
    int id_prev = 0, id_cur = 0;
    bool ok_prev = false, ok_cur = false;   // lane's item exists

    if (nt > 0) {
        tile_load(t0, pA_h, pA_l);
        lane_consts(t0, pop_cur, ni_cur, id_cur);
        tile_store(pA_h, pA_l);
        hb_cur = hist_bits(t0);
        ok_cur = (t0 * 32 + j) < a.n_items_local;
    }
    if (tid < 4) votes[tid] = 0;
    __syncthreads();

    const uint16_t* bhrow = Bh + j * D;
    const uint16_t* blrow = Bl + j * D;
    const int bsw = swzb<D>(j);

    // One iteration.  `cur` holds tile t+1 (loaded one iteration ago, stored at the end of this one);
    // `nxt` receives tile t+2.
    auto iteration = [&](int k, u32x4 (&cur_h)[NLD], u32x4 (&cur_l)[NLD]) __attribute__((always_inline)) -> bool {
        const bool has_next = (k + 1) < nt;
        const int tn = tile_of(min(k + 1, nt - 1));
        float pop_next, ni_next;
        int id_next;
        if constexpr (!(ABL & 8)) tile_load(tn, cur_h, cur_l);
        lane_consts(tn, pop_next, ni_next, id_next);
        __builtin_amdgcn_sched_barrier(0);

        f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 acc1 = acc0;
        uint64_t M[16];
        float neg_eps, ipop, cc;
        test_consts(pop_prev, ni_prev, neg_eps, ipop, cc);
#pragma unroll
        for (int mm = 0; mm < NM; ++mm) {
            const int off = 8 * ((2 * mm + h) ^ bsw);
            bf16x8 bh, bl;
            if constexpr (ABL & 16) {
                bh = __builtin_bit_cast(bf16x8, ah[(mm + 1) % NM]);
                bl = __builtin_bit_cast(bf16x8, al[(mm + 1) % NM]);
            } else {
                bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bhrow + off));
                if constexpr (!BF) bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(blrow + off));
                else bl = bh;
            }
            const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[mm]);
            const bf16x8 xl = __builtin_bit_cast(bf16x8, al[mm]);
            if constexpr (BF) {
                if (mm & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh, acc0, 0, 0, 0);
            } else if (mm & 1) {
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh, acc1, 0, 0, 0);
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh, acc0, 0, 0, 0);
            }
            // upper-bound test of 16/NM registers of the previous tile
#pragma unroll
            for (int r = (16 * mm) / NM; r < (16 * (mm + 1)) / NM; ++r) {
                if constexpr (ABL & 2) M[r] = 0;
                else M[r] = test_reg(acc_prev[r], thr[r], neg_eps, ipop, cc);
            }
        }
        const f32x16 acc_new = acc0 + acc1;   // summed here so that only 16 accumulator registers stay live across the slow path
        uint64_t many = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) many |= M[r];
        const uint64_t okm = __ballot(ok_prev);
        many &= okm;
        if constexpr (ABL & 2) asm volatile("" ::"v"(acc_prev[0]), "v"(acc_prev[5]), "v"(acc_prev[15]));
        if constexpr (ABL & 1) { asm volatile("" ::"s"(many)); many = 0; }

        if constexpr (!(ABL & 32)) __syncthreads();  // every wave is done reading the tile
        uint32_t hb_next = 0;
        if (has_next) {
            if constexpr (!(ABL & 8)) tile_store(cur_h, cur_l);
            PDA_T0(th);
            hb_next = hist_bits(tn);
            PDA_T1(th, dbg_entries);
        }
        if (many) push_masks(M, okm, hb_prev, id_prev, pop_prev, acc_prev);
        bool stop = false;
        if constexpr (ORD) {
            // every 4th tile: can anything at or behind the next tile still reach one of my rows?  (see the kernel comment)
            if ((k & 3) == 3 && has_next && aa.sufA != nullptr) {
                const float sa = aa.sufA[tn], sb = aa.sufB[tn];
                bool dead = true;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float ub = __builtin_fmaf(nu[r] * (1.0f / kEps), sb, sa) * 1.000002f;
                    dead = dead && (ub < thr[r]);
                }
                const bool alldead = __all(dead);
                if (lane == 0) votes[wave] = alldead ? 1 : 0;
            }
        }
        if constexpr (!(ABL & 32)) __syncthreads();  // next tile visible
        if constexpr (ORD) {
            if ((k & 3) == 3 && has_next && aa.sufA != nullptr) stop = (votes[0] & votes[1] & votes[2] & votes[3]) != 0;
        }

        acc_prev = acc_new;
        hb_prev = hb_cur;
        pop_prev = pop_cur;
        ni_prev = ni_cur;
        ok_prev = ok_cur;
        hb_cur = hb_next;
        pop_cur = pop_next;
        ni_cur = ni_next;
        id_prev = id_cur;
        id_cur = id_next;
        ok_cur = has_next && (tn * 32 + j) < a.n_items_local;
        return stop;
    };
    int n_done = 0;
    for (int k = 0; k < nt; ++k) {
        ++n_done;
        if (iteration(k, pA_h, pA_l)) break;
    }
    if (tid == 0) atomicAdd(aa.visited, (unsigned long long)n_done);
    if (nt > 0) {   // drain the last tile
        uint64_t M[16];
        float neg_eps, ipop, cc;
        test_consts(pop_prev, ni_prev, neg_eps, ipop, cc);
#pragma unroll
        for (int r = 0; r < 16; ++r) M[r] = test_reg(acc_prev[r], thr[r], neg_eps, ipop, cc);
        push_masks(M, __ballot(ok_prev), hb_prev, id_prev, pop_prev, acc_prev);
    }

    // ---- finalise: exact rescoring of each row's survivors (one lane per candidate, the fmaf chains of v1), exact
    //      sort, K packed keys out.  ~58 candidates x 32 rows per wave: <1 % of the sweep.
    PDA_T0(tp);
    for (int rr = 0; rr < 32; ++rr) {
        const int rb = utile * kUserTile + wave * 32 + rr;
        if (rb >= a.n_users_blk) break;                       // wave-uniform
        const uint64_t* buf = my_lists + rr * kCap2;
        pda_wave_sync();
        const int c = min(__builtin_amdgcn_readfirstlane(cntl[wave * 32 + rr]), kCap2);
        const bool valid = lane < c;
        const int item = valid ? pda_key_item(buf[lane]) : a.item_offset;
        const int urow = __builtin_amdgcn_readlane(uid, rr);
        // user row: wave-uniform address -> scalar loads (constant address space; U is read-only for the kernel).
        // candidate row: the whole row (or 128 floats of it) in flight at once -- the gather latency is what this loop pays.
        typedef const __attribute__((address_space(4))) float* cfp;
        typedef const __attribute__((address_space(4))) uint16_t* chp;
        cfp up = (cfp)(a.U + (size_t)urow * D);
        chp up16 = (chp)(reinterpret_cast<const uint16_t*>(a.U) + (size_t)urow * D);
        auto uval = [&](int k) __attribute__((always_inline)) -> float {
            if constexpr (BF) return __uint_as_float((uint32_t)up16[k] << 16);
            else return up[k];
        };
        (void)up; (void)up16;
        const size_t ibase = (size_t)(item - a.item_offset) * D;
        float c0 = 0.f, c1 = 0.f;
        float popc = 1.0f;
        if constexpr (HEAD == PDA_HEAD_POP) popc = a.pop[item - a.item_offset];   // issued with the row gather, not after the chain
        constexpr int CH = D > 128 ? 128 : D;          // floats per gather step
#pragma unroll 1
        for (int base = 0; base < D; base += CH) {
            f32x4 iv[CH / 4];
#pragma unroll
            for (int q = 0; q < CH / 4; ++q) iv[q] = pda_load4<BF>(a.I, ibase + base + 4 * q);
#pragma unroll
            for (int c8 = 0; c8 < CH / 8; c8 += 2) {
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    c0 = __builtin_fmaf(uval(base + 8 * c8 + sidx), iv[2 * c8][sidx], c0);
                    c0 = __builtin_fmaf(uval(base + 8 * c8 + 4 + sidx), iv[2 * c8 + 1][sidx], c0);
                }
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    c1 = __builtin_fmaf(uval(base + 8 * c8 + 8 + sidx), iv[2 * c8 + 2][sidx], c1);
                    c1 = __builtin_fmaf(uval(base + 8 * c8 + 12 + sidx), iv[2 * c8 + 3][sidx], c1);
                }
            }
        }
        float sc = c0 + c1;
        if constexpr (HEAD == PDA_HEAD_POP) sc = (sc > 0.0f ? sc + 1.0f : __expf(sc)) * popc;
        const uint64_t key = valid ? pda_pack_key(sc, (uint32_t)item) : (uint64_t)(63 - lane);
        // rank = number of better keys.  The exact keys go back into the row's LDS slots and every lane reads all of them
        // with broadcast (uniform-address) LDS reads: independent loads, no v_readlane -> SGPR -> VALU chain per entry.
        uint64_t* xbuf = my_lists + rr * kCap2;
        if (lane < kCap2) xbuf[lane] = key;                 // lanes >= c hold distinct tiny pad keys: never above a real one
        pda_wave_sync();
        int rank = 0;
#pragma unroll
        for (int jj = 0; jj < kCap2; ++jj) rank += (xbuf[jj] > key) ? 1 : 0;
        uint64_t* out = a.out_keys + ((size_t)split * a.n_users_blk + rb) * K;
        if (valid && rank < K) out[rank] = key;
        if (lane >= c && lane < K) out[lane] = 0ull;
        // entries dropped by a band overflow: harmless unless they could still have reached the exact K-th value
        const float dmax = dropl[wave * 32 + rr];
        if (dmax != -INFINITY) {
            float kth = -INFINITY;
            if (c >= K) kth = pda_unordf((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), __builtin_ctzll(__ballot(valid && rank == K - 1))));
            if (dmax >= kth) ovf_tile = true;
        }
    }
    PDA_T1(tp, dbg_proc);
    if (ovf_tile && lane == 0) aa.tile_flags[utile] = 1;
#ifdef PDA_ABLATION
    if (lane == 0) {
        long long tot = 0;
        PDA_T1(t_all, tot);
        atomicAdd(&pda_dbg[0], (unsigned long long)dbg_entries);
        atomicAdd(&pda_dbg[1], (unsigned long long)dbg_proc);
        atomicAdd(&pda_dbg[2], (unsigned long long)dbg_comp);
        atomicAdd(&pda_dbg[3], (unsigned long long)dbg_push);
        atomicAdd(&pda_dbg[4], (unsigned long long)tot);
        atomicAdd(&pda_dbg[5], 1ull);
    }
#endif
}

template <int D, int HEAD, bool ORD = false, int ABL = 0, int NP = 3>
int launch_v2(const ScoreArgs2& aa, hipStream_t stream) {
    const size_t smem = (NP == 1 ? 1 : 2) * 32 * D * sizeof(uint16_t) + (size_t)kUserTile * (kCap2 * sizeof(uint64_t) + 16) + 16;
    static int attr_set = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&score_topk_v2_kernel<D, HEAD, ORD, ABL, NP>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return PDA_ERR_LAUNCH;
        attr_set = 1;
    }
    const int utiles = (aa.a.n_users_blk + kUserTile - 1) / kUserTile;
    hipLaunchKernelGGL((score_topk_v2_kernel<D, HEAD, ORD, ABL, NP>), dim3((unsigned)(utiles * aa.a.n_splits)), dim3(kThreads), smem, stream, aa);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

}  // namespace

#ifdef PDA_ABLATION
extern "C" int pda_debug_counters(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(pda_dbg), sizeof(unsigned long long) * 8) != hipSuccess) return PDA_ERR_LAUNCH;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(pda_dbg), z, sizeof(z)) != hipSuccess) return PDA_ERR_LAUNCH; }
    return PDA_OK;
}
#endif

static inline size_t prep_plane_bytes(int n, int d) { return (((size_t)n * d * 2) + 255) & ~(size_t)255; }
static inline size_t prep_norm_bytes(int n) { return (((size_t)n * 4) + 255) & ~(size_t)255; }

namespace {
// prep blob: [norm f32 n][max-norm bits, bad-order flag (256 B)] then, ordered only, [pop_p f32 n][order i32 n]
// [pos_of i32 n][sufA f32 n_tiles][sufB f32 n_tiles]; the bf16 planes come LAST (2 for fp32 tables, 0 / 1 for bf16
// tables unordered / ordered) so that the offsets of the small arrays do not depend on the table type.
struct PrepLayout {
    size_t plane, hi, lo, norm, nmax, pop_p, order, pos_of, sufA, sufB, bex, total;
};
PrepLayout prep_layout(int n, int d, bool ordered, int n_planes) {
    PrepLayout L{};
    L.plane = prep_plane_bytes(n, d);
    L.norm = 0;
    L.nmax = L.norm + prep_norm_bytes(n);      // int bits of the max norm at +0, "order is not a permutation" flag at +4
    size_t off = L.nmax + 256;
    if (ordered) {
        const size_t tiles = (((size_t)(n + 31) / 32) * 4 + 255) & ~(size_t)255;
        L.pop_p = off;
        L.order = L.pop_p + prep_norm_bytes(n);
        L.pos_of = L.order + prep_norm_bytes(n);
        L.sufA = L.pos_of + prep_norm_bytes(n);
        L.sufB = L.sufA + tiles;
        off = L.sufB + tiles;
    }
    L.bex = off;
    off += ((size_t)n * 32 + 255) & ~(size_t)255;
    L.hi = off;
    L.lo = off + L.plane;
    L.total = off + (size_t)n_planes * L.plane;
    return L;
}
inline int planes_of(bool bf16, bool ordered) { return bf16 ? (ordered ? 1 : 0) : 2; }

int run_item_prep(const void* I_shard, bool bf16, const float* pop, const int* order, int n, int d, void* prep, hipStream_t s) {
    const bool ordered = order != nullptr;
    const PrepLayout L = prep_layout(n, d, ordered, planes_of(bf16, ordered));
    char* pb = reinterpret_cast<char*>(prep);
    uint16_t* hi = planes_of(bf16, ordered) > 0 ? reinterpret_cast<uint16_t*>(pb + L.hi) : nullptr;
    uint16_t* lo = planes_of(bf16, ordered) > 1 ? reinterpret_cast<uint16_t*>(pb + L.lo) : nullptr;
    float* nrm = reinterpret_cast<float*>(pb + L.norm);
    int* nmax = reinterpret_cast<int*>(pb + L.nmax);
    float* pop_p = ordered && pop ? reinterpret_cast<float*>(pb + L.pop_p) : nullptr;
    int* pos_of = ordered ? reinterpret_cast<int*>(pb + L.pos_of) : nullptr;
    uint16_t* bex = reinterpret_cast<uint16_t*>(pb + L.bex);
    if (hipMemsetAsync(nmax, 0, 256, s) != hipSuccess) return PDA_ERR_LAUNCH;
    if (ordered) {
        if (hipMemsetAsync(pos_of, 0xFF, (size_t)n * 4, s) != hipSuccess) return PDA_ERR_LAUNCH;
        if (hipMemcpyAsync(pb + L.order, order, (size_t)n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return PDA_ERR_LAUNCH;
    }
#define PDA_PREP(DD)                                                                                         \
    case DD: {                                                                                               \
        constexpr int RPB = 256 / (DD / 8);                                                                  \
        const dim3 grid((unsigned)((n + RPB - 1) / RPB));                                                    \
        if (bf16) hipLaunchKernelGGL((item_prep_kernel<DD, true>), grid, dim3(256), 0, s, I_shard, n, hi, lo, nrm, nmax, order, pop, pop_p, pos_of, nmax + 1, bex); \
        else hipLaunchKernelGGL((item_prep_kernel<DD, false>), grid, dim3(256), 0, s, I_shard, n, hi, lo, nrm, nmax, order, pop, pop_p, pos_of, nmax + 1, bex); \
        break;                                                                                               \
    }
    switch (d) {
        PDA_PREP(64) PDA_PREP(128) PDA_PREP(256)
        default: return PDA_ERR_UNSUPPORTED;
    }
#undef PDA_PREP
    PDA_CHECK_LAUNCH();
    if (ordered) {
        const int n_tiles = (n + 31) / 32;
        float* tA = reinterpret_cast<float*>(pb + L.sufA);
        float* tB = reinterpret_cast<float*>(pb + L.sufB);
        hipLaunchKernelGGL(tile_bound_kernel, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, s, pop_p, nrm, n, n_tiles, tA, tB);
        PDA_CHECK_LAUNCH();
        hipLaunchKernelGGL(suffix_max_kernel, dim3(1), dim3(1024), 0, s, tA, tB, n_tiles);
        PDA_CHECK_LAUNCH();
    }
    return PDA_OK;
}

int run_score_prepped(const void* U, const void* I_shard, bool bf16, const void* prep, bool ordered, const float* pop_shard,
                      const int32_t* users, int n_users_blk, int item_offset, int n_items_local, int d,
                      const int64_t* hist_indptr, const int32_t* hist_indices, const int32_t* hist_indices_ord,
                      int hist_row_mode, int K, int head, int early_stop, int n_splits, uint64_t* out_keys, void* workspace,
                      hipStream_t s) {
    if (!U || !I_shard || !prep || !users || !out_keys || !workspace) return PDA_ERR_ARG;
    if (n_users_blk <= 0 || n_items_local <= 0 || item_offset < 0) return PDA_ERR_ARG;
    if (K < 1 || K > PDA_TOPK_CAP - 4) return PDA_ERR_ARG;
    if (head != PDA_HEAD_RAW && head != PDA_HEAD_POP) return PDA_ERR_ARG;
    if (head == PDA_HEAD_POP && !pop_shard) return PDA_ERR_ARG;
    if (hist_indptr && (!hist_indices || (ordered && !hist_indices_ord))) return PDA_ERR_ARG;
    if (n_splits <= 0) n_splits = pda_score_topk_auto_splits(n_users_blk, n_items_local);
    const PrepLayout L = prep_layout(n_items_local, d, ordered, planes_of(bf16, ordered));
    const char* pb = reinterpret_cast<const char*>(prep);
    int* ws = reinterpret_cast<int*>(workspace);
    if (hipMemsetAsync(workspace, 0, pda_score_topk_workspace_bytes(n_users_blk), s) != hipSuccess) return PDA_ERR_LAUNCH;
    if (head == PDA_HEAD_POP) {
        hipLaunchKernelGGL(pop_max_kernel, dim3(64), dim3(256), 0, s, pop_shard, n_items_local, ws);
        PDA_CHECK_LAUNCH();
    }
    // bf16 tables, natural order: the B tiles are read from the table itself
    const uint16_t* plane_hi = (bf16 && !ordered) ? reinterpret_cast<const uint16_t*>(I_shard) : reinterpret_cast<const uint16_t*>(pb + L.hi);
    ScoreArgs2 aa{{reinterpret_cast<const float*>(U), reinterpret_cast<const float*>(I_shard), pop_shard, users, hist_indptr,
                   ordered ? hist_indices_ord : hist_indices, out_keys, n_users_blk, item_offset, n_items_local, hist_row_mode, K,
                   n_splits, nullptr},
                  plane_hi, reinterpret_cast<const uint16_t*>(pb + L.lo),
                  reinterpret_cast<const float*>(pb + L.norm), reinterpret_cast<const float*>(pb + L.nmax),
                  reinterpret_cast<const float*>(ws), ws + 4,
                  ordered ? reinterpret_cast<const int*>(pb + L.order) : nullptr,
                  ordered ? reinterpret_cast<const float*>(pb + L.pop_p) : nullptr,
                  ordered && early_stop ? reinterpret_cast<const float*>(pb + L.sufA) : nullptr,   // NULL: visit everything
                  ordered && early_stop ? reinterpret_cast<const float*>(pb + L.sufB) : nullptr,
                  reinterpret_cast<unsigned long long*>(ws + 2),
                  reinterpret_cast<const uint16_t*>(pb + L.bex), hist_indices};
    // Which pre-filtered kernel: v3 (1 MFMA per k-step, candidate ring, exact lists, exact fp32-MFMA warm-up of the first
    // tiles of an ordered sweep) everywhere except the early-terminating sweep over bf16 tables at d = 256, where v3 has no
    // room for the warm-up block and v2 (approximate lists; one MFMA per k-step as well on bf16 tables) measures 15 %
    // faster.  PDA_SCORE_KERNEL=v2|v3 forces one (A/B measurements, cross-checks).
    bool use_v3 = !(bf16 && d == 256 && ordered && early_stop);
    // v3 packs (row, item id) into 32-bit ring words and uses 32-bit plane offsets
    const bool v3_fits = (uint64_t)item_offset + (uint64_t)n_items_local <= (1ull << 27) && (uint64_t)n_items_local * (uint64_t)d < (1ull << 32);
    use_v3 = use_v3 && v3_fits;
    if (const char* kk = getenv("PDA_SCORE_KERNEL")) {
        if (kk[0] == 'v' && kk[1] == '3') use_v3 = v3_fits;
        if (kk[0] == 'v' && kk[1] == '2') use_v3 = false;
    }
    if (use_v3) return pda_topk::launch_score_v3(aa, d, head, ordered, bf16, s);
    int rc = PDA_ERR_UNSUPPORTED;
#ifdef PDA_ABLATION
    if (const char* e = getenv("PDA_ABLATE")) {
        if (d == 128 && head == PDA_HEAD_POP && !ordered && !bf16) switch (atoi(e)) {
            case 1: return launch_v2<128, PDA_HEAD_POP, false, 1>(aa, s);
            case 3: return launch_v2<128, PDA_HEAD_POP, false, 3>(aa, s);
            case 7: return launch_v2<128, PDA_HEAD_POP, false, 7>(aa, s);
            case 4: return launch_v2<128, PDA_HEAD_POP, false, 4>(aa, s);
            case 15: return launch_v2<128, PDA_HEAD_POP, false, 15>(aa, s);
            case 31: return launch_v2<128, PDA_HEAD_POP, false, 31>(aa, s);
            case 47: return launch_v2<128, PDA_HEAD_POP, false, 47>(aa, s);
            case 63: return launch_v2<128, PDA_HEAD_POP, false, 63>(aa, s);
            case 39: return launch_v2<128, PDA_HEAD_POP, false, 39>(aa, s);
            case 23: return launch_v2<128, PDA_HEAD_POP, false, 23>(aa, s);
            default: break;
        }
    }
#endif
#define PDA_V2_(DD, ORDV, NPV) \
    (head == PDA_HEAD_POP ? launch_v2<DD, PDA_HEAD_POP, ORDV, 0, NPV>(aa, s) : launch_v2<DD, PDA_HEAD_RAW, ORDV, 0, NPV>(aa, s))
#define PDA_V2(DD)                                                                                   \
    case DD:                                                                                         \
        if (bf16) rc = ordered ? PDA_V2_(DD, true, 1) : PDA_V2_(DD, false, 1);                       \
        else rc = ordered ? PDA_V2_(DD, true, 3) : PDA_V2_(DD, false, 3);                            \
        break;
    switch (d) {
        PDA_V2(64) PDA_V2(128) PDA_V2(256)
        default: return PDA_ERR_UNSUPPORTED;
    }
#undef PDA_V2
#undef PDA_V2_
    if (rc != PDA_OK) return rc;
    // exact recomputation of the (normally zero) user tiles whose near-tie band overflowed: natural item order, the
    // caller's original history
    ScoreArgs v1 = aa.a;
    v1.hist_indices = hist_indices;
    v1.tile_flags = ws + 4;
    return pda_topk::launch_score_v1(v1, d, head, s, bf16);
}
}  // namespace

extern "C" size_t pda_item_prep_bytes(int n_items_local, int d) { return prep_layout(n_items_local, d, false, 2).total; }
extern "C" size_t pda_item_prep_ordered_bytes(int n_items_local, int d) { return prep_layout(n_items_local, d, true, 2).total; }
extern "C" size_t pda_item_prep_bf16_bytes(int n_items_local, int d) { return prep_layout(n_items_local, d, false, 0).total; }
extern "C" size_t pda_item_prep_ordered_bf16_bytes(int n_items_local, int d) { return prep_layout(n_items_local, d, true, 1).total; }

extern "C" size_t pda_score_topk_workspace_bytes(int n_users_blk) {
    // [max |pop| f32][pad][u64 item tiles scored, summed over workgroups][tile_flags i32 per 128-user tile]
    const size_t tiles = (size_t)(n_users_blk + kUserTile - 1) / kUserTile;
    return (16 + tiles * 4 + 255) & ~(size_t)255;
}

extern "C" int pda_item_prep_f32(const float* I_shard, int n_items_local, int d, void* prep, void* stream) {
    if (!I_shard || !prep || n_items_local <= 0) return PDA_ERR_ARG;
    return run_item_prep(I_shard, false, nullptr, nullptr, n_items_local, d, prep, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pda_item_prep_bf16(const uint16_t* I_shard, int n_items_local, int d, void* prep, void* stream) {
    if (!I_shard || !prep || n_items_local <= 0) return PDA_ERR_ARG;
    return run_item_prep(I_shard, true, nullptr, nullptr, n_items_local, d, prep, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pda_item_prep_ordered_f32(const float* I_shard, const float* pop_shard, const int32_t* order, int n_items_local,
                                         int d, void* prep, void* stream) {
    if (!I_shard || !order || !prep || n_items_local <= 0) return PDA_ERR_ARG;
    return run_item_prep(I_shard, false, pop_shard, order, n_items_local, d, prep, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pda_item_prep_ordered_bf16(const uint16_t* I_shard, const float* pop_shard, const int32_t* order, int n_items_local,
                                          int d, void* prep, void* stream) {
    if (!I_shard || !order || !prep || n_items_local <= 0) return PDA_ERR_ARG;
    return run_item_prep(I_shard, true, pop_shard, order, n_items_local, d, prep, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pda_item_prep_ordered_check(const void* prep, int n_items_local, int d, void* stream) {
    if (!prep || n_items_local <= 0) return PDA_ERR_ARG;
    const PrepLayout L = prep_layout(n_items_local, d, true, 0);
    int bad = 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(&bad, reinterpret_cast<const char*>(prep) + L.nmax + 4, 4, hipMemcpyDeviceToHost, s) != hipSuccess) return PDA_ERR_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return PDA_ERR_LAUNCH;
    return bad ? PDA_ERR_ARG : PDA_OK;
}

extern "C" int pda_hist_reorder(const void* prep, int n_items_local, int d, int item_offset, const int64_t* hist_indptr,
                                const int32_t* hist_indices, int n_rows, int32_t* out_indices, void* stream) {
    if (!prep || !hist_indptr || !hist_indices || !out_indices || n_items_local <= 0 || n_rows < 0 || item_offset < 0) return PDA_ERR_ARG;
    if (n_rows == 0) return PDA_OK;
    const PrepLayout L = prep_layout(n_items_local, d, true, 0);      // the small arrays sit in front of the planes
    hipLaunchKernelGGL(hist_reorder_kernel, dim3((unsigned)n_rows), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), hist_indptr,
                       hist_indices, reinterpret_cast<const int*>(reinterpret_cast<const char*>(prep) + L.pos_of), item_offset,
                       n_items_local, out_indices);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

#define PDA_SCORE_ARGS_DECL                                                                                                   \
    const float* pop_shard, const int32_t* users, int n_users_blk, int item_offset, int n_items_local, int d,                \
        const int64_t* hist_indptr, const int32_t* hist_indices
#define PDA_SCORE_TAIL_DECL int hist_row_mode, int K, int head, int n_splits, uint64_t* out_keys, void* workspace, void* stream
#define PDA_SCORE_TAIL_ORD_DECL \
    int hist_row_mode, int K, int head, int early_stop, int n_splits, uint64_t* out_keys, void* workspace, void* stream

extern "C" int pda_score_topk_prepped_f32(const float* U, const float* I_shard, const void* prep, PDA_SCORE_ARGS_DECL,
                                          PDA_SCORE_TAIL_DECL) {
    return run_score_prepped(U, I_shard, false, prep, false, pop_shard, users, n_users_blk, item_offset, n_items_local, d, hist_indptr,
                             hist_indices, nullptr, hist_row_mode, K, head, 0, n_splits, out_keys, workspace,
                             reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pda_score_topk_ordered_f32(const float* U, const float* I_shard, const void* prep, PDA_SCORE_ARGS_DECL,
                                          const int32_t* hist_indices_ord, PDA_SCORE_TAIL_ORD_DECL) {
    return run_score_prepped(U, I_shard, false, prep, true, pop_shard, users, n_users_blk, item_offset, n_items_local, d, hist_indptr,
                             hist_indices, hist_indices_ord, hist_row_mode, K, head, early_stop, n_splits, out_keys, workspace,
                             reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pda_score_topk_bf16(const uint16_t* U, const uint16_t* I_shard, const void* prep, PDA_SCORE_ARGS_DECL,
                                   PDA_SCORE_TAIL_DECL) {
    return run_score_prepped(U, I_shard, true, prep, false, pop_shard, users, n_users_blk, item_offset, n_items_local, d, hist_indptr,
                             hist_indices, nullptr, hist_row_mode, K, head, 0, n_splits, out_keys, workspace,
                             reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pda_score_topk_ordered_bf16(const uint16_t* U, const uint16_t* I_shard, const void* prep, PDA_SCORE_ARGS_DECL,
                                           const int32_t* hist_indices_ord, PDA_SCORE_TAIL_ORD_DECL) {
    return run_score_prepped(U, I_shard, true, prep, true, pop_shard, users, n_users_blk, item_offset, n_items_local, d, hist_indptr,
                             hist_indices, hist_indices_ord, hist_row_mode, K, head, early_stop, n_splits, out_keys, workspace,
                             reinterpret_cast<hipStream_t>(stream));
}
