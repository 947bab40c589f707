// The "huge" geometry of the sweep (round 4): sweep5_kernel -- included by pda_score_topk_v4.hip behind the shared device pieces.
//
// What generation 4's wide geometry is bound by: the chip is power-limited on the block loop, and what the MFMAs are fed
// from decides the clock -- one ds_read_b128 per two MFMAs 1 500 TFLOP/s executed, register operands 2 050.  And 1/9 of the MFMAs it
// executes (1/5 at d = 64) are the folded threshold test.  This geometry removes both (tools/ubench/mfma_struct5.hip measured the
// mapping first: 1 777 TFLOP/s, all of it algorithmic, against 1 364 algorithmic for the wide mapping on the same box):
//   * FOUR waves per workgroup, one per SIMD, 512 registers each; a wave owns 256 users, whose bf16 rows sit in AGPRs as eight MFMA
//     operands of 32 users (re-loaded from the user image at every entry of the asm loop: the
//     compiler spills into AGPRs between the statements) -- every item fragment read from the LDS feeds EIGHT MFMAs, and a 32-item half-tile is DMA-ed into the LDS
//     once per 1 024 users;
//   * the product is TRANSPOSED (A = item fragment, B = user fragment): an accumulator lane holds 16 items of ONE user, so the test
//     "could this pair reach the user's threshold" is a per-lane compare of the lane's maximum (VALU in the MFMA shadow) -- no test k-step;
//   * the item image is PRE-SCALED by the popularity: i' = pop * i, so that for s > 0 the head (s + 1) pop = u . i' + pop, and the
//     per-item terms (pop, the bf16 error bound eps ||u|| ||i'||) are replaced by their maxima over the 32-item half-tile (pmax, nmax:
//     8 bytes of meta per half-tile; in visiting order the popularities of a half-tile are all but equal):
//         flag  <=>  max(max_i s~'(u, i) + pmax + eps_w nmax, pmax) > thr'(u)         (eps_w = eps x the wave's largest ||u||)
//     which is implied by "some pair of the half-tile has an exact head >= the user's K-th value" (s <= 0: the head is <= pop <= pmax);
//   * no loader and no rescoring waves: the MFMA waves issue the LDS-DMA themselves (two 1 KiB pieces per wave and half-tile at
//     d = 128) and run in step, one s_barrier per half-tile; when a half-tile raised a flag in any of them, all four leave their asm
//     loop together, score that half-tile (and the one behind it) again with compiler-visible MFMAs, rescore the candidates exactly
//     (the fp32 chain of the oracle, generation 4's lists and compaction) and re-enter.  In a dense sweep in visiting order that is a
//     fraction of a candidate per user behind the exact warm-up.
// Everything else is generation 4's: warm4_kernel's exact lists (handed over through the workspace), the packed keys, the epilogue.
// Popularity head, d = 64 / 128, dense sweeps (no early termination); selected by the caller's hint PDA_SWEEP_HUGE.
#pragma once
#ifdef PDA_V6_LOOP_HEADER
#include PDA_V6_LOOP_HEADER
#else
#include "pda_v6_loop_asm.h"           // the loop, on v_mfma_f32_16x16x32_bf16 (tools/gen_v6_loop_asm.py)
#endif

#ifdef PDA_V5_LOG
__device__ unsigned pda_v5_log[1 << 18];      // debug build: [0] = entries used; then (block << 8 | wave, kind, a, b) per event
#define V5LOG(kind, a, b) do { if (lane == 0) { const unsigned i_ = atomicAdd(&pda_v5_log[0], 1u); if (i_ < (1u << 16) - 1u) { \
    pda_v5_log[4 * i_ + 4] = ((unsigned)blockIdx.x << 8) | (unsigned)wave; pda_v5_log[4 * i_ + 5] = (kind); pda_v5_log[4 * i_ + 6] = (a); pda_v5_log[4 * i_ + 7] = (b); } } } while (0)
#else
#define V5LOG(kind, a, b) do {} while (0)
#endif
constexpr int kRing5 = 192;           // candidate ring entries per wave (u64 each); a push needs 64 free



// UPW users per wave: 256 -- one 1 024-user workgroup per CU, 512 registers per wave; d = 256: 128 (the rows of 128 users fill the 256 AGPRs).
// With one wave per SIMD nothing overlaps the wave's own VALU tests, LDS reads and scalar work with its MFMAs (PMC: matrix pipe 77 % busy at
// 2.03 GHz, 87 % without the tests).  Round 4 measured the two obvious alternatives and round 5 removed them (profiles/round4_huge_variants.txt,
// profiles/README.md): two 512-user workgroups per CU at d <= 128 -- the pipe 82 % busy, but at 1.86 GHz: twice the LDS reads and LDS-DMA per
// MFMA cost more clock than the overlap buys (8.38 vs 8.18 ms) -- and the loop on v_mfma_f32_32x32x16_bf16 (9.16 ms).
template <int D, bool BF, bool S16, int UPW>
__global__ void __launch_bounds__(256, 1) sweep5_kernel(Args4 g) {
    [[maybe_unused]] constexpr int HB = half_bytes5(D);
    static_assert(slot_bytes5(D) == Loop6<D, 8>::kSlotBytes, "one LDS image for all loops");
    static_assert(S16, "the loop on v_mfma_f32_32x32x16_bf16 (round 4's first form: 9.16 against 8.07 ms, profiles/README.md) left with round 5; the parameter keeps the kernel's name");
    static_assert((D <= 128 && UPW == 256) || (D == 256 && UPW == 128), "users per wave: 256, or (d = 256) 128 -- 8 blocks x 8 k-steps = 256 AGPRs, one 512-user workgroup per CU");
    // NK k-steps per product, NU user blocks of UBW = 16 users per wave (16 x 16 x 32 MFMAs)
    constexpr int NK = D / 32, UBW = 16, NU = UPW / UBW;
    constexpr int SS = slot_bytes5(D), UT = 4 * UPW, CAPL = kCap4, RB4 = row_bytes(D);
    constexpr int LPC = D / 32, CPP = 64 / LPC;                          // lanes per candidate, candidates per rescoring pass
    constexpr float kEps5 = BF ? 4.0234375e-3f : 8.046875e-3f;           // 2^-8 x 1.03 (only the scaled items are rounded: one unit roundoff of bf16)  |  2^-7 x 1.03 (both sides)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* tiles = smem;                                         // kNSlot5 x SS (at LDS address 0: the loop XORs fragment offsets into slot addresses)
    int* cntl = reinterpret_cast<int*>(smem + kNSlot5 * SS);            // [UT]
    float* taul = reinterpret_cast<float*>(cntl + UT);                   // [UT]
    uint64_t* crings = reinterpret_cast<uint64_t*>(taul + UT);           // [4][kRing5]  (user row of the wave << 32 | visiting position)
    unsigned* sync = reinterpret_cast<unsigned*>(crings + 4 * kRing5);   // [2] the shared flag words, one per half-tile parity (pda_v5_loop_asm.h)
    unsigned* s_uns = sync + 8;                                          // [32] one bit per user row: its list came in unsorted
    unsigned* touched = sync + 40;                                       // [32] one bit per user row: the sweep appended to its list
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x % g.n_splits, utile = blockIdx.x / g.n_splits;
    // the exact lists: the hand-over rows of the warm-up IN PLACE (a one-call sweep; the rows of the workgroup's padding do not exist
    // and are never touched: no candidate, no output), else the workgroup's rows of the workspace
    uint64_t* lists = g.handover != nullptr ? g.handover + ((size_t)split * g.n_users_blk + (size_t)utile * UT) * kCap4
                                            : g.lists_ws + (size_t)blockIdx.x * UT * CAPL;
    const int K = g.K;
    const int nt = split_tiles(g.n_tiles, split, g.n_splits);
    const int wt = warm_tiles_of(g, split);                              // the split's tiles the warm-up has scored
    const bool starts_empty = g.lists_empty || (g.warm_shared && split > 0);   // shared warm-up: its lists went to split 0; this split has the seed and empty lists
    const bool warm_final = g.warm_final && !starts_empty;               // (an empty split writes every row of its out_keys itself)
    const int n_it = max(0, nt - wt);                                    // 64-item tiles behind the warm-up
    const unsigned hend = 2u * (unsigned)n_it;                           // 32-item half-tiles
    // (words 0, 1: the shared flag words; 8 .. 39: every list comes in unsorted -- unless the warm-up sorted them: warm_final; 40 .. 71: no row touched)
    if (tid < 72) sync[tid] = (tid < 8 || tid >= 40 || warm_final || starts_empty) ? 0u : 0xFFFFFFFFu;
    // kernel identity (workspace + 16): generation 4 | geometry (4: the 16 x 16 x 32 loop, one 1 024-user workgroup per CU; 6: the same, two 512-user workgroups; 5: the 32 x 32 x 16 loop) << 8 | head << 13 | bf16 tables << 14 | d / 64
    if (tid == 0 && blockIdx.x == 0) g.stats[4] = (4u << 28) | (4u << 8) | (1u << 13) | ((BF ? 1u : 0u) << 14) | (unsigned)(D >> 6);
    // ---- the counts and K-th values of the warm-up's lists -> LDS (all waves); without a hand-over buffer the lists themselves -> the workspace
    if (starts_empty) {
        for (int rr = tid; rr < UT; rr += 256) {
            cntl[rr] = 0;
            taul[rr] = utile * UT + rr < g.n_users_blk ? -INFINITY : INFINITY;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    } else if (warm_final) {
        // sorted lists of at most K keys (warm4_kernel): a LANE per row -- its K-th key says everything (0: fewer than K keys; count them)
        for (int rr = tid; rr < UT; rr += 256) {
            const int rb = utile * UT + rr;
            int c = 0;
            float tau = INFINITY;
            if (rb < g.n_users_blk) {
                const uint64_t kl = lists[(size_t)rr * CAPL + (K - 1)];
                c = K;
                tau = pda_unordf((uint32_t)(kl >> 32));
                if (kl == 0ull) {
                    tau = -INFINITY;
                    for (c = 0; c < K - 1 && lists[(size_t)rr * CAPL + c] != 0ull; ++c) {}
                }
            }
            cntl[rr] = c;
            taul[rr] = tau;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    } else
    {
        constexpr int NW = 4, PB = 8;
        for (int r0 = wave; r0 < UT; r0 += PB * NW) {
            uint64_t keyv[PB];
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                const int rr = r0 + q * NW, rb = utile * UT + rr;
                keyv[q] = (rr >= UT || rb >= g.n_users_blk) ? 0ull
                          : g.handover != nullptr ? (lane < kCap4 ? g.handover[((size_t)split * g.n_users_blk + rb) * kCap4 + lane] : 0ull)
                          : (lane < K ? g.out_keys[((size_t)split * g.n_users_blk + rb) * K + lane] : 0ull);
            }
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                const int rr = r0 + q * NW;
                if (rr >= UT) break;
                const int rb = utile * UT + rr;
                const uint64_t key = keyv[q];
                const int c = __popcll(__ballot(key != 0ull));
                if (g.handover == nullptr && lane < K) lists[(size_t)rr * CAPL + lane] = key;
                uint32_t mn = key != 0ull ? (uint32_t)(key >> 32) : 0xFFFFFFFFu;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
                if (lane == 0) {
                    cntl[rr] = c;
                    taul[rr] = rb < g.n_users_blk ? (c >= K ? pda_unordf(mn) : -INFINITY) : INFINITY;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    __syncthreads();

    unsigned n_cand = 0;
    if (hend > 0) {
        const int row0 = wave * UPW;                                     // this wave's user rows of the workgroup
        const int j = lane & (UBW - 1), hh = lane / UBW;               // the lane's user of a block; its 8-element group of a fragment's k-range
        const unsigned lane16 = (unsigned)lane * 16u;
        const unsigned char* my_ufrag = g.ufrag + ((size_t)utile * 4 + wave) * (size_t)(NU * NK * 1024);
        // the wave's largest padded ||u||, the external seeds and the history bounds of the lane's users (block u: row UBW u + j)
        float eu = 0.f, seedv[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int rb = utile * UT + row0 + UBW * u + j;
            eu = fmaxf(eu, rb < g.n_users_blk ? g.unorm[rb] : 0.f);
            seedv[u] = (g.seed != nullptr && rb < g.n_users_blk) ? g.seed[rb] : -INFINITY;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) eu = fmaxf(eu, __shfl_xor(eu, o, 64));
        eu = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(eu * (kEps5 * 1.001f))));       // eps x the wave's largest ||u||, wave-uniform
        const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)tiles;
        const unsigned flags_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(sync);
        uint64_t* my_ring = crings + wave * kRing5;
        unsigned ring_n = 0;                                             // wave-uniform: entries in my_ring
        const unsigned t0 = (unsigned)(split + wt * g.n_splits);
        const size_t img = (size_t)g.rows5, meta = (size_t)g.meta5;
        const bool hist_on = g.hist_indptr != nullptr;

        // the lowered threshold of the lane's user of block u: strictly below the exact K-th value (ties must pass) and below the fp32
        // roundings between the bound and the rescored head; +-1e30 stand for +-inf
        auto thr_of = [&](int u) __attribute__((always_inline)) -> float {
            const float tq = fmaxf(taul[row0 + UBW * u + j], seedv[u]);
            float tf = (tq == INFINITY || tq == -INFINITY) ? tq : tq - fabsf(tq) * 1.52587890625e-5f - 1e-30f;
            return fminf(fmaxf(tf, -1.0e30f), 1.0e30f);
        };
        // ---- exact rescoring of the ring's candidates: the fp32 chain of the oracle (oracle/pda_oracle.c dot_chain; sweep4_kernel's
        // interleaved row loads), the train-item check, the append.  CPP candidates per pass, LPC lanes each.
        auto rescore_ring = [&]() __attribute__((always_inline)) {
            const int q = lane % LPC, ci = lane / LPC;
            for (unsigned base = 0; base < ring_n; base += CPP) {
                const bool have = base + (unsigned)ci < ring_n;
                const uint64_t e = have ? my_ring[base + ci] : 0ull;
                const int row = (int)(e >> 32);
                const unsigned pos = (unsigned)e;
                bool valid = have && pos < (unsigned)g.n_items_local;
                const unsigned char* tail = g.rows + (size_t)(valid ? pos : 0u) * RB4 + 2 * D + 32;
                const float pv = *reinterpret_cast<const float*>(tail);
                const int loc = *reinterpret_cast<const int*>(tail + 4);
                const int rb = utile * UT + row0 + row;
                valid = valid && rb < g.n_users_blk;
                const int uid = valid ? g.users[rb] : 0;
                const float c_sd = (g.seed != nullptr && valid) ? g.seed[rb] : -INFINITY;
                const float c_tau = taul[row0 + (valid ? row : 0)];
                f32x4 uu[8], ii[8];
                // d <= 128: lane q holds chunks LPC cq + q (interleaved: a cache line per candidate and load); d = 256: its 32 consecutive elements
                const size_t ub = (size_t)uid * D + q * (LPC == 8 ? 32 : 8), ib = (size_t)(valid ? loc : 0) * D + q * (LPC == 8 ? 32 : 8);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int off = LPC == 8 ? 4 * c : 8 * LPC * (c >> 1) + 4 * (c & 1);
                    uu[c] = pda_load4<BF>(g.U, ub + off);
                    ii[c] = pda_load4<BF>(g.I, ib + off);
                }
                long long hb = 0, he = 0;
                if (hist_on && valid) {
                    const int64_t hr = g.hist_row_mode == PDA_HIST_BY_USER_ID ? (int64_t)uid : (int64_t)rb;
                    hb = g.hist_indptr[hr];
                    he = g.hist_indptr[hr + 1];
                }
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
                    // d = 256: chunk x = 4 q + cc of the row feeds chain x & 1, the chunks of a chain in ascending order: the two chains
                    // travel from lane to lane (sweep4_kernel's rows that are not interleaved)
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
                float sc = o0 + o;                                  // meaningful on the candidate's last lane
                sc = (sc > 0.0f ? sc + 1.0f : __expf(sc)) * pv;
                const float tt = (valid && q == LPC - 1) ? sc : -INFINITY;
                const int item = g.item_offset + loc;
                bool p = valid && q == LPC - 1 && (tt >= fmaxf(c_tau, c_sd));
                if (hist_on && p) {                                 // train items are masked here: a binary search in the row's id-sorted history
                    long long lo = hb, hi2 = he;
                    while (lo < hi2) {
                        const long long mid = (lo + hi2) >> 1;
                        if (g.hist_indices[mid] < item) lo = mid + 1; else hi2 = mid;
                    }
                    if (lo < he && g.hist_indices[lo] == item) p = false;
                }
                const uint64_t key = pda_pack_key(tt, (uint32_t)item);
                if (p) atomicOr(&touched[(row0 + row) >> 5], 1u << ((row0 + row) & 31));
                append_keys<CAPL, true>(p, row0 + row, tt, key, lists, cntl, taul, row0, UPW, K, lane, s_uns);
            }
            n_cand += ring_n;
            ring_n = 0;
        };
        // ---- a half-tile that raised a flag, scored again with compiler-visible MFMAs: every pair whose bound reaches the user's
        // threshold -> the ring
        auto extract = [&](unsigned ft) __attribute__((always_inline)) {
            const unsigned T = t0 + (ft >> 1) * (unsigned)g.n_splits;                   // its 64-item tile
            const unsigned pos0 = T * 64u + (ft & 1u) * 32u;                            // visiting position of its first item
            const float2 mt = *reinterpret_cast<const float2*>(g.meta5 + 4 * (size_t)(2u * T + (ft & 1u)));
            const float ct = __builtin_fmaf(eu, mt.y, mt.x);
            const unsigned char* tb = tiles + (ft & (kNSlot5 - 1)) * SS + j * (2 * D);
            // accumulator register r of the lane <-> item of the half-tile (r = 4 ib + register of chain ib)
            auto item_of = [&](int r) __attribute__((always_inline)) -> unsigned {
                return 16u * (r >> 2) + 4u * hh + (r & 3);
            };
            constexpr int NR = 8;
            u32x4 af[2 * NK];
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int k = 0; k < NK; ++k) af[ib * NK + k] = *reinterpret_cast<const u32x4*>(tb + ib * 16 * (2 * D) + (((4 * k + hh) ^ swz5<D>(j)) << 4));
            for (int u = 0; u < NU; ++u) {
                const float tl = thr_of(u);
                const bool clampy = mt.x > tl;
                uint32_t m = 0;
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < NK; ++k) {
                        const u32x4 bf = *reinterpret_cast<const u32x4*>(my_ufrag + (((size_t)u * NK + k) * 64 + lane) * 16);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[ib * NK + k]), __builtin_bit_cast(bf16x8, bf), acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) m |= (acc[r] + ct > tl) ? (1u << (4 * ib + r)) : 0u;
                }
                if (__any(clampy)) {
                    // (rare: a user whose threshold lies below a popularity of this half-tile -- a head pop x exp(s), s <= 0, may qualify)
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const unsigned pp = pos0 + item_of(r);
                        const float pi = clampy ? *reinterpret_cast<const float*>(g.rows + (size_t)pp * RB4 + 2 * D + 32) : 0.f;
                        m |= (clampy && pi > tl) ? (1u << r) : 0u;
                    }
                }
                while (__any(m != 0u)) {
                    if (ring_n + 64u > (unsigned)kRing5) rescore_ring();
                    const bool act = m != 0u;
                    const int r = __builtin_ctz(m | 0x10000u);
                    m &= ~(1u << r);
                    const uint64_t pm = __ballot(act);
                    const unsigned slot = ring_n + (unsigned)__builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0));
                    if (act) my_ring[slot] = ((uint64_t)(unsigned)(UBW * u + j) << 32) | (uint64_t)(pos0 + item_of(r));
                    ring_n += (unsigned)__popcll(pm);
                }
            }
        };

        unsigned h = 0, issued = 0, n_entries = 0;
        bool hend_ok = true;
        if ((ring_lds & (D == 256 ? 511u : 255u)) != 0u) { if (lane == 0) g.stats[0] = 6u; hend_ok = false; }      // (the slots must start at multiples of 256 / 512)
        for (unsigned guard = 0; hend_ok && guard < 2u * hend + 8u; ++guard) {
            ++n_entries;
            float thr[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) thr[u] = thr_of(u);
            unsigned reason = 0;
            // the wave's lowest threshold: the clamp test (a popularity of the half-tile above a user's threshold) is wave-uniform
            float tmin = thr[0];
#pragma unroll
            for (int u = 1; u < NU; ++u) tmin = fminf(tmin, thr[u]);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) tmin = fminf(tmin, __shfl_xor(tmin, o, 64));
            tmin = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(tmin)));
            Loop6<D, NU>::run(h, issued, reason, hend, ring_lds, flags_lds, 1024u * (unsigned)wave, t0, (unsigned)g.n_splits,
                              (unsigned)img, (unsigned)(img >> 32), (unsigned)meta, (unsigned)(meta >> 32), eu, tmin, my_ufrag, thr, lane16);
            V5LOG(reason, h, issued);
            if (reason == 0u) break;
            if (reason != 1u) { if (lane == 0) g.stats[0] = 5u; break; }
#ifdef PDA_V5_LOG
            // LDS integrity: the half-tiles about to be scored again and the one in progress, in their slots, against the image
            for (unsigned tq = (h >= 2u ? h - 2u : 0u); tq <= h && tq < hend; ++tq) {
                const unsigned Tq = t0 + (tq >> 1) * (unsigned)g.n_splits;
                const unsigned char* gsrc = g.rows5 + ((size_t)(2u * Tq + (tq & 1u))) * HB;
                const unsigned char* lsrc = tiles + (tq & (kNSlot5 - 1)) * SS;
                unsigned pieces = 0;
                for (int i = 0; i < HB / 1024; ++i) {
                    const u32x4 a4 = *reinterpret_cast<const u32x4*>(gsrc + i * 1024 + lane * 16);
                    const u32x4 b4 = *reinterpret_cast<const u32x4*>(lsrc + i * 1024 + lane * 16);
                    if (__ballot(((a4[0] ^ b4[0]) | (a4[1] ^ b4[1]) | (a4[2] ^ b4[2]) | (a4[3] ^ b4[3])) != 0u) != 0ull) pieces |= 1u << i;
                }
                if (pieces != 0u) V5LOG(13u, (tq << 8) | (h - tq), pieces);
            }
#endif
            // every wave of the workgroup left at half-tile h because SOME wave's lanes flagged h - 2; the flags of h - 1 were still being
            // worked out: both are scored again here (a half-tile without a candidate of this wave costs its 8 NK MFMAs)
            [[maybe_unused]] const unsigned before = ring_n + n_cand;
            if (h >= 2u && h - 2u < hend) extract(h - 2u);
            if (h >= 1u && h - 1u < hend) extract(h - 1u);
            V5LOG(10u, h, ring_n + n_cand - before);
            // (thresholds rise only through the lists: rescoring a ring that holds a pass's worth keeps them fresh enough)
            if (ring_n >= (unsigned)CPP) rescore_ring();
        }
        if (ring_n > 0u) rescore_ring();
        if (lane == 0) atomicAdd(g.stats + 1, n_cand);
        if (lane == 0) atomicAdd(g.stats + 5, n_entries);                  // (workspace + 20: entries of the asm loop, summed over the waves)
        if (lane == 0 && wave == 0) atomicAdd(reinterpret_cast<unsigned long long*>(g.stats + 2), (unsigned long long)(2 * n_it * (UT / kUserTile)));
    }
    // ================================== all waves: sort and emit ==================================
    // Every list comes out of the warm-up unsorted with K .. kCap4 keys, and all but a few per cent of them are untouched since: the
    // final top-K selection and sort of all 1 024 rows is a fixed cost of the launch.  With one wave per SIMD nothing hides a round
    // trip, so: the keys of eight rows are requested together, and a row is ranked out of the LDS (the row's keys written once, then
    // read back as broadcasts, two keys per ds_read_b128) -- ~150 instructions per row where compact_list's loop over the list in
    // global memory was a dependent round trip per four keys (0.87 ms of a 9.0 ms launch -> 0.2).  A wave emits its own 256 rows.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __syncthreads();
    {
        constexpr int EB = 8;
        uint64_t* scr = crings + wave * kRing5;                          // (the rings are empty now)
        auto emit_row = [&](int rb, int cnt, uint64_t kraw) __attribute__((always_inline)) {
            const int c = min(__builtin_amdgcn_readfirstlane(cnt), CAPL);                    // (failed appends may have pushed it past the capacity)
            const uint64_t key = lane < c ? kraw : (uint64_t)(63 - lane);                    // fillers: unique, below any real key
            scr[lane] = key;
            pda_wave_sync();
            int rank = 0;
            for (int jj = 0; jj < c; jj += 4) {
                const uint64_t k0 = scr[jj], k1 = scr[jj + 1], k2 = scr[jj + 2], k3 = scr[jj + 3];
                rank += ((k0 > key) ? 1 : 0) + ((k1 > key) ? 1 : 0) + ((k2 > key) ? 1 : 0) + ((k3 > key) ? 1 : 0);
            }
            pda_wave_sync();
            uint64_t* orow = g.out_keys + ((size_t)split * g.n_users_blk + rb) * K;
            if (lane < c && rank < K) orow[rank] = key;
            if (lane >= c && lane < K) orow[lane] = 0ull;
        };
        if (warm_final) {
            // out_keys holds the warm-up's sorted rows: only the rows the sweep appended to are ranked and written again
            for (int w = 0; w < UPW / 32; ++w) {
                unsigned m = touched[wave * (UPW / 32) + w];
                while (m != 0u) {
                    const int b = __builtin_ctz(m);
                    m &= m - 1u;
                    const int rr = wave * UPW + 32 * w + b;
                    emit_row(utile * UT + rr, cntl[rr], lists[(size_t)rr * CAPL + (lane < CAPL ? lane : CAPL - 1)]);
                }
            }
        } else if (starts_empty) {
            // an empty split behind a shared warm-up: every row is written here -- zeros, then the few rows the sweep appended to
            const int rb0 = utile * UT + wave * UPW, nv = max(0, min(UPW, g.n_users_blk - rb0));
            uint64_t* ob = g.out_keys + ((size_t)split * g.n_users_blk + rb0) * K;
            for (int i = lane; i < nv * K; i += 64) ob[i] = 0ull;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // (the same wave writes the touched rows again below)
            for (int w = 0; w < UPW / 32; ++w) {
                unsigned m = touched[wave * (UPW / 32) + w];
                while (m != 0u) {
                    const int b = __builtin_ctz(m);
                    m &= m - 1u;
                    const int rr = wave * UPW + 32 * w + b;
                    emit_row(utile * UT + rr, cntl[rr], lists[(size_t)rr * CAPL + (lane < CAPL ? lane : CAPL - 1)]);
                }
            }
        } else {
            for (int i0 = 0; i0 < UPW; i0 += EB) {
                uint64_t kraw[EB];
                int cv[EB];
#pragma unroll
                for (int q = 0; q < EB; ++q) {
                    const int rr = wave * UPW + i0 + q, rb = utile * UT + rr;
                    const bool ok = rb < g.n_users_blk;
                    kraw[q] = ok ? lists[(size_t)rr * CAPL + (lane < CAPL ? lane : CAPL - 1)] : 0ull;
                    cv[q] = ok ? cntl[rr] : 0;
                }
#pragma unroll
                for (int q = 0; q < EB; ++q) {
                    const int rb = utile * UT + wave * UPW + i0 + q;
                    if (rb >= g.n_users_blk) break;
                    emit_row(rb, cv[q], kraw[q]);
                }
            }
        }
    }
}

template <int D, bool BF, bool S16, int UPW>
int launch_sweep5(const Args4& g, hipStream_t stream) {
    constexpr int UT = 4 * UPW;
    constexpr size_t lds = (size_t)kNSlot5 * slot_bytes5(D) + (size_t)UT * 8 + 4 * kRing5 * 8 + 80 * 4 + 64;
    static_assert(UPW == 256 || D == 256 || 2 * lds <= 160 * 1024, "two workgroups per CU");
    static_assert(lds <= 160 * 1024, "LDS per workgroup");
    static int attr_set = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep5_kernel<D, BF, S16, UPW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PDA_ERR_LAUNCH;
        attr_set = 1;
    }
    const int utiles = (g.n_users_blk + UT - 1) / UT, n_pad = utiles * UT;
    hipLaunchKernelGGL((uprep5_kernel<D, BF, S16, UPW>), dim3((unsigned)(((size_t)n_pad * (D / 8) + 255) / 256)), dim3(256), 0, stream, g.U, g.users, g.n_users_blk, n_pad,
                       const_cast<unsigned char*>(g.ufrag), const_cast<float*>(g.unorm));
    PDA_CHECK_LAUNCH();
    hipLaunchKernelGGL((sweep5_kernel<D, BF, S16, UPW>), dim3((unsigned)(utiles * g.n_splits)), dim3(256), lds, stream, g);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}
