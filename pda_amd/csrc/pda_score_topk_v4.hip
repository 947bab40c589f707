// score + mask + top-K, generation 4: the v3 algorithm (ONE bf16 MFMA per k-step as a rigorous pre-filter, threshold test
// folded into one extra k-step, exact fp32 rescoring of the few survivors, exact per-user lists) on a different machine
// mapping.  Same packed keys as v1 / v3, bit for bit (tests/test_gpu_score_topk.py runs every case through all of them).
//
// What v3's profile said (profiles/round1k_pmc.txt): matrix pipe 39 % busy, ~11 non-MFMA instructions per MFMA, 35 % of
// the wave cycles waiting -- every wave amortised its B reads, barriers and loop control over 32 user rows only, and every
// ring drain (a chain of dependent gathers) stopped the MFMAs of its wave.  v4 therefore
//   * splits the workgroup into MFMA waves, LOADER waves and RESCORING waves (a rescoring wave owns the lists of its rows,
//     drains their candidate rings, gathers the exact rows, runs the fp32 fmaf chains, appends, compacts, publishes the
//     thresholds).  The MFMA waves never wait for a gather: the latency-bound half of the algorithm runs beside them on the
//     same SIMDs.  Results do not depend on the timing: a stale threshold is a LOWER threshold (more candidates, never
//     fewer), every candidate is rescored exactly, and the lists keep the exact best K.  Two geometries (Geo4<D, GM>, chosen by
//     the caller's hint, identical keys): 8 MFMA x 32 rows + 4 loaders + 4 rescoring waves with the lists in the LDS (default,
//     d <= 128; d = 256: 8 + 3 + 1, the lists in the workspace and four tile slots); MANY: 4 MFMA x 32 rows + 4 loaders + 8
//     rescoring waves, 128 users per workgroup -- the raw head and natural order, hundreds of list insertions per user, are
//     bound by rescoring.  (Rounds 3 and 4 also shipped a FEW_CANDIDATES geometry -- the default with its lists in the workspace --
//     and a WIDE one -- 8 MFMA x 64 rows, 512 users per workgroup; the huge geometry, pda_v5_sweep.h, took over every block
//     they served and round 5 removed them: profiles/README.md keeps their measurements.)
//   * has no s_barrier in the loop (it would tie the rescoring waves to the tile cadence): the MFMA waves hand tiles to each
//     other through one monotonic LDS counter -- "my share of tile i+1 has landed and I am done reading tile i" -- which
//     orders both the RAW and the WAR side of the two tile buffers;
//   * streams the tiles with global_load_lds (LDS-DMA, no staging registers, no ds_write pass) from a layout the prep
//     kernel pads per item: [d bf16][16 bf16 test pieces][pop f32][local id i32][norm f32][0] = 2d + 48 bytes.  The row
//     stride is an ODD number of 16-byte slots, so the B-operand ds_read_b128 of any 16 rows hits 16 different bank slots
//     WITHOUT a swizzle: the LDS image is the global image, a tile is one contiguous run of whole 1 KiB DMA pieces, every
//     LDS address in the loop is a per-lane base plus an immediate, and the test pieces, the popularity and the item id
//     travel with the tile (no per-tile scalar-indexed side loads).  Tail tiles are padded with null items that can never
//     become candidates, so the loop carries no "item exists" masks;
//   * takes the exact warm-up out of the sweep: warm4_kernel scores the first tiles of every split exactly on the fp32
//     matrix cores (the v1 k order = the fmaf chain of the oracle) and leaves sorted lists in out_keys; sweep4_kernel
//     starts from them.  (Inside one kernel the warm-up block dictated the register allocation of the main loop.)
//   * treats natural item order as the visiting order "identity": ONE kernel for all three sweep modes; train items are
//     masked at the candidate stage for all of them (binary search in the caller's id-sorted history by the rescoring wave).
//
// Error bound of the filter, folded test, tie handling: pda_score_topk_v3.hip (unchanged).
#include "pda_v4_shared.h"
#include <cstdlib>
#include <type_traits>

using namespace pda_topk;

namespace {

constexpr int kCap4 = 57;            // list slots per user (LDS budget at d = 128: 256 users x 57 x 8 B = 114 KiB)
constexpr int kRing4 = 128;          // ring entries per MFMA wave (u32 each); a push needs 64 free
constexpr int kWarmTiles = 4;        // 64-item tiles per split scored by warm4_kernel (256 items)
constexpr int kMainWaves = 8;
#ifndef PDA_W4_ABL
#define PDA_W4_ABL 0      // warm-up timing ablations (results are wrong): 1 no history walk, 2 no appends, 4 no final sort, 8 no MFMA, 16 no gather
#endif
#ifndef PDA_V4_ABL
#define PDA_V4_ABL 0      // timing-only ablations (results are wrong): 1 no filter, 2 no hand-over between MFMA waves, 4 no tile loads, 8 rescoring waves leave at once
#endif
#ifdef PDA_V4_PROF
// profiling build only (tools/build_variant.sh prof -DPDA_V4_PROF): cycle counters summed over waves
__device__ unsigned long long pda_prof4[24];
#define PROF_T0(v) const long long v = __builtin_readcyclecounter()
#define PROF_T1(v, slot) prof[slot] += (unsigned long long)(__builtin_readcyclecounter() - v)
#define PROF_INC(slot, x) prof[slot] += (unsigned long long)(x)
#define PROF_FLUSH(first, last) do { if (lane == 0) for (int pq = (first); pq <= (last); ++pq) atomicAdd(&pda_prof4[pq], prof[pq]); } while (0)
#else
#define PROF_T0(v) do {} while (0)
#define PROF_T1(v, slot) do {} while (0)
#define PROF_INC(slot, x) do {} while (0)
#define PROF_FLUSH(first, last) do {} while (0)
#endif
constexpr unsigned kSpinMax = 1u << 26;   // every spin is bounded: a protocol error sets stats[0] and leaves instead of hanging the GPU


struct Args4 {
    const void* U;               // f32 or bf16 [n_users_total, d]
    const void* I;               // f32 or bf16 [n_items_local, d]   (exact rescoring)
    const float* pop;            // f32 [n_items_local] or NULL
    const int32_t* users;
    const int64_t* hist_indptr;
    const int32_t* hist_indices; // GLOBAL ids, ascending per row
    uint64_t* out_keys;          // [n_splits, n_users_blk, K]: warm4 writes, sweep4 reads and overwrites
    const unsigned char* rows;   // prep: padded item rows in visiting order, whole 64-item tiles
    const float* sufA;           // [n_tiles] suffix bounds per 64-item tile, or NULL: no early termination
    const float* sufB;
    const int* pos_of;           // [n_items_local] local id -> visiting position, or NULL: identity
    unsigned* stats;             // workspace as v3: u32 at +4 pairs rescored, u64 at +8 32-item tiles x 128-user tiles scored
    uint64_t* lists_ws;          // list slots of the workgroups whose lists live in HBM (Geo4::GL): [workgroup][UT][kCap4]
    int32_t* regroup_ws;         // [1024 + 2 n_users_blk] or NULL: bins | bin of every user | row_perm (launch4 fills them)
    float* pred_ws;              // [n_users_blk][2] or NULL: warm4_kernel leaves (a lower bound of the K-th value, padded ||u||) for the regrouping
    const int32_t* row_perm;     // [n_users_blk] or NULL: sweep row -> block row (users regrouped by predicted stopping tile)
    const uint32_t* hmask_ws;    // [workgroups of warm4_kernel][128][2 kWarmTiles]: train-item bits of the warm positions, or NULL
    const uint32_t* bloom;       // [n_users_blk][32]: 1024-bit Bloom filter (two hashes) of every block row's train items, or NULL
    const int* prep_hdr;         // header of the item prep: word 2 = built with a visiting order
    const float* seed;           // [n_users_blk] or NULL: an external LOWER bound of every user's final K-th value (other item shards)
    int n_users_blk, item_offset, n_items_local, hist_row_mode, K, n_splits, n_tiles;
    int warm_tiles;              // 1 .. kWarmTiles: 64-item tiles per split scored exactly by warm4_kernel
    int warm_sorted;             // the warm-up hands its lists over sorted (phase 1 alone: pda_topk_kth_value reads ranks) or as they are
    uint64_t* handover;          // [n_splits, n_users_blk, kCap4] or NULL: warm-up and sweep of ONE call hand the lists over here, K .. kCap4 keys
                                 // per row (zero-padded), instead of exactly K through out_keys
    // the huge geometry (pda_v5_sweep.h): the popularity-scaled, swizzled item image and its per-half-tile (pmax, nmax); the user block as
    // bf16 MFMA operands and the users' padded norms (workspace)
    const unsigned char* rows5;
    const float* meta5;
    const unsigned char* ufrag;
    const float* unorm;
    int prep_hdr_pop;            // host copy of "the prep was built with a popularity" (the image is scaled by it): set by the entry points
    int warm_final;              // the huge geometry behind a one-call warm-up: warm4_kernel hands SORTED lists of K keys over and writes them to
                                 // out_keys as well -- the sweep then sorts and emits only the rows it appended to (a few per cent)
    int warm_shared;             // n_splits > 1, one call: ONE warm-up per user on tiles 0 .. warm_tiles - 1 of the whole visiting order, handed to
                                 // split 0; the other splits start empty and prune against its K-th value (seed); see warm_tiles_of
    float* seed_out;             // [n_users_blk] or NULL: warm4_kernel leaves a lower bound of every row's K-th value here (the shared warm-up's seed)
    const int* n_users_dev;      // or NULL: a device-side count of the rows that exist (the funnel's exact fallback: the block is sized for the worst
                                 // case, workgroups whose user tile lies beyond the count leave at once; rows beyond it inside a tile score the padding user)
    int lists_empty;             // phase 4: no warm-up ran on this catalogue -- every split starts with empty lists and sweeps ALL its tiles against
                                 // the caller's seed (the warm-up ran elsewhere: on replicated hot items, pda_amd.dist)
};

// ---------------------------------------------------------------------------------------------------------------------
// prep: padded rows in visiting order
// ---------------------------------------------------------------------------------------------------------------------

// one row per D/8 threads.  pos >= n: a null item (zero vector, constant slot -3e38: never a candidate, pop NaN)
template <int D, bool BF>
__global__ void __launch_bounds__(256) prep4_kernel(const void* __restrict__ I, const float* __restrict__ pop, const int* __restrict__ order,
                                                    int n, int n_pad, unsigned char* __restrict__ rows, int* __restrict__ pos_of,
                                                    int* __restrict__ hdr, unsigned char* __restrict__ rows5, int* __restrict__ meta5, u32x4* __restrict__ pinfo,
                                                    int f16img) {
    constexpr int TPR = D / 8, RB = row_bytes(D);
    const int pos = blockIdx.x * (256 / TPR) + threadIdx.x / TPR, e = threadIdx.x % TPR;
    if (pos >= n_pad) return;
    unsigned char* rp = rows + (size_t)pos * RB;
    float ss = 0.f, popv = 1.0f, rs = 0.f;
    int src = -1;
    if (pos < n) {
        src = pos;
        if (order) {
            src = order[pos];
            if (src < 0 || src >= n) { if (e == 0) atomicOr(hdr, 1); src = 0; }
            else if (e == 0 && atomicExch(&pos_of[src], pos) != -1) atomicOr(hdr, 1);   // not a permutation
        } else if (e == 0) {
            pos_of[pos] = pos;                      // natural order: the identity
        }
        if (pop) popv = pop[src];
        const f32x4 a = pda_load4<BF>(I, (size_t)src * D + 8 * e);
        const f32x4 b = pda_load4<BF>(I, (size_t)src * D + 8 * e + 4);
        u32x4 hq, lq;
        split8(a, b, hq, lq);                       // bf16 tables: the RNE of a bf16 value is itself
        *reinterpret_cast<u32x4*>(rp + 16 * e) = hq;
#pragma unroll
        for (int k = 0; k < 4; ++k) ss += a[k] * a[k] + b[k] * b[k];
        {
            // the huge geometry's image: chunk e of row pos & 31 of half-tile pos >> 5, scaled by the popularity (NaN: a zero row -- such an
            // item never ranks), at the swizzled place the MFMA waves read it from (pda_v5_sweep.h)
            const float pv = pop ? ((popv == popv) ? popv : 0.f) : 1.0f;
            f32x4 as = a, bs = b;
#pragma unroll
            for (int k = 0; k < 4; ++k) { as[k] *= pv; bs[k] *= pv; }
            u32x4 hs, ls;
            if (f16img) {                                    // the funnel's prep (pda_item_prep7_*): the image in fp16, its residuals accordingly
                hs = pack_half8(as, bs, rs);
            } else {
                split8(as, bs, hs, ls);
#pragma unroll
                for (int k = 0; k < 4; ++k) {                // the rounding residual of the image row (exact differences): ||i' - bf16(i')||^2
                    const float r0 = as[k] - __uint_as_float((k & 1) ? (hs[k >> 1] & 0xFFFF0000u) : (hs[k >> 1] << 16));
                    const float r1 = bs[k] - __uint_as_float((k & 1) ? (hs[2 + (k >> 1)] & 0xFFFF0000u) : (hs[2 + (k >> 1)] << 16));
                    rs += r0 * r0 + r1 * r1;
                }
            }
            const int r5 = pos & 31;
            const int sw = D >= 128 ? (r5 & 15) : ((r5 >> 1) & 7);
            *reinterpret_cast<u32x4*>(rows5 + (size_t)(pos >> 5) * (64 * D) + r5 * (2 * D) + ((e ^ sw) << 4)) = hs;
        }
    } else {
        const u32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(rp + 16 * e) = z;
        {
            const int r5 = pos & 31;
            const int sw = D >= 128 ? (r5 & 15) : ((r5 >> 1) & 7);
            *reinterpret_cast<u32x4*>(rows5 + (size_t)(pos >> 5) * (64 * D) + r5 * (2 * D) + ((e ^ sw) << 4)) = z;
        }
    }
#pragma unroll
    for (int o = TPR / 2; o > 0; o >>= 1) { ss += __shfl_xor(ss, o, 64); rs += __shfl_xor(rs, o, 64); }
    if (e == 0) {
        const float v = sqrtf(ss) * 1.0009765625f * 1.0001f;
        const float rv = sqrtf(rs) * 1.0009765625f * 1.0001f;       // padded ||i' - bf16(i')||: the funnel's bound uses the ACTUAL rounding residuals (pda_v7_funnel.h)
        {
            if (pos < n) {                                   // (pmax, nmax) of the half-tile: maxima of non-negative floats = maxima of their bit patterns
                const float pv = pop ? ((popv == popv) ? popv : 0.f) : 1.0f;
                atomicMax(&meta5[4 * (pos >> 5)], __float_as_int(pop ? pv : 0.f));
                atomicMax(&meta5[4 * (pos >> 5) + 1], __float_as_int(pv * v * 1.000001f));
                atomicMax(&meta5[4 * (pos >> 5) + 2], __float_as_int(rv));
            }
        }
        // the B side of the extra k-step (see pda_score_topk_v2.hip:item_prep_kernel): k 0..7 pieces of (1/pop)' rounded
        // down (1 for the raw head), k 8 the constant 1 (raw: 8e-6), k 9 8e-6 (raw: 0), k 10 the padded norm rounded up
        uint32_t p1 = 0x3F80u, p2 = 0, p3 = 0, k1 = 0, k2 = 0;
        if (pop) {
            const float ip = (popv > 0.f) ? fminf((1.0f / popv) * 0.9999995f, 1.0e6f) : 1.0e6f;
            bf16_split3(ip, p1, p2, p3);
            k1 = 0x3F80u;
            k2 = bf16_up(8.0e-6f);
            if (!(popv == popv)) k1 = 0xFF61u;      // NaN popularity: never a candidate
        } else {
            k1 = bf16_up(8.0e-6f);
        }
        if (pos >= n) { k1 = 0xFF61u; k2 = 0; p1 = 0x3F80u; p2 = p3 = 0; }
        u32x4 lo4, hi4;
        lo4[0] = p1 | (p2 << 16);
        lo4[1] = p1 | (p2 << 16);
        lo4[2] = p3 | (p1 << 16);
        lo4[3] = p3 | (p2 << 16);
        hi4[0] = k1 | (k2 << 16);
        hi4[1] = pos < n ? bf16_up(v) : 0u;
        hi4[2] = 0;
        hi4[3] = 0;
        *reinterpret_cast<u32x4*>(rp + 2 * D) = lo4;
        *reinterpret_cast<u32x4*>(rp + 2 * D + 16) = hi4;
        u32x4 tail;
        tail[0] = pos < n ? __float_as_uint(pop ? popv : 1.0f) : 0x7FC00000u;     // null item: NaN (no comparison is ever true)
        tail[1] = (uint32_t)(pos < n ? src : 0);
        tail[2] = __float_as_uint(pos < n ? v : 0.f);
        tail[3] = pos < n ? __float_as_uint(rv) : 0u;                 // (the residual norm of the row's huge-geometry image)
        *reinterpret_cast<u32x4*>(rp + 2 * D + 32) = tail;
        pinfo[pos] = u32x4{tail[2], tail[3], tail[1], tail[0]};                 // (||i||, ||i' - i~'||, local id, popularity)
    }
}

// per-tile maxima of |pop|, |pop| ||i|| and ||i||, then suffix maxima (single workgroup)
__global__ void __launch_bounds__(256) tile_bound4_kernel(const unsigned char* __restrict__ rows, int rb, int d2, int n_tiles, int has_pop,
                                                          float* __restrict__ tA, float* __restrict__ tB, float* __restrict__ tR) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    float ma = 0.f, mb = 0.f, mr = 0.f;
    for (int q = 0; q < 64; ++q) {
        const float* tl = reinterpret_cast<const float*>(rows + ((size_t)t * 64 + q) * rb + d2 + 32);
        const float p = tl[0], nr = tl[2];
        if (p == p) {                       // null items and NaN popularities never rank
            const float pa = has_pop ? fabsf(p) : 0.f;
            ma = fmaxf(ma, pa);
            mb = fmaxf(mb, pa * nr * 1.000001f);
            mr = fmaxf(mr, nr);
        }
    }
    tA[t] = ma;
    tB[t] = mb;
    tR[t] = mr;
}
__global__ void __launch_bounds__(1024) suffix_max4_kernel(float* __restrict__ tA, float* __restrict__ tB, float* __restrict__ tR, int n_tiles) {
    __shared__ float sa[1024], sb[1024], sr[1024];
    const int per = (n_tiles + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, n_tiles);
    float ma = 0.f, mb = 0.f, mr = 0.f;
    for (int t = lo; t < hi; ++t) { ma = fmaxf(ma, tA[t]); mb = fmaxf(mb, tB[t]); mr = fmaxf(mr, tR[t]); }
    sa[threadIdx.x] = ma;
    sb[threadIdx.x] = mb;
    sr[threadIdx.x] = mr;
    __syncthreads();
    float ra = 0.f, rb = 0.f, rr = 0.f;
    for (int q = threadIdx.x + 1; q < 1024; ++q) { ra = fmaxf(ra, sa[q]); rb = fmaxf(rb, sb[q]); rr = fmaxf(rr, sr[q]); }
    for (int t = hi - 1; t >= lo; --t) {
        ra = fmaxf(ra, tA[t]);
        rb = fmaxf(rb, tB[t]);
        rr = fmaxf(rr, tR[t]);
        tA[t] = ra;
        tB[t] = rb;
        tR[t] = rr;
    }
}


int run_prep4(const void* I_shard, bool bf16, const float* pop, const int* order, int n, int d, void* prep, hipStream_t s, int f16img = 0) {
    if (d != 64 && d != 128 && d != 256) return PDA_ERR_UNSUPPORTED;
    const Prep4Layout L = prep4_layout(n, d);
    unsigned char* pb = reinterpret_cast<unsigned char*>(prep);
    int* hdr = reinterpret_cast<int*>(pb + L.hdr);
    int* pos_of = reinterpret_cast<int*>(pb + L.pos_of);
    if (hipMemsetAsync(hdr, 0, 256, s) != hipSuccess) return PDA_ERR_LAUNCH;
    if (order && hipMemsetAsync(pos_of, 0xFF, (size_t)n * 4, s) != hipSuccess) return PDA_ERR_LAUNCH;
    if (order && hipMemsetAsync(hdr + 2, 1, 1, s) != hipSuccess) return PDA_ERR_LAUNCH;          // header word 2 := 1: a visiting order was given
    if (pop && hipMemsetAsync(hdr + 1, 1, 1, s) != hipSuccess) return PDA_ERR_LAUNCH;            // header word 1 := 1: the test operands carry 1/pop pieces
    if (f16img && hipMemsetAsync(hdr + 3, 1, 1, s) != hipSuccess) return PDA_ERR_LAUNCH;         // header word 3 := 1: the half-tile image is fp16 (the funnel's)
    const int n_pad = L.n_tiles * 64;
    if (hipMemsetAsync(pb + L.meta5, 0, (size_t)L.n_tiles * 2 * 16, s) != hipSuccess) return PDA_ERR_LAUNCH;
    unsigned char* r5 = pb + L.rows5;
    int* m5 = reinterpret_cast<int*>(pb + L.meta5);
    u32x4* pi5 = reinterpret_cast<u32x4*>(pb + L.pinfo);
#define PDA_P4(DD)                                                                                                              \
    case DD: {                                                                                                                  \
        constexpr int RPB = 256 / (DD / 8);                                                                                     \
        const dim3 grid((unsigned)((n_pad + RPB - 1) / RPB));                                                                   \
        if (bf16) hipLaunchKernelGGL((prep4_kernel<DD, true>), grid, dim3(256), 0, s, I_shard, pop, order, n, n_pad, pb + L.rows, pos_of, hdr, r5, m5, pi5, f16img); \
        else hipLaunchKernelGGL((prep4_kernel<DD, false>), grid, dim3(256), 0, s, I_shard, pop, order, n, n_pad, pb + L.rows, pos_of, hdr, r5, m5, pi5, f16img);    \
        break;                                                                                                                  \
    }
    switch (d) { PDA_P4(64) PDA_P4(128) PDA_P4(256) }
#undef PDA_P4
    PDA_CHECK_LAUNCH();
    float* tA = reinterpret_cast<float*>(pb + L.sufA);
    float* tB = reinterpret_cast<float*>(pb + L.sufB);
    float* tR = reinterpret_cast<float*>(pb + L.sufR);
    hipLaunchKernelGGL(tile_bound4_kernel, dim3((unsigned)((L.n_tiles + 255) / 256)), dim3(256), 0, s, pb + L.rows, row_bytes(d), 2 * d,
                       L.n_tiles, pop ? 1 : 0, tA, tB, tR);
    PDA_CHECK_LAUNCH();
    hipLaunchKernelGGL(suffix_max4_kernel, dim3(1), dim3(1024), 0, s, tA, tB, tR, L.n_tiles);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// shared device pieces
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned lds_ld(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#define PDA_CBAR() asm volatile("" ::: "memory")

__device__ __forceinline__ f32x16 zero16v() {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return z;
}

// how many of split s's first tiles the warm-up has scored: warm_tiles of its own -- or, behind a shared warm-up (tiles 0 .. warm_tiles - 1
// of the whole order), those of them that are the split's: s, s + S, ... below warm_tiles
__device__ __forceinline__ int warm_tiles_of(const Args4& g, int split) {
    return !g.warm_shared ? g.warm_tiles : (split < g.warm_tiles ? (g.warm_tiles - split + g.n_splits - 1) / g.n_splits : 0);
}

// append one exact key per flagged lane to its row's list; compaction (whole wave) when a list is full.  Returns true
// when a threshold may have changed.
template <int CAP, bool GLB = false>
__device__ __forceinline__ bool append_keys(bool p, int lrow, float tt, uint64_t key, uint64_t* lists, int* cntl, float* taul, int row0,
                                            int n_rows, int K, int lane, unsigned* uns = nullptr, [[maybe_unused]] unsigned long long* prof = nullptr) {
    bool changed = false;
    for (;;) {
        bool ov = false;
        if (p) {
            const int slot = atomicAdd(&cntl[lrow], 1);
            if (slot < CAP) lists[(size_t)lrow * CAP + slot] = key;
            else ov = true;
        }
        uint64_t todo = __ballot(ov);
        if (todo == 0ull) break;
        list_sync<GLB>();
        // the rows that overflowed are the rows of the lanes that say so (a row sits at CAP untouched until an append fails): no
        // scan over the wave's counters -- two LDS round trips per event on the rescoring path
        bool again = false;
        while (todo) {
            const int rr = __builtin_amdgcn_readlane(lrow, __builtin_ctzll(todo));
            const uint64_t mine_m = __ballot(ov && lrow == rr);
            todo &= ~mine_m;
#ifdef PDA_V4_PROF
            const long long tc0 = __builtin_readcyclecounter();
#endif
            const float tau = compact_list<CAP, GLB>(lists + (size_t)rr * CAP, &cntl[rr], &taul[rr], K, lane, uns ? &uns[rr >> 5] : nullptr, 1u << (rr & 31));
#ifdef PDA_V4_PROF
            if (prof) { prof[22] += (unsigned long long)(__builtin_readcyclecounter() - tc0); prof[23] += 1; }
#endif
            // the row holds K keys now and its threshold is known: the lanes that failed on it take the slots behind them directly
            // (an atomic and a threshold read per retry were two more round trips per event)
            const bool mine = ((mine_m >> lane) & 1ull) != 0ull;
            const bool keep = mine && tt >= tau;
            const uint64_t km = __ballot(keep);
            if (km != 0ull) {
                const int before = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0));
                const int nk = __popcll(km);
                if (keep && K + before < CAP) lists[(size_t)rr * CAP + K + before] = key;
                if (keep && K + before >= CAP) again = true;            // (more than CAP - K keys for one row in one pass)
                if (lane == 0) cntl[rr] = min(K + nk, CAP);
            }
        }
        list_sync<GLB>();
        changed = true;
        p = again;
    }
    return changed;
}

// Train items among the warm positions of a 128-user tile -> bit masks [128][2 kWarmTiles] in LDS.  A wave walks the histories of its 32 rows as ONE
// flat list of (row, entry) pairs, 64 x 8 of them per round with all eight id loads and then all eight position loads of a lane in flight (two
// dependent loads per train item); a row's length no longer decides the number of dependent rounds.  Measured (round 6, tools/warm_ablate_small.sh and
// two load ablations): the walk is 70 - 80 of the warm-up's ~190 us at config 2 (150 train items per user) -- ~40 the id loads, ~16 the position
// gathers, the rest index arithmetic -- in THIS form as in the four-rows-at-a-time form of rounds 2 - 5: the rewrite bought nothing measurable there
// (0.197 vs 0.186 - 0.200 ms lease to lease) and is kept for its bounded worst case (a single 800-item row used to cost 13 dependent pairs).
__device__ __forceinline__ void warm_hist_walk(const Args4& g, unsigned* hmask, int utile, int split, int nwarm, int wave, int lane) {
#ifndef PDA_W4_WALK_UN
#define PDA_W4_WALK_UN 8
#endif
    constexpr int UN = PDA_W4_WALK_UN;
    long long hb_l = 0;
    int len_l = 0;
    {
        const int rb = utile * kUserTile + wave * 32 + (lane & 31);
        if (lane < 32 && rb < g.n_users_blk) {
            const int64_t hr = g.hist_row_mode == PDA_HIST_BY_USER_ID ? (int64_t)g.users[rb] : (int64_t)rb;
            hb_l = g.hist_indptr[hr];
            len_l = (int)min((long long)0x3FFFFFF, g.hist_indptr[hr + 1] - hb_l);          // (a row of more than 2^26 train items cannot exist: n_items <= 2^26)
        }
    }
    // exclusive prefix of the rows' lengths over lanes 0 .. 31 (lanes 32 .. 63 hold the total)
    int inc = len_l;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if ((lane & 31) >= o) inc += v;
    }
    const int total = __shfl(inc, 31, 64);
    const int pre_l = lane < 32 ? inc - len_l : total;
    const int warm_end = 64 * nwarm * g.n_splits;       // visiting positions behind the last warm tile of any split
    for (int base = 0; base < total; base += 64 * UN) {
        int row[UN], loc[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int f = base + q * 64 + lane;
            // the row of flat entry f: the last lane r < 32 with pre[r] <= f (rows without entries are skipped by construction)
            int r = 0;
#pragma unroll
            for (int st = 16; st >= 1; st >>= 1) {
                const int pm = __shfl(pre_l, r + st, 64);
                if (pm <= f) r += st;
            }
            const int pr = __shfl(pre_l, r, 64);
            const long long hbr = __shfl(hb_l, r, 64);
            row[q] = r;
            loc[q] = f < total ? g.hist_indices[hbr + (f - pr)] - g.item_offset : -1;
        }
        int pp[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const bool in = loc[q] >= 0 && loc[q] < g.n_items_local;
            pp[q] = in ? (g.pos_of ? g.pos_of[loc[q]] : loc[q]) : -1;
        }
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            if (pp[q] >= 0 && pp[q] < warm_end) {        // (the cheap test first: the integer division below is ~80 instructions, and ~99 % of the train items fail here)
                const int T = pp[q] >> 6;
                if (T % g.n_splits == split) {
                    const int qq = (T - split) / g.n_splits;
                    if (qq < nwarm) atomicOr(&hmask[(wave * 32 + row[q]) * (2 * kWarmTiles) + 2 * qq + ((pp[q] & 63) >> 5)], 1u << (pp[q] & 31));
                }
            }
        }
    }
}

// The same as a kernel of its own (grid and workgroup = warm4_kernel's): at eight waves per SIMD the two dependent, mostly missing
// loads per train item are hidden; inside warm4_kernel (256 VGPRs, two waves per SIMD) they were a quarter of its time.
__global__ void __launch_bounds__(kThreads) warm_mask4_kernel(Args4 g, uint32_t* __restrict__ out) {
    __shared__ unsigned hmask[kUserTile * 2 * kWarmTiles];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.x % g.n_splits, utile = blockIdx.x / g.n_splits;
    if (g.n_users_dev != nullptr && utile * kUserTile >= *g.n_users_dev) return;
    const int nwarm = min(g.warm_tiles, split_tiles(g.n_tiles, split, g.n_splits));
    for (int q = tid; q < kUserTile * 2 * kWarmTiles; q += kThreads) hmask[q] = 0u;
    __syncthreads();
    warm_hist_walk(g, hmask, utile, split, nwarm, wave, lane);
    __syncthreads();
    uint32_t* dst = out + (size_t)blockIdx.x * kUserTile * 2 * kWarmTiles;
    for (int q = tid; q < kUserTile * 2 * kWarmTiles; q += kThreads) dst[q] = hmask[q];
}

// ---------------------------------------------------------------------------------------------------------------------
// warm-up: the first kWarmTiles tiles of every split, exact (fp32 matrix cores, k order of v1), lists -> out_keys
// ---------------------------------------------------------------------------------------------------------------------
template <int D, int HEAD, bool BF>
__global__ void __launch_bounds__(kThreads, (D <= 128 ? 2 : 1)) warm4_kernel(Args4 g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RB = row_bytes(D);
    constexpr int NC = D / 8;                 // k-chunks of 8
    constexpr int CPR4 = D / 4;               // 16-byte chunks per fp32 row
    constexpr int NLD4 = (32 * CPR4) / kThreads;
    constexpr int CAP = kCap4;
    float* Bt = reinterpret_cast<float*>(smem);                                              // [32][D] fp32, swizzled
    uint64_t* lists = reinterpret_cast<uint64_t*>(smem + 32 * D * 4);                        // [128][CAP]
    int* cntl = reinterpret_cast<int*>(lists + (size_t)kUserTile * CAP);                     // [128]
    float* taul = reinterpret_cast<float*>(cntl + kUserTile);                                // [128]
    unsigned* hmask = reinterpret_cast<unsigned*>(taul + kUserTile);                         // [128][2 kWarmTiles]
    float* popw = reinterpret_cast<float*>(hmask + kUserTile * 2 * kWarmTiles);              // [32]
    int* idw = reinterpret_cast<int*>(popw + 32);                                            // [32]
    float* predt = reinterpret_cast<float*>(idw + 32);                                       // [128] for pred_ws: K-th value (bound)
    float* nul = predt + kUserTile;                                                          // [128] for pred_ws: padded norm

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int split = blockIdx.x % g.n_splits, utile = blockIdx.x / g.n_splits;
    if (g.n_users_dev != nullptr && utile * kUserTile >= *g.n_users_dev) return;
    const int K = g.K;
    const int nt = split_tiles(g.n_tiles, split, g.n_splits);
    const int nwarm = min(g.warm_tiles, nt);

    const int row_blk = utile * kUserTile + wave * 32 + j;
    const bool row_ok = row_blk < g.n_users_blk;
    const int uid = row_ok ? g.users[row_blk] : 0;

    if (lane < 32) {
        cntl[wave * 32 + lane] = 0;
        taul[wave * 32 + lane] = row_ok ? -INFINITY : INFINITY;
    }
    for (int q = tid; q < kUserTile * 2 * kWarmTiles; q += kThreads) hmask[q] = 0u;
    __syncthreads();
    // train items among the warm positions: precomputed by warm_mask4_kernel (hmask_ws), or walked here
    if (g.hmask_ws != nullptr) {
        const uint32_t* src = g.hmask_ws + (size_t)blockIdx.x * kUserTile * 2 * kWarmTiles;
        for (int q = tid; q < kUserTile * 2 * kWarmTiles; q += kThreads) hmask[q] = src[q];
    } else if (g.hist_indptr != nullptr && !(PDA_W4_ABL & 1)) {
        warm_hist_walk(g, hmask, utile, split, nwarm, wave, lane);
    }
    f32x4 areg[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row_ok) v = pda_load4<BF>(g.U, (size_t)uid * D + 4 * h + 8 * c);
        areg[c] = v;
    }
    if (g.pred_ws != nullptr) {          // the row's padded norm, as the sweep's votes use it (stop_predict4_kernel)
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) ss += areg[c][0] * areg[c][0] + areg[c][1] * areg[c][1] + areg[c][2] * areg[c][2] + areg[c][3] * areg[c][3];
        ss += __shfl_xor(ss, 32, 64);
        if (h == 0) nul[wave * 32 + j] = sqrtf(ss) * 1.0009765625f * 1.0001f;
    }
    const float* brow = Bt + j * D;
    const int bswz = swz<D>(j);
    uint64_t* my_lists = lists + (size_t)(wave * 32) * CAP;
    // All scores of the warm-up stay in registers (ordered-uint form, 0 = not a candidate: beyond the shard, a train item,
    // NaN), newest half-tile first.
    constexpr int NHT = 2 * kWarmTiles;
    static_assert(NHT == 8, "eight named half-tile registers below");
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    // (separate vector values shifted by assignment: an array shifted in a rolled loop ends up in scratch memory)
    u32x16 o0 = 0u, o1 = 0u, o2 = 0u, o3 = 0u, o4 = 0u, o5 = 0u, o6 = 0u, o7 = 0u;
    int i0 = 0, i1 = 0, i2 = 0, i3 = 0, i4 = 0, i5 = 0, i6 = 0, i7 = 0;
#pragma unroll 1
    for (int ht = 0; ht < 2 * nwarm; ++ht) {
        const int w = ht >> 1, cb = ht & 1;
        const int t = split + w * g.n_splits;
        __syncthreads();                  // the previous block has been read by everyone (and hmask is complete)
        if (tid < 32) {
            const float* tl = reinterpret_cast<const float*>(g.rows + ((size_t)t * 64 + 32 * cb + tid) * RB + 2 * D + 32);
            popw[tid] = tl[0];
            idw[tid] = reinterpret_cast<const int*>(tl)[1];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NLD4; ++q) {
            if constexpr ((PDA_W4_ABL & 16) != 0) break;
            const int id = tid + kThreads * q;
            const int jj = id / CPR4, ch = id % CPR4;
            // (null items of a tail tile carry id 0: a valid row, masked by okw below)
            *reinterpret_cast<f32x4*>(Bt + jj * D + 4 * (ch ^ swz<D>(jj))) = pda_load4<BF>(g.I, (size_t)idw[jj] * D + 4 * ch);
        }
        __syncthreads();
        f32x16 acc0 = zero16v(), acc1 = zero16v();
#pragma unroll
        for (int c = 0; c < NC; c += 2) {
            if constexpr ((PDA_W4_ABL & 8) != 0) break;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(brow + 4 * ((2 * c + h) ^ bswz));
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(brow + 4 * ((2 * c + 2 + h) ^ bswz));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[c][q], b0[q], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[c + 1][q], b1[q], acc1, 0, 0, 0);
            }
        }
        const f32x16 accx = acc0 + acc1;
        const bool okw = (t * 64 + 32 * cb + j) < g.n_items_local;
        const float pv = popw[j];
        const unsigned hb_mine = hmask[(wave * 32 + j) * (2 * kWarmTiles) + 2 * w + cb];
        const bool any_hb = __any(hb_mine != 0u);
        o7 = o6; o6 = o5; o5 = o4; o4 = o3; o3 = o2; o2 = o1; o1 = o0;
        i7 = i6; i6 = i5; i5 = i4; i4 = i3; i3 = i2; i2 = i1; i1 = i0;
        i0 = g.item_offset + idw[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            float sc = accx[r];
            if constexpr (HEAD == PDA_HEAD_POP) sc = (sc > 0.0f ? sc + 1.0f : __expf(sc)) * pv;
            bool p = okw && (sc >= -INFINITY);                                             // (NaN never ranks)
            if (any_hb) {
                const uint32_t hbr = (uint32_t)__shfl((int)hb_mine, row, 64);              // train items never enter
                if ((hbr >> j) & 1u) p = false;
            }
            o0[r] = p ? pda_ordf(sc + 0.0f) : 0u;
        }
    }
    const u32x16 ordv[NHT] = {o0, o1, o2, o3, o4, o5, o6, o7};
    const int itemv[NHT] = {i0, i1, i2, i3, i4, i5, i6, i7};
    // Per row (two at a time: one per half-wave) a threshold t with K <= #(scores >= t) <= CAP by bitwise descent over the
    // ordered-uint scores (counts = ballots over the register file), then the survivors go to their list slots directly.
    // More than CAP survivors at the exact K-th value (ties): the general append path with its compactions.
    // Two registers (= four rows) per descent: the steps of one descent are a dependent chain (compare, ballot, popcount, select),
    // two independent chains share the wave's issue slots.
    // unsorted hand-over through out_keys: exactly K survivors (K slots per row) -- a bisection down to the K-th value itself,
    // ~20 rounds of ballots over the 256 scores of a row, 0.25 of the warm-up's 0.6 ms at 262 144 users.  Through the workspace
    // (g.handover: CAP slots per row) any threshold with K .. CAP survivors will do: a handful of rounds.
    const int cap_t = (g.warm_sorted || g.handover != nullptr) ? CAP : K;
    auto emit_row = [&](const int r, const uint32_t t, const int c_t) __attribute__((always_inline)) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int lrow = wave * 32 + row;
        if (j == 0) predt[lrow] = c_t >= K ? pda_unordf(t) : -INFINITY;       // K scores at or above t
        if (__any(c_t > cap_t)) {
#pragma unroll
            for (int k = 0; k < NHT; ++k) {
                const bool p = ordv[k][r] >= t;
                const uint64_t key = ((uint64_t)ordv[k][r] << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)itemv[k]);
                append_keys<CAP>(p, lrow, pda_unordf(ordv[k][r]), key, lists, cntl, taul, wave * 32, 32, K, lane);
            }
        } else {
            int run = 0;
#pragma unroll
            for (int k = 0; k < NHT; ++k) {
                const bool p = ordv[k][r] >= t;
                const uint64_t bm = __ballot(p);
                const uint32_t bh = h ? (uint32_t)(bm >> 32) : (uint32_t)bm;
                const int slot = run + __popc(bh & ((1u << j) - 1u));
                if (p) lists[(size_t)lrow * CAP + slot] = ((uint64_t)ordv[k][r] << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)itemv[k]);
                run += __popc(bh);
            }
            if (j == 0) cntl[lrow] = run;
        }
    };
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        if constexpr ((PDA_W4_ABL & 2) != 0) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int k = 0; k < NHT; ++k) asm volatile("" ::"v"(ordv[k][r]), "v"(ordv[k][r + 1]));
#endif
            continue;
        }
        auto count_ge2 = [&](uint32_t ca, uint32_t cb, int& na, int& nb) __attribute__((always_inline)) {
            int loa = 0, hia = 0, lob = 0, hib = 0;
#pragma unroll
            for (int k = 0; k < NHT; ++k) {
                const uint64_t bma = __ballot(ordv[k][r] >= ca);
                const uint64_t bmb = __ballot(ordv[k][r + 1] >= cb);
                loa += __popc((uint32_t)bma);
                hia += __popc((uint32_t)(bma >> 32));
                lob += __popc((uint32_t)bmb);
                hib += __popc((uint32_t)(bmb >> 32));
            }
            na = h ? hia : loa;
            nb = h ? hib : lob;
        };
        int ca_t, cb_t;
        count_ge2(1u, 1u, ca_t, cb_t);
        uint32_t ta = 1u, tb = 1u;
        const bool needa = ca_t > cap_t, needb = cb_t > cap_t;
        if (__any(needa || needb)) {
            // bisection between the row's smallest and largest score in the ordered-uint domain (~ the log domain for
            // positive scores: a handful of steps for smooth or heavy-tailed scores, <= 32 always); invariant:
            // #(>= t) = c_t >= K, #(>= hi) < K
            uint32_t mxa = 0u, mna = 0xFFFFFFFFu, mxb = 0u, mnb = 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < NHT; ++k) {
                const uint32_t va = ordv[k][r], vb = ordv[k][r + 1];
                mxa = max(mxa, va);
                mna = min(mna, va != 0u ? va : 0xFFFFFFFFu);
                mxb = max(mxb, vb);
                mnb = min(mnb, vb != 0u ? vb : 0xFFFFFFFFu);
            }
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) {                      // (xor < 32: stays inside the half-wave)
                mxa = max(mxa, (uint32_t)__shfl_xor((int)mxa, o, 64));
                mna = min(mna, (uint32_t)__shfl_xor((int)mna, o, 64));
                mxb = max(mxb, (uint32_t)__shfl_xor((int)mxb, o, 64));
                mnb = min(mnb, (uint32_t)__shfl_xor((int)mnb, o, 64));
            }
            uint32_t hia = mxa + 1u, hib = mxb + 1u;                 // (mx < 0xFFFFFFFF: NaNs are not candidates)
            if (needa) ta = mna;
            if (needb) tb = mnb;
            for (int it = 0; it < 40; ++it) {
                const bool opena = needa && ca_t > cap_t && hia - ta > 1u;
                const bool openb = needb && cb_t > cap_t && hib - tb > 1u;
                if (!__any(opena || openb)) break;
                const uint32_t mida = opena ? ta + ((hia - ta) >> 1) : ta;
                const uint32_t midb = openb ? tb + ((hib - tb) >> 1) : tb;
                int cnta, cntb;
                count_ge2(mida, midb, cnta, cntb);
                if (opena) {
                    if (cnta >= K) {
                        ta = mida;
                        ca_t = cnta;
                    } else {
                        hia = mida;
                    }
                }
                if (openb) {
                    if (cntb >= K) {
                        tb = midb;
                        cb_t = cntb;
                    } else {
                        hib = midb;
                    }
                }
            }
        }
        emit_row(r, ta, ca_t);
        emit_row(r + 1, tb, cb_t);
    }
    pda_wave_sync();
    if (tid == 0) atomicAdd(reinterpret_cast<unsigned long long*>(g.stats + 2), (unsigned long long)(2 * nwarm));
    for (int rr = 0; rr < 32; ++rr) {
        uint64_t* buf = my_lists + rr * CAP;
        // sorted hand-over, or a row that went through the general append path (ties at its K-th value) and may hold more than K
        if (((PDA_W4_ABL & 4) == 0) && (g.warm_sorted || g.warm_final || (g.handover == nullptr && __builtin_amdgcn_readfirstlane(cntl[wave * 32 + rr]) > K)))
            compact_list<CAP>(buf, &cntl[wave * 32 + rr], &taul[wave * 32 + rr], K, lane);
        const int c = cntl[wave * 32 + rr];
        const int rb = utile * kUserTile + wave * 32 + rr;
        if (g.seed_out != nullptr) {
            // the shared warm-up's seed: the smallest of the row's (>= K) keys -- at least K items reach it; -inf for a shorter list
            uint32_t mn = lane < c ? (uint32_t)(buf[lane] >> 32) : 0xFFFFFFFFu;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
            if (lane == 0 && rb < g.n_users_blk) g.seed_out[rb] = c >= K ? pda_unordf(mn) : -INFINITY;
        }
        if (g.handover != nullptr) {
            if (rb < g.n_users_blk && lane < CAP) g.handover[((size_t)split * g.n_users_blk + rb) * CAP + lane] = lane < c ? buf[lane] : 0ull;
            // (warm_final: the row is sorted and holds at most K keys -- it is the final answer unless the sweep appends to it)
            if (g.warm_final && rb < g.n_users_blk && lane < K) g.out_keys[((size_t)split * g.n_users_blk + rb) * K + lane] = lane < c ? buf[lane] : 0ull;
        } else if (rb < g.n_users_blk && lane < K) {
            const uint64_t k = lane < c ? buf[lane] : 0ull;
            g.out_keys[((size_t)split * g.n_users_blk + rb) * K + lane] = k;
        }
    }
    if (g.pred_ws != nullptr && lane < 32) {
        const int rb = utile * kUserTile + wave * 32 + lane;
        if (rb < g.n_users_blk) {
            g.pred_ws[2 * (size_t)rb] = predt[wave * 32 + lane];
            g.pred_ws[2 * (size_t)rb + 1] = nul[wave * 32 + lane];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// the sweep
// ---------------------------------------------------------------------------------------------------------------------
// What the matrix pipe needs (tools/ubench/mfma_lds.hip, MI355X, random operands):
//     1 wave per SIMD, 2 accumulator chains   1298 TFLOP/s      1 wave, 4 chains   1543      1 wave, 8 chains   1727
//     2 waves per SIMD, 2 chains each         1994 TFLOP/s  (= the power-limited ceiling: 3 or 4 waves, more chains: the same)
// i.e. back-to-back MFMAs on one accumulator do not fill the pipe; it takes >= 4 independent chains, best from two waves.
// And (tools/ubench/mfma_valu*.hip, profiling build): a VALU read of an accumulator stalls its wave until the pipe has
// worked its way up to that MFMA -- with one MFMA wave per SIMD nothing is queued meanwhile.  Hence:
//   waves 0..7   MFMA waves, two per SIMD: 32 user rows each; per block (NB half-tiles of 32 items = NB chains) NB (NM + 1)
//                MFMAs, then the filter on the block's own accumulators (sign bit of the OR of a lane's registers) while the
//                other MFMA wave of the SIMD has the pipe;
//   waves 8, 9   loaders (LDS-DMA only: a 1 KiB piece costs its issuing wave ~100 cycles of issue time -- in the MFMA waves
//                that was 0.8 of 3.7 ms); block b goes into slot b & 1 once every MFMA wave has released block b - 2;
//   waves 10, 11 rescoring waves (lists, exact rescoring, thresholds) of 128 user rows each.
// No s_barrier after the start.  Hand-over words in LDS, one word per WRITER (a sum over writers cannot tell "everyone is
// past b" from "some are ahead, one is behind"), relaxed LDS atomics; the LDS executes a wave's operations in order:
//   landed[l]     blocks whose pieces (of loader l) have landed        released[w]   blocks MFMA wave w is done reading
// d <= 128: the kernel needs <= 128 VGPRs -- four waves per SIMD: 8 MFMA waves + 4 loaders + 4 rescoring waves (two loaders
// could not keep up: the MFMA waves waited 39 % of their time for tiles); d = 256 (168 VGPRs): 8 + 2 + 2.

#include "pda_v4_block_asm.h"
#ifndef PDA_V4_ASM
#define PDA_V4_ASM 1      // d <= 128: the reads and MFMAs of a block as ONE inline-asm statement with a counted software pipeline
#endif
#ifndef PDA_V4_RSLEEP
#define PDA_V4_RSLEEP 8     // idle rescoring waves: s_sleep between polls of their rings (x 64 cycles)
#endif
#ifndef PDA_V4_LSLEEP
#define PDA_V4_LSLEEP 1     // loaders: s_sleep between polls for a free slot
#endif
#ifndef PDA_V4_GL
#define PDA_V4_GL 2       // exact lists: 0 in LDS, 1 in HBM, 2 = in HBM for d = 256
#endif
#ifndef PDA_V4_HANDOVER_WS
#define PDA_V4_HANDOVER_WS 1   // one-call sweeps: warm-up lists handed over through the workspace with K .. kCap4 keys per row
#endif
#ifndef PDA_V4_ROWS_INTERLEAVED
#define PDA_V4_ROWS_INTERLEAVED 1   // rescoring: a candidate's lanes share every load (d <= 128)
#endif
#ifndef PDA_V4_RESCORE_AHEAD
#define PDA_V4_RESCORE_AHEAD 1   // rescoring waves request the rows of the next pass before the appends of this one
#endif
#ifndef PDA_V4_RING_MANY
#define PDA_V4_RING_MANY 128   // candidate ring entries of the many-candidates geometry (256 / 512: no faster -- later thresholds, more candidates)
#endif
#ifndef PDA_V4_L256
#define PDA_V4_L256 3          // d = 256: loader waves (three: config-5 shard 27.78 -> 26.21 ms dense, 0.51 of the roof; 35 DMA pieces per 64-item block were
                              // too many for two) (of the four waves beside the eight MFMA waves; the rest rescore)
#endif
#ifndef PDA_V4_ASM256
#define PDA_V4_ASM256 1       // d = 256: the block as one asm statement too (config-5 shard: 29.15 -> 28.24 ms dense, 5.43 -> 5.08 ms early-terminating)
#endif
#ifndef PDA_V4_NB256
#define PDA_V4_NB256 2        // d = 256: half-tiles per block.  64-item blocks = two accumulator chains per wave: config-5 shard 30.65 -> 29.10 ms dense,
                              // 5.87 -> 5.39 ms early-terminating (one chain per wave, two waves per SIMD, left the matrix pipe waiting on dependent MFMAs)
#endif
#ifndef PDA_V4_NSLOT_MAX
#define PDA_V4_NSLOT_MAX 5   // (timing experiments raise it: the votes of the early termination are then wrong)
#endif
#ifndef PDA_V4_NSLOT
#define PDA_V4_NSLOT 4    // tile slots in LDS when the lists live in HBM (<= 5: the vote words of the early termination)
#endif
// GLX (d <= 128; round 3): the exact lists in the workspace instead of the LDS, which pays for FOUR tile slots and loaders that run
// ahead.  With two slots the loaders can start on block b + 2 only when the SLOWEST MFMA wave has released block b, and need
// ~900 cycles from there (5 pieces of ~136 issue cycles per loader, then the latency) against ~1 150 cycles of a block's MFMAs:
// any skew between the eight MFMA waves turns into waiting -- 22 % of their time in the dense sweep (cycle counters of the
// profiling build).  Candidate-heavy sweeps (natural order, raw head: hundreds of list insertions per user) keep their lists in
// the LDS: every insertion would be a round trip to L2.  The caller says which (pda_score_topk4_*: early_stop bit 1).
template <int D, int GM = 0>
struct Geo4 {
    static_assert(GM == 0 || GM == 3, "geometries: 0 = default, 3 = many candidates");
    static constexpr bool MANY = D <= 128 && GM == 3;
    static constexpr int MW = MANY ? 4 : kMainWaves;                  // MFMA waves
    static constexpr int UA = 1;                                      // A operands per B read: 32 UA user rows per MFMA wave
    // the exact lists in HBM (workspace) free the LDS for four tile slots (d = 256: 8.1 instead of 9.0 ms on a config-5 shard);
    // 512 users x 57 x 8 B would not fit the LDS anyway
    static constexpr bool GL = PDA_V4_GL == 2 ? D > 128 : PDA_V4_GL != 0;
#ifdef PDA_V4_NBX   /* timing experiment only (results are wrong): NBX half-tiles per block at d <= 128 */
    static constexpr int NB = D <= 128 ? PDA_V4_NBX : 1;
#else
    static constexpr int NB = D <= 128 ? 2 : PDA_V4_NB256;          // half-tiles (32 items) per block; accumulator chains per wave = NB UA
#endif
    static constexpr int LOADERS = D <= 128 ? 4 : PDA_V4_L256;
    static constexpr int RESCORERS = MANY ? 8 : (D <= 128 ? 4 : 4 - PDA_V4_L256);
    static constexpr int ROWS = 32 * UA;                 // user rows per MFMA wave
    // candidate rings: one per MFMA wave, or (MANY) two -- rows 0..15 and 16..31 of the wave -- each with a rescoring wave of its own
    static constexpr int NRINGS = MW > RESCORERS ? MW : RESCORERS;
    static constexpr int RPW = NRINGS / MW;              // rings per MFMA wave
    static constexpr int RPR = ROWS / RPW;               // user rows per ring
    static constexpr int MPR = NRINGS / RESCORERS;       // rings per rescoring wave
    static constexpr int WAVES = MW + LOADERS + RESCORERS;
    static constexpr int UT = MW * ROWS;                 // user rows per workgroup
    static constexpr int RB = row_bytes(D), TB = tile_bytes(D);
    static constexpr int HB = 32 * RB;                   // one half-tile
    static constexpr int BB = NB * HB;                   // one block = one ring slot
    static constexpr int NP = (BB + 1023) / 1024;        // 1 KiB DMA pieces per block; the last one may be half a piece (32 lanes)
    static constexpr int LASTL = (BB % 1024) ? (BB % 1024) / 16 : 64;
    // (MANY: 128 users' lists leave the LDS room for four slots)
    static constexpr int NSLOT = (GL || MANY) ? PDA_V4_NSLOT : 2;
    static constexpr size_t lds_tiles = NSLOT * (size_t)BB;
    // list slots per user row: a full list is compacted to its best K, i.e. every CAP - K insertions (half of what a candidate costs
    // its rescoring wave); 128 users leave the LDS room for 64
    static constexpr int CAP = MANY ? 64 : kCap4;
    static constexpr size_t lds_lists = GL ? 0 : (size_t)UT * CAP * 8;
    static constexpr int RING = MANY ? PDA_V4_RING_MANY : kRing4;      // entries per candidate ring (a push needs 64 free)
    static constexpr size_t lds_total = lds_tiles + lds_lists + (size_t)UT * 8 + kMainWaves * RING * 4 + 512;
    static_assert(NRINGS <= kMainWaves && RPW * MW == NRINGS && MPR * RESCORERS == NRINGS, "ring bookkeeping: eight words each");
    static_assert(NSLOT >= 2 && NSLOT <= PDA_V4_NSLOT_MAX, "vote timing of the early termination");
};

// ES: exact early termination on (sufA / sufB non-NULL).  Two instantiations: the votes and the dead-wave path are a handful of
// instructions, but their presence in the loop cost the DENSE sweep 6 % at d = 256 (register allocation of the prefetched loop).
template <int D, int HEAD, bool BF, bool ES, int GM = 0>
__global__ void __launch_bounds__((64 * Geo4<D, GM>::WAVES)) sweep4_kernel(Args4 g) {
    using G = Geo4<D, GM>;
    constexpr int kLoaders = G::LOADERS, kMPR = G::MPR, MW = G::MW, RPW = G::RPW, RPR = G::RPR;
    constexpr int NB = G::NB, ROWS = G::ROWS, UT = G::UT, RB = G::RB, HB = G::HB, BB = G::BB, NP = G::NP, UA = G::UA, NSLOT = G::NSLOT;
    constexpr bool GL = G::GL;
    constexpr int CAPL = G::CAP;                       // list slots per user row
    constexpr int RINGL = G::RING;                     // entries per candidate ring
    constexpr int NM = D / 16;
    constexpr float kEps = BF ? 6.103515625e-5f : 7.890625e-3f;   // 2^-14  |  2^-7 * 1.01 (pda_score_topk_v3.hip: the unit roundoff of bf16 is 2^-8)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* tiles = smem;                                                             // NSLOT x BB
    uint64_t* lists = GL ? g.lists_ws + (size_t)blockIdx.x * UT * CAPL                      // [UT][CAPL] exact keys
                         : reinterpret_cast<uint64_t*>(smem + G::lds_tiles);
    int* cntl = reinterpret_cast<int*>(smem + G::lds_tiles + G::lds_lists);                  // [UT]
    float* taul = reinterpret_cast<float*>(cntl + UT);                                       // [UT] exact K-th value (-inf until K entries)
    unsigned* rings = reinterpret_cast<unsigned*>(taul + UT);                                // [8][RINGL]
    unsigned* sync = rings + kMainWaves * RINGL;
    unsigned* s_landed = sync;                // [4]
    unsigned* s_stop = sync + 108;            // [1]  early termination: loaders leave
    unsigned* s_released = sync + 4;          // [8]
    unsigned* s_tail = sync + 12;             // [8]  ring write positions (ring RPW w + rg of MFMA wave w)
    unsigned* s_head = sync + 20;             // [8]  ring read positions
    unsigned* s_done = sync + 28;             // [8]  MFMA wave w has pushed its last candidate
    unsigned* s_tver = sync + 36;             // [8]  bumped by the rescoring wave whenever a threshold of wave w's rows rose
    unsigned* s_vote = sync + 44;             // [8][8]  early termination votes about tile it & 7
    unsigned* s_uns = sync + 112;             // [16]  one bit per user row: its list came in unsorted and has not been compacted yet

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x % g.n_splits, utile = blockIdx.x / g.n_splits;
    if (g.n_users_dev != nullptr && utile * UT >= *g.n_users_dev) return;
    const int K = g.K;
    const int nt = split_tiles(g.n_tiles, split, g.n_splits);
    const int wt = warm_tiles_of(g, split);                        // the split's tiles the warm-up has scored
    const bool starts_empty = g.lists_empty || (g.warm_shared && split > 0);   // shared warm-up: its lists went to split 0, this split has the seed only
    const int n_it = max(0, nt - wt);                              // 64-item tiles of the pre-filtered loop: local index i <-> tile split + (wt + i) S
    const int n_blk = NB > 2 ? (2 * n_it) / NB : n_it * (2 / NB);                           // blocks: block b = half-tiles NB b .. NB b + NB - 1 of that sequence
    // block row of sweep row rb (rb < n_users_blk): the users of an early-terminating sweep are regrouped (stop_predict4_kernel)
    auto orig_row = [&](int rb) __attribute__((always_inline)) -> int { if constexpr (!ES) return rb; else return (g.row_perm != nullptr && rb < g.n_users_blk) ? g.row_perm[rb] : rb; };
#ifdef PDA_V4_STAMP      /* debug build: phase clocks (100 MHz) of a few workgroups, printed */
    const unsigned long long st0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0;
#define PDA_STAMP(x) x = __builtin_amdgcn_s_memrealtime()
#else
#define PDA_STAMP(x)
#endif
    if (tid < 128) sync[tid] = tid >= 112 ? 0xFFFFFFFFu : 0u;       // (words 112 .. 127: every list comes in unsorted, see s_uns)
    // kernel identity (workspace + 16; tests read it back to prove WHICH kernel and geometry a call ran):
    // generation 4 | geometry << 8 | early-terminating << 12 | head << 13 | bf16 tables << 14 | d / 64
    if (tid == 0 && blockIdx.x == 0)
        g.stats[4] = (4u << 28) | ((unsigned)GM << 8) | ((ES ? 1u : 0u) << 12) | ((unsigned)HEAD << 13) | ((BF ? 1u : 0u) << 14) | (unsigned)(D >> 6);
    // The lists of the warm-up -> LDS (or the workspace), their counts and K-th values: ALL waves share the rows (the MFMA
    // waves wait for the thresholds behind the barrier: 64 rows per rescoring wave, one after the other, were 50 us of every
    // workgroup's life -- 13 % of an early-terminating sweep).  A list may come in unsorted: its K-th value is the smallest key.
    {
        // (eight rows of a wave in flight: one row after the other was 16 x the latency of a load -- two with the permutation)
        constexpr int NW = G::WAVES, PB = 8;
        for (int r0 = wave; r0 < UT; r0 += PB * NW) {
          int rov[PB];
          uint64_t keyv[PB];
#pragma unroll
          for (int q = 0; q < PB; ++q) {
              const int rb = utile * UT + r0 + q * NW;
              rov[q] = (r0 + q * NW < UT && rb < g.n_users_blk) ? orig_row(rb) : -1;
          }
#pragma unroll
          for (int q = 0; q < PB; ++q)
              keyv[q] = (rov[q] < 0 || starts_empty) ? 0ull
                        : g.handover != nullptr ? (lane < kCap4 ? g.handover[((size_t)split * g.n_users_blk + rov[q]) * kCap4 + lane] : 0ull)
                        : (lane < K ? g.out_keys[((size_t)split * g.n_users_blk + rov[q]) * K + lane] : 0ull);
#pragma unroll
          for (int q = 0; q < PB; ++q) {
            const int rr = r0 + q * NW;
            if (rr >= UT) break;
            const int rb = utile * UT + rr;
            const uint64_t key = keyv[q];
            const int c = __popcll(__ballot(key != 0ull));
            if (lane < (g.handover != nullptr ? kCap4 : K)) lists[(size_t)rr * CAPL + lane] = key;
            uint32_t mn = key != 0ull ? (uint32_t)(key >> 32) : 0xFFFFFFFFu;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
            if (lane == 0) {
                cntl[rr] = c;
                taul[rr] = rb < g.n_users_blk ? (c >= K ? pda_unordf(mn) : -INFINITY) : INFINITY;
            }
          }
        }
        if constexpr (GL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
#ifdef PDA_V4_PROF
    unsigned long long prof[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    PDA_STAMP(st1);

    if (wave >= MW + kLoaders) {
        // ============================== rescoring wave ==============================
        constexpr int RR = kMPR * RPR;                       // rows of this wave (128: two per lane)
        const int r = wave - MW - kLoaders;
        const int row0 = r * RR;
        uint64_t* my_lists = lists + (size_t)row0 * CAPL;
        constexpr int NRL = (RR + 63) / 64;                  // rows per lane
        int uidv[NRL], rblv[NRL];                            // user id and place in the caller's block (-1: padding) of the lane's rows
        float seedv[NRL];
        long long hbv[NRL], hev[NRL];
#pragma unroll
        for (int s2 = 0; s2 < NRL; ++s2) { uidv[s2] = 0; rblv[s2] = -1; seedv[s2] = -INFINITY; hbv[s2] = 0; hev[s2] = 0; }
        // row `row` of this wave: entry row / 64 of lane row % 64
        auto row_i = [&](const int (&a)[NRL], int row) __attribute__((always_inline)) -> int {
            int v = __shfl(a[0], row & 63, 64);
#pragma unroll
            for (int s2 = 1; s2 < NRL; ++s2) { const int t = __shfl(a[s2], row & 63, 64); v = (row >> 6) == s2 ? t : v; }
            return v;
        };
        auto row_f = [&](const float (&a)[NRL], int row) __attribute__((always_inline)) -> float {
            float v = __shfl(a[0], row & 63, 64);
#pragma unroll
            for (int s2 = 1; s2 < NRL; ++s2) { const float t = __shfl(a[s2], row & 63, 64); v = (row >> 6) == s2 ? t : v; }
            return v;
        };
        auto row_l = [&](const long long (&a)[NRL], int row) __attribute__((always_inline)) -> long long {
            long long v = __shfl(a[0], row & 63, 64);
#pragma unroll
            for (int s2 = 1; s2 < NRL; ++s2) { const long long t = __shfl(a[s2], row & 63, 64); v = (row >> 6) == s2 ? t : v; }
            return v;
        };
        const bool hist_on = g.hist_indptr != nullptr;
        const bool bloom_on = g.bloom != nullptr && !(HEAD == PDA_HEAD_POP && g.prep_hdr[2] != 0);
#pragma unroll
        for (int s2 = 0; s2 < NRL; ++s2) {
            const int rl = 64 * s2 + lane;
            const int rb_s = utile * UT + row0 + rl;
            const bool ok = rl < RR && rb_s < g.n_users_blk;
            const int rb_l = ok ? orig_row(rb_s) : 0;
            uidv[s2] = ok ? g.users[rb_l] : 0;
            rblv[s2] = ok ? rb_l : -1;
            if (g.seed != nullptr && ok) seedv[s2] = g.seed[rb_l];
            if (hist_on && ok) {
                const int64_t hr = g.hist_row_mode == PDA_HIST_BY_USER_ID ? (int64_t)uidv[s2] : (int64_t)rb_l;
                hbv[s2] = g.hist_indptr[hr];
                hev[s2] = g.hist_indptr[hr + 1];
            }
        }
        if constexpr (GL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        unsigned head[kMPR], n_cand = 0;
#pragma unroll
        for (int z = 0; z < kMPR; ++z) head[z] = 0;
        constexpr int LPC = D / 32;                 // lanes per candidate
        constexpr bool kRowsInterleaved = D <= 128 && PDA_V4_ROWS_INTERLEAVED != 0;
        constexpr int CPP = 64 / LPC;               // candidates per pass
        const int q = lane % LPC, ci = lane / LPC;
        int sel = 0;                                // the ring looked at last
        int next_sort = 0;                          // rows [0, next_sort) have been sorted in idle time
        unsigned idle = 0;
        PROF_T0(tr0);
        // One pass = up to CPP candidates of ONE ring: their user and item rows are loaded (a round trip to L2 / HBM), the exact dot
        // products taken, the survivors checked against the history and appended to their lists.  The rows of pass p + 1 are
        // requested between the dot products and the appends of pass p -- the row registers are dead by then.  Cycle counters on
        // the raw head (C3, ~330 candidates per user, the rescoring waves 98 % busy): of a pass's 8 000 cycles 3 400 were the wait
        // for its rows, 4 300 the history check and the appends.
        f32x4 uu[8], ii[8];
        bool have = false;                          // a pass is in flight: its words read, its rows requested
        int f_ring = 0;                             // (per lane: the ring its candidate came from)
        unsigned f_word = 0u, f_bh1 = 0u, f_bh2 = 0u, f_taken = 0u;
        uint32_t f_bw1 = 0xFFFFFFFFu, f_bw2 = 0xFFFFFFFFu;
        float f_pv = 1.0f;
        // A pass is filled from ALL rings of the wave (served one ring at a time, passes carried 10.8 of 16 candidates on the raw
        // head, and a pass costs the same round trips whatever it carries).  What a pass costs IS its LDS round trips -- ~450 cycles
        // each beside eight MFMA waves streaming B fragments, ten of them per pass -- so the ring words are read WITH the tails:
        // the rings of a wave share the pass's slots in pairs, one ring filling its group of GS slots from the bottom, the other
        // from the top, and every lane knows where its word would be before it knows whether there is one.  The row's threshold
        // and seed come with the user id (one more round trip); the threshold may be one pass old -- a candidate let in against
        // an old threshold is dropped by the next compaction of its row.
        constexpr int NG = kMPR >= 2 ? kMPR / 2 : 1, GS = CPP / NG;
        static_assert(kMPR == 1 || kMPR % 2 == 0, "rings share the slots of a pass in pairs");
        bool f_valid = false;
        float f_tau = -INFINITY, f_sd = -INFINITY;
        auto fetch = [&](unsigned* dn_out) __attribute__((always_inline)) -> bool {
            const int gi = ci / GS, li = ci % GS;
            const int yA = kMPR >= 2 ? 2 * gi : 0, yB = kMPR >= 2 ? 2 * gi + 1 : 0;
            unsigned hA = 0, hB = 0;
#pragma unroll
            for (int y = 0; y < kMPR; ++y) { hA = yA == y ? head[y] : hA; hB = yB == y ? head[y] : hB; }
            unsigned tl[kMPR];
            if (dn_out != nullptr) {
                unsigned dn = 1u;
#pragma unroll
                for (int z = 0; z < kMPR; ++z) dn &= lds_ld(&s_done[(kMPR * r + z) / RPW]);         // read BEFORE the tails
                *dn_out = dn;
            }
#pragma unroll
            for (int z = 0; z < kMPR; ++z) tl[z] = lds_ld(&s_tail[kMPR * r + z]);
            PDA_CBAR();                                                                            // (the words BEHIND the tails)
            const unsigned wA = rings[(kMPR * r + yA) * RINGL + (hA + (unsigned)li) % RINGL];
            const unsigned wB = kMPR >= 2 ? rings[(kMPR * r + yB) * RINGL + (hB + (unsigned)(GS - 1 - li)) % RINGL] : 0u;
            unsigned take[kMPR], taken = 0u, total = 0u;
            if constexpr (kMPR == 1) {
                take[0] = (unsigned)min((int)(tl[0] - head[0]), CPP);
            } else {
#pragma unroll
                for (int gq = 0; gq < NG; ++gq) {
                    const int avA = (int)(tl[2 * gq] - head[2 * gq]), avB = (int)(tl[2 * gq + 1] - head[2 * gq + 1]);
                    const bool prioA = ((sel ^ gq) & 1) == 0;                                      // the pair's first claim alternates
                    const int tA = prioA ? min(avA, GS) : min(avA, GS - min(avB, GS));
                    const int tB = prioA ? min(avB, GS - tA) : min(avB, GS);
                    take[2 * gq] = (unsigned)tA;
                    take[2 * gq + 1] = (unsigned)tB;
                }
            }
            unsigned tAl = 0, tBl = 0;
#pragma unroll
            for (int y = 0; y < kMPR; ++y) {
                tAl = yA == y ? take[y] : tAl;
                tBl = yB == y ? take[y] : tBl;
                total += take[y];
                taken |= take[y] != 0u ? 1u << y : 0u;
            }
            if (total == 0u) return false;
            sel ^= 1;
#pragma unroll
            for (int y = 0; y < kMPR; ++y) head[y] += take[y];
            n_cand += total;
            const bool fromA = (unsigned)li < tAl;
            const bool valid = fromA || (kMPR >= 2 && (unsigned)(GS - 1 - li) < tBl);
            const int my_ring = fromA ? yA : yB;
            const unsigned word = valid ? (fromA ? wA : wB) : 0u;
            const int row = my_ring * RPR + (int)(word >> 26);                      // row of this wave
            const int loc = (int)(word & 0x3FFFFFFu);                               // local item id
            const int urow = row_i(uidv, row);
            const int rbb = row_i(rblv, row);                                       // (its place in the caller's block: the filter words)
            f_sd = row_f(seedv, row);
            f_tau = taul[row0 + row];
            // d <= 128: the 8-float chunks of a row go round the candidate's LPC lanes (lane q: chunks q, LPC + q, 2 LPC + q, ..), so
            // that a load touches ONE cache line per candidate.  With 32 consecutive floats per lane every load touched 64
            // different lines, 1 024 tag look-ups per pass: the CU's vector memory pipe was what capped rescoring at ~5 candidates
            // per 1 000 cycles, whether four or eight waves were at it (cycle counters, raw head).
            const size_t ub = (size_t)urow * D + (kRowsInterleaved ? q * 8 : q * 32), ib = (size_t)loc * D + (kRowsInterleaved ? q * 8 : q * 32);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int off = kRowsInterleaved ? 8 * LPC * (c >> 1) + 4 * (c & 1) : 4 * c;
                uu[c] = pda_load4<BF>(g.U, ub + off);
                ii[c] = pda_load4<BF>(g.I, ib + off);
            }
            f_pv = 1.0f;
            if constexpr (HEAD == PDA_HEAD_POP) f_pv = g.pop[loc];
            // the two filter words of (row, item), in flight with the rows
            f_bh1 = 0u; f_bh2 = 0u; f_bw1 = 0xFFFFFFFFu; f_bw2 = 0xFFFFFFFFu;
            if (bloom_on && valid && q == LPC - 1) {
                if (rbb >= 0) {
                    f_bh1 = bloom_h1(g.item_offset + loc);
                    f_bh2 = bloom_h2(g.item_offset + loc);
                    f_bw1 = g.bloom[(size_t)rbb * 32 + (f_bh1 >> 5)];
                    f_bw2 = g.bloom[(size_t)rbb * 32 + (f_bh2 >> 5)];
                }
            }
            PDA_CBAR();
#pragma unroll
            for (int y = 0; y < kMPR; ++y)
                if (taken & (1u << y)) lds_st(&s_head[kMPR * r + y], head[y]);      // the words are in registers: the slots are free
            f_ring = my_ring;
            f_taken = taken;
            f_valid = valid;
            f_word = word;
            have = true;
            return true;
        };
        for (;;) {
            if constexpr ((PDA_V4_ABL & 8) != 0) break;
            PROF_T0(tp0);
            if (!have) {
                unsigned dn = 1u;
                if (!fetch(&dn)) {
                    if (dn) break;
                    if (next_sort < RR) {
                        // nothing to rescore: put one more of the lists that came in unsorted in order (it has to be sorted for the
                        // output anyway; done here it costs nothing, done behind the sweep it is a serial tail of the launch)
                        const int rr = next_sort++;
                        compact_list<CAPL, GL>(my_lists + (size_t)rr * CAPL, &cntl[row0 + rr], &taul[row0 + rr], K, lane, &s_uns[(row0 + rr) >> 5],
                                                1u << ((row0 + rr) & 31));
                        continue;
                    }
                    if (++idle > kSpinMax) { if (lane == 0) g.stats[0] = 3u; break; }
                    PROF_T0(ti);
                    __builtin_amdgcn_s_sleep(PDA_V4_RSLEEP);
                    PROF_T1(ti, 7);
                    continue;
                }
            }
            idle = 0;
            PROF_INC(8, 1);
            // ---- the pass in flight
            const unsigned c_taken = f_taken;
            const bool valid = f_valid;
            const float c_tau = f_tau, c_sd = f_sd;
            const int row = f_ring * RPR + (int)(f_word >> 26);
            const int loc = (int)(f_word & 0x3FFFFFFu);
            const float pv = f_pv;
            const unsigned bh1 = f_bh1, bh2 = f_bh2;
            const uint32_t bw1 = f_bw1, bw2 = f_bw2;
#ifdef PDA_V4_PROF
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            PROF_T1(tp0, 16);
            PROF_T0(tp1);
            float o0 = 0.f, o1 = 0.f;
            if constexpr (kRowsInterleaved) {
                // The sum order is the oracle's (oracle/pda_oracle.c dot_chain): chunk x feeds chain x & 1, the chunks of a chain in
                // ascending order.  Lane q holds chunks LPC cq + q in its registers 2 cq, 2 cq + 1.  d = 64: lane 0 IS chain 0, lane 1
                // chain 1.  d = 128: chain 0 alternates between lanes 0 and 2, chain 1 between lanes 1 and 3 -- eight FMAs, one DPP
                // move across the quad, eight FMAs, one move back, per four chunks.
                auto fma8 = [&](float acc, int cq) __attribute__((always_inline)) -> float {
#pragma unroll
                    for (int sidx = 0; sidx < 4; ++sidx) {
                        acc = __builtin_fmaf(uu[2 * cq][sidx], ii[2 * cq][sidx], acc);
                        acc = __builtin_fmaf(uu[2 * cq + 1][sidx], ii[2 * cq + 1][sidx], acc);
                    }
                    return acc;
                };
                float o = 0.f;
                if constexpr (LPC == 4) {
                    const bool hi = q >= 2;
                    auto swap2 = [](float x) __attribute__((always_inline)) -> float {
                        return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
                    };
#pragma unroll
                    for (int cq = 0; cq < 4; ++cq) {
                        const float a = fma8(o, cq);                  // (lanes 0, 1: the chains behind chunks 4 cq, 4 cq + 1)
                        const float a_sw = swap2(a);
                        const float b = fma8(hi ? a_sw : o, cq);      // (lanes 2, 3: behind chunks 4 cq + 2, 4 cq + 3)
                        const float b_sw = swap2(b);
                        o = hi ? b : b_sw;
                    }
                } else {
                    static_assert(LPC == 2, "d = 64");
#pragma unroll
                    for (int cq = 0; cq < 4; ++cq) o = fma8(o, cq);
                }
                // even lanes end with chain 0, odd lanes with chain 1
                o0 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(o), 0xB1, 0xF, 0xF, true));            // quad_perm [1,0,3,2]
                o1 = o;
            } else {
            float c0 = 0.f, c1 = 0.f;
#pragma unroll
            for (int ph = 0; ph < LPC; ++ph) {                o0 = c0;
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
            }
            float sc = o0 + o1;                               // meaningful on the candidate's last lane
            if constexpr (HEAD == PDA_HEAD_POP) sc = (sc > 0.0f ? sc + 1.0f : __expf(sc)) * pv;
            float tt = (valid && q == LPC - 1) ? sc : -INFINITY;
            have = false;
            PROF_T1(tp1, 17);
#if PDA_V4_RESCORE_AHEAD
            PROF_T0(tp4);
            {
                // ---- the next pass: its rows travel while this one's survivors go through the history and the lists
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+v"(tt));                  // (the dot products first: the row registers are free only behind them)
                __builtin_amdgcn_sched_barrier(0);
#endif
                fetch(nullptr);
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            PROF_T1(tp4, 21);
#endif
            PROF_T0(tp2);
            const int lrow = row0 + row;
            const int item = g.item_offset + loc;
            // ">=": equal scores are decided by the key (lower item id wins) at the next compaction, so ties must get in.
            // With a seed (item-sharded evaluation: the K-th values of the OTHER shards' warm-up lists) nothing below it can be
            // in the merged top K: it stays out of this shard's list, which may then end shorter than K.
            bool p = valid && q == LPC - 1 && (tt >= fmaxf(c_tau, c_sd));
            if (hist_on) {
                // train items are masked HERE: a binary search in the row's id-sorted history for a candidate that has passed
                // the pre-filter and the exact threshold -- and whose two Bloom bits are set (hist_bloom4_kernel)
                const bool look = p && (((bw1 >> (bh1 & 31u)) & (bw2 >> (bh2 & 31u)) & 1u) != 0u);
                if (__any(look)) {                         // (the bounds come through shuffles: every lane takes part)
                    long long lo = row_l(hbv, row), hi = row_l(hev, row);
                    const long long he = hi;
                    if (look) {
                        while (lo < hi) {
                            const long long mid = (lo + hi) >> 1;
                            if (g.hist_indices[mid] < item) lo = mid + 1; else hi = mid;
                        }
                        if (lo < he && g.hist_indices[lo] == item) p = false;
                    }
                }
            }
            const uint64_t key = pda_pack_key(tt, (uint32_t)item);
            PROF_T1(tp2, 18);
            PROF_T0(tp3);
            PROF_INC(20, __popcll(__ballot(p)));
#ifdef PDA_V4_PROF
            const bool changed = append_keys<CAPL, GL>(p, lrow, tt, key, lists, cntl, taul, row0, RR, K, lane, s_uns, prof);
#else
            const bool changed = append_keys<CAPL, GL>(p, lrow, tt, key, lists, cntl, taul, row0, RR, K, lane, s_uns);
#endif
            PROF_T1(tp3, 19);
            if (changed && lane < kMPR && ((c_taken >> lane) & 1u) != 0u)
                __hip_atomic_fetch_add(&s_tver[(kMPR * r + lane) / RPW], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        PROF_T1(tr0, 6);
        PROF_INC(9, n_cand);
        PROF_FLUSH(6, 9);
        PROF_FLUSH(16, 23);
        if (lane == 0) atomicAdd(g.stats + 1, n_cand);
    } else if (wave >= MW) {
        // ================================== loader ==================================
        // Issued through inline asm: hipcc counts a __builtin_amdgcn_global_load_lds as a pending LDS write and puts
        // s_waitcnt vmcnt(0) in front of the next ds_read of ANY address (the hand-over polls): the loads would be
        // synchronous.  M0 = LDS byte address of the piece, saved and restored inside the statement.
        const int l = wave - MW;
        const unsigned lds_tiles0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)tiles;
        __syncthreads();
        // Three or more slots: a loader keeps TWO blocks in flight -- it announces block b - 1 behind the issue of its pieces
        // of block b (a counted vmcnt), so a block is delivered per issue time, not per issue time + latency (2 500 cycles
        // against the 1 600 of a block's MFMAs).  Two slots: nothing to issue ahead; wait for the block and say so at once.
        constexpr bool PIPE = NSLOT >= 3;
        constexpr int MYP = (NP + kLoaders - 1) / kLoaders;       // pieces per loader and block (the last loader may have one less)
        const int my_pieces = (NP - l + kLoaders - 1) / kLoaders;
        bool stop = false;
        for (int b = 0; b < n_blk && !stop; ++b) {
            PROF_T0(tl0);
            if (b >= NSLOT && !(PDA_V4_ABL & 2)) {                 // slot b % NSLOT is free once every MFMA wave has released block b - NSLOT
                const unsigned want = (unsigned)(b - NSLOT + 1);
                unsigned spin = 0;
                auto min_released = [&]() __attribute__((always_inline)) -> unsigned {
                    unsigned mn = 0xFFFFFFFFu;
#pragma unroll
                    for (int z = 0; z < MW; ++z) mn = min(mn, lds_ld(&s_released[z]));
                    return mn;
                };
                while (min_released() < want) {
                    if (lds_ld(s_stop)) { stop = true; break; }
                    if (++spin > kSpinMax) { if (lane == 0) g.stats[0] = 4u; stop = true; break; }
                    __builtin_amdgcn_s_sleep(PDA_V4_LSLEEP);
                }
                if (stop) break;
            }
            PROF_T1(tl0, 10);
            PROF_T0(tl1);
            const int hf0 = b * NB;                                // first half-tile of the block
            const int t = split + (wt + (hf0 >> 1)) * g.n_splits;
            [[maybe_unused]] const unsigned char* src = g.rows + (size_t)t * G::TB + (size_t)(hf0 & 1) * HB + lane * 16;
            [[maybe_unused]] const unsigned dst = lds_tiles0 + (unsigned)((b % NSLOT) * BB);
#pragma unroll
            for (int c = 0; c < MYP; ++c) {
                const int piece = l + kLoaders * c;
                if (piece < NP && (piece < NP - 1 || lane < G::LASTL) && !(PDA_V4_ABL & 4)) {
#if defined(__HIP_DEVICE_COMPILE__)
                    unsigned keep;
                    const unsigned char* gsrc = src + (size_t)piece * 1024;
                    const unsigned ldst = __builtin_amdgcn_readfirstlane(dst + (unsigned)piece * 1024u);
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(gsrc), "s"(ldst) : "memory");
#endif
                }
            }
            PROF_T1(tl1, 11);
            PROF_T0(tl2);
            if constexpr (PIPE) {
                if (b >= 1) {
                    if (my_pieces == MYP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MYP) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MYP > 0 ? MYP - 1 : 0) : "memory");
                    lds_st(&s_landed[l], (unsigned)b);              // blocks 0 .. b - 1
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_st(&s_landed[l], (unsigned)(b + 1));            // blocks 0 .. b
            }
            PROF_T1(tl2, 12);
        }
        if constexpr (PIPE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!stop) lds_st(&s_landed[l], (unsigned)n_blk);
        }
        PROF_FLUSH(10, 12);
    } else {
    // ================================== MFMA wave ==================================
    const int w = wave;
    const int j = lane & 31, h = lane >> 5;
    const int row0 = w * ROWS;
    // A operand: row j of the wave (k = 16 m + 8 h .. + 7) rounded to bf16 and NEGATED (all of the A side, the extra k-step
    // too): the accumulators hold -(s~ - thr/pop + 1 + eps), and "candidate" is "negative", i.e. the sign bit.
    u32x4 ah[UA][NM];
    float nu_row[UA];
#pragma unroll
    for (int u = 0; u < UA; ++u) {
        const int rb = utile * UT + row0 + 32 * u + j;
        const bool ok = rb < g.n_users_blk;
        const int uid = ok ? g.users[orig_row(rb)] : 0;
        float ss = 0.f;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = x;
            if (ok) {
                x = pda_load4<BF>(g.U, (size_t)uid * D + 8 * h + 16 * m);
                y = pda_load4<BF>(g.U, (size_t)uid * D + 8 * h + 16 * m + 4);
            }
            u32x4 lo_unused;
            split8(x, y, ah[u][m], lo_unused);
#pragma unroll
            for (int k = 0; k < 4; ++k) ah[u][m][k] ^= 0x80008000u;
#pragma unroll
            for (int k = 0; k < 4; ++k) ss += x[k] * x[k] + y[k] * y[k];
        }
        ss += __shfl_xor(ss, 32, 64);
        nu_row[u] = sqrtf(ss) * 1.0009765625f * 1.0001f;           // padded ||u||
    }
    PDA_STAMP(st2);
    __syncthreads();                       // lists, thresholds and hand-over words are initialised
    PDA_STAMP(st3);

    // threshold of the lane's own row (finite: +-1e30 stand for +-inf), lowered by 2^-16 relative (the rounding of the
    // extra k-step, pda_score_topk_v3.hip), as the A operand of the extra k-step:
    //   k 0..7 (lanes < 32): (t1,t1,t2,t2,t1,t3,t2,t3), thr = t1 + t2 + t3 exactly;  k 8..10: -1, -1, -eps scale of the row
    //   (negated like the rest of the A side: v3 carries the opposite signs)
    float thr_own[UA], thr_min[UA];
    u32x4 aex[UA];
    float seed_own[UA];                         // external lower bound of the row's final K-th value (see the rescoring wave)
#pragma unroll
    for (int u = 0; u < UA; ++u) {
        const int rb = utile * UT + row0 + 32 * u + j;
        seed_own[u] = (g.seed != nullptr && rb < g.n_users_blk) ? g.seed[orig_row(rb)] : -INFINITY;
    }
    auto refresh_thr = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const float tq = fmaxf(taul[row0 + 32 * u + j], seed_own[u]);
            float tf = (tq == INFINITY || tq == -INFINITY) ? tq : tq - fabsf(tq) * 1.52587890625e-5f - 1e-30f;
            tf = fminf(fmaxf(tf, -1.0e30f), 1.0e30f);
            thr_own[u] = tf;
            float mn = tf;
            uint32_t t1, t2, t3;
            bf16_split3(tf, t1, t2, t3);
            const uint32_t nnu = bf16_up(nu_row[u] * (kEps * 1.001f * 1.08f)) | 0x8000u;
            aex[u][0] = h ? 0xBF80BF80u : (t1 | (t1 << 16));
            aex[u][1] = h ? nnu : (t2 | (t2 << 16));
            aex[u][2] = h ? 0u : (t1 | (t3 << 16));
            aex[u][3] = h ? 0u : (t2 | (t3 << 16));
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o, 64));
            thr_min[u] = mn;
        }
    };
    // the exact threshold of accumulator register r (lane half hv) of row set u, strictly below tau (ties must pass)
    auto thr_of = [&](int r, int hv, int u) __attribute__((always_inline)) -> float {
        const int rw = (r & 3) + 8 * (r >> 2) + 4 * hv;
        const float tq = fmaxf(taul[row0 + 32 * u + rw], __shfl(seed_own[u], rw, 64));
        return (tq == INFINITY || tq == -INFINITY) ? tq : tq - fabsf(tq) * 9.5367431640625e-7f - 1e-30f;
    };
    refresh_thr();
    unsigned tver_seen = 0, landed_c = 0;
#ifdef PDA_V4_PRIO
    if (w < 4) __builtin_amdgcn_s_setprio(PDA_V4_PRIO);     // experiment: the first-dispatched MFMA wave of every SIMD goes first
#endif

    // ring RPW w + rg of this wave takes the candidates of rows rg RPR .. rg RPR + RPR - 1
    unsigned tail[RPW], head_c[RPW];        // wave-uniform
#pragma unroll
    for (int rg = 0; rg < RPW; ++rg) { tail[rg] = 0u; head_c[rg] = 0u; }
    auto publish_tails = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int rg = 0; rg < RPW; ++rg) lds_st(&s_tail[RPW * w + rg], tail[rg]);
    };
    // push the flagged registers (bit 15 - r <-> register r; registers 0 .. 7 are rows 0 .. 15 of the row set, 8 .. 15 rows 16 .. 31)
    auto push_mask = [&](uint32_t m_all, int loc, int u) __attribute__((always_inline)) {
#pragma unroll
        for (int rg = 0; rg < RPW; ++rg) {
            uint32_t m = RPW == 1 ? m_all : (rg == 0 ? (m_all & 0xFF00u) : (m_all & 0x00FFu));
            unsigned* ring = rings + (RPW * w + rg) * RINGL;
            while (__any(m != 0)) {
                const bool act = m != 0;
                const int bit = 31 - __builtin_clz(m | 1u);
                const int r = 15 - bit;
                const int row = 32 * u + (r & 3) + 8 * (r >> 2) + 4 * h - rg * RPR;
                m &= ~(1u << bit);
                const uint64_t pm = __ballot(act);
                if (tail[rg] + 64u - head_c[rg] > (unsigned)RINGL) {      // ring full: publish what is there and wait for the rescoring wave
                    PDA_CBAR();
                    lds_st(&s_tail[RPW * w + rg], tail[rg]);
                    unsigned spin = 0;
                    PROF_T0(tq);
                    do {
                        head_c[rg] = lds_ld(&s_head[RPW * w + rg]);
                        if (++spin > kSpinMax) { if (lane == 0) g.stats[0] = 2u; break; }
                    } while (tail[rg] + 64u - head_c[rg] > (unsigned)RINGL);
                    PROF_T1(tq, 2);
                }
                const unsigned slot = (tail[rg] + (unsigned)__builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0))) % RINGL;
                if (act) ring[slot] = ((uint32_t)row << 26) | (uint32_t)loc;
                tail[rg] += (unsigned)__popcll(pm);
            }
        }
    };
    auto ensure_landed = [&](int b) __attribute__((always_inline)) {
        const unsigned want = (unsigned)(b + 1);
        if (landed_c < want && !(PDA_V4_ABL & 2)) {
            unsigned spin = 0;
            PROF_T0(te);
            do {
                landed_c = 0xFFFFFFFFu;
#pragma unroll
                for (int z = 0; z < kLoaders; ++z) landed_c = min(landed_c, lds_ld(&s_landed[z]));
                if (++spin > kSpinMax) { if (lane == 0) g.stats[0] = 1u; break; }
            } while (landed_c < want);
            PROF_T1(te, 1);
            PDA_CBAR();
        }
    };
    const unsigned char* lane_base = tiles + j * RB + 16 * h;       // B fragment (cb, m) of slot s: + s BB + cb HB + 32 m
    [[maybe_unused]] const unsigned lane_base_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)tiles + (unsigned)(j * RB + 16 * h);
    // (raw head: a prep built WITH a popularity carries 1/pop pieces in its test operands -- header word 1)
    [[maybe_unused]] const bool raw_on_pop_prep = HEAD == PDA_HEAD_RAW && g.prep_hdr[1] != 0;

#ifndef PDA_V4_PF
#define PDA_V4_PF 4       // B fragments in flight per MFMA wave (the compiler keeps fewer when registers are short)
#endif
#ifndef PDA_V4_STAGGER
#define PDA_V4_STAGGER 0
#endif
    // The two MFMA waves of a SIMD (w and w + 4) half a block apart: while one waits for its last MFMA and tests the
    // accumulators, the other one's MFMAs keep the pipe busy.
    if (PDA_V4_STAGGER > 0 && w >= 4) __builtin_amdgcn_s_sleep(PDA_V4_STAGGER);
    PROF_T0(tm0);
    int n_done = 0;
    bool stopped = false;
#ifndef PDA_V4_VOTE_EVERY
#define PDA_V4_VOTE_EVERY 2
#endif
    constexpr int kVoteEvery = PDA_V4_VOTE_EVERY;      // a vote in front of every kVoteEvery-th tile (1, 2 or 4)
    constexpr int kVL = (NSLOT + 2) / 2;               // a vote is about the tile kVL behind the one it is stored in front of
    float sa_nx = 0.0f, sb_nx = 0.0f;          // suffix bounds at tile it + kVL of the coming vote
    bool nx_ok = ES && kVoteEvery - 1 + kVL < n_it;
    if (nx_ok) {
        const int tn = split + (wt + kVoteEvery - 1 + kVL) * g.n_splits;
        sa_nx = g.sufA[tn];
        sb_nx = g.sufB[tn];
    }
    constexpr int S = NB * NM, PF = S < PDA_V4_PF ? S : PDA_V4_PF;
    constexpr bool kAsmGeo = PDA_V4_ASM != 0 && (D <= 128 || PDA_V4_ASM256 != 0) && UA == 1 && NB == 2;     // the block as one asm statement (below)
    constexpr bool PFX = !kAsmGeo && NSLOT >= 3 && S % PF == 0;          // prefetch across the block boundary
    u32x4 bq[PF];
    int dead_from = 0x7FFFFFFF;                // first tile from which no row of this wave can be reached (early termination)
    for (int b = 0; b < n_blk && !stopped; ++b) {
        const unsigned pr_tv = lds_ld(&s_tver[w]);      // read here, used at the end of the iteration
        // ---- early termination.  In front of the last block of tile i every wave votes "nothing at or behind tile i + kVL
        // can reach my rows" (candidates still in the ring can only raise thresholds) -- BEFORE it releases that block.  Once
        // the first block of tile i + kVL has landed, every wave has released the block NSLOT before it, which is not earlier
        // than the one the votes were stored in front of (2 kVL - 1 >= NSLOT): the votes are complete, every wave reads the
        // same eight words (while the MFMAs of that block run) and, if they all agree, leaves behind that block.  The words
        // of tile i are written again for tile i + 8: by then every wave has released block 2 i + 17 - NSLOT, far behind.
        const int it = (b * NB) >> 1;
        const bool tile_first = NB == 2 || (b & 1) == 0, tile_last = NB == 2 || (b & 1) == 1;
        if (ES && tile_last && (it & (kVoteEvery - 1)) == kVoteEvery - 1) {
            bool dead = nx_ok;
#pragma unroll
            for (int u = 0; u < UA; ++u) dead = dead && (__builtin_fmaf(nu_row[u], sb_nx, sa_nx) * 1.000002f < thr_own[u]);
            const bool alldead = __all(dead);
            if (alldead && dead_from > it + kVL) dead_from = it + kVL;          // (suffix bounds: dead for every later tile as well)
            if (lane == 0) lds_st(&s_vote[(it & 7) * kMainWaves + w], alldead ? 1u : 0u);
            // the bounds of the next vote: loaded a tile ahead
            const int inx = it + kVoteEvery + kVL;
            const int tn = split + (wt + min(inx, max(n_it - 1, 0))) * g.n_splits;
            sa_nx = g.sufA[tn];               // (no use before the next vote: the loads stay in flight over the block)
            sb_nx = g.sufB[tn];
            nx_ok = inx < n_it;
        }
        // A wave none of whose rows anything can reach any more stops scoring: it keeps in step with the others (landing, votes,
        // release) and leaves the matrix pipe of its SIMD to the wave that still has live rows -- the workgroup as a whole
        // goes on until every wave is dead, i.e. for its slowest user.
        if (ES && it >= dead_from) {
            ensure_landed(b);
            if (tile_first && it >= kVL && ((it - kVL) & (kVoteEvery - 1)) == kVoteEvery - 1) {
                const unsigned* v = &s_vote[((it - kVL) & 7) * kMainWaves];
                unsigned all = 1u;
#pragma unroll
                for (int z = 0; z < MW; ++z) all &= lds_ld(&v[z]);
                stopped = all != 0u;
            }
            PDA_CBAR();
            lds_st(&s_released[w], (unsigned)(b + 1));
            PDA_CBAR();
            if (!(NB == 1 && (b & 1) == 0)) ++n_done;
            continue;
        }
        // With three slots the first B fragments of block b + 1 are read while the MFMAs of block b are still being issued: the
        // wave comes back from the accumulator test of block b with its operands in registers.
        const bool has_next = (PFX || kAsmGeo) && b + 1 < n_blk;
        if (b == 0 || !PFX) ensure_landed(b);       // (asm path: a poll only when the answer read under the previous block said "not yet")
        // "has block b + 1 landed?" is asked here and looked at half a block later, in front of the first read of block b + 1:
        // a synchronous poll costs an LDS round trip per block (440 cycles of a block's 1 600 on a busy LDS)
        unsigned lnd[kLoaders];
        const bool asked = has_next && landed_c < (unsigned)min(b + 2, n_blk);
        if (asked) {
#pragma unroll
            for (int z = 0; z < kLoaders; ++z) lnd[z] = lds_ld(&s_landed[z]);
        } else {
#pragma unroll
            for (int z = 0; z < kLoaders; ++z) lnd[z] = 0xFFFFFFFFu;
        }
        const unsigned char* tb = lane_base + (b % NSLOT) * BB;
        [[maybe_unused]] const unsigned char* tbn = lane_base + ((b + 1) % NSLOT) * BB;
        // ---- the block: NB UA chains, k-step major; S = NB NM B reads, PF in flight ----
        f32x16 acc[UA][NB];
        float popv[NB];
        int locv[NB];
        // d <= 128: the block as one inline-asm statement (pda_v4_block_asm.h): BlockAsm<D>::kFragmentsInFlight B reads ahead of
        // their MFMAs, counted waits.  Left to hipcc, every read sat one or two instructions in front of its MFMA whatever depth
        // the source asked for -- an LDS round trip per pair of MFMAs, ~1 740 cycles per block and wave for 576 cycles of MFMAs,
        // the pipe 66 % busy with two such waves per SIMD (what PMC measured in round 2).  The raw head on a prep that carries
        // popularity pieces (never built by pda_amd.ops) rewrites the test operands and keeps the compiler's schedule.
        constexpr bool kAsmBlock = kAsmGeo;
        bool asm_done = false;
        if constexpr (kAsmBlock) {
            if (HEAD == PDA_HEAD_POP || !raw_on_pop_prep) {
                const unsigned a0 = lane_base_lds + (unsigned)((b % NSLOT) * BB);
                u32x4 piq;
                BlockAsm<D>::run(acc[0][0], acc[0][1], piq, ah[0], aex[0], a0, a0 - 16u * (unsigned)h + (unsigned)(2 * D + 32));
                popv[0] = __uint_as_float(piq[0]);
                locv[0] = (int)piq[1];
                popv[NB - 1] = __uint_as_float(piq[2]);
                locv[NB - 1] = (int)piq[3];
                asm_done = true;
                if (asked) {                      // the hand-over words were read in front of the block: an LDS round trip hidden
                    unsigned mn = 0xFFFFFFFFu;
#pragma unroll
                    for (int z = 0; z < kLoaders; ++z) mn = min(mn, lnd[z]);
                    landed_c = mn;
                }
            }
        }
        if (!asm_done) {
            auto b_load = [&](const unsigned char* base, int s_) __attribute__((always_inline)) -> u32x4 {
                const int m = s_ / NB, cb = s_ % NB;
                return *reinterpret_cast<const u32x4*>(base + cb * HB + 32 * m);
            };
            if (!PFX || b == 0) {
#pragma unroll
                for (int s_ = 0; s_ < PF; ++s_) bq[s_] = b_load(tb, s_);
            }
            u32x4 bx[NB];
            uint2 pi[NB];
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                bx[cb] = *reinterpret_cast<const u32x4*>(tb + cb * HB + 2 * D);
                pi[cb] = *reinterpret_cast<const uint2*>(tb - 16 * h + cb * HB + 2 * D + 32);
            }
#pragma unroll
            for (int s_ = 0; s_ < S; ++s_) {
                const int m = s_ / NB, cb = s_ % NB;
#pragma unroll
                for (int u = 0; u < UA; ++u)
                    acc[u][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[u][m]), __builtin_bit_cast(bf16x8, bq[s_ % PF]),
                                                                         m == 0 ? zero16v() : acc[u][cb], 0, 0, 0);
                if (s_ + PF < S) {
                    bq[s_ % PF] = b_load(tb, s_ + PF);
                } else if (has_next) {
                    if (s_ + PF == S) {
                        unsigned mn = landed_c;
                        if (mn < (unsigned)(b + 2)) {
                            mn = 0xFFFFFFFFu;
#pragma unroll
                            for (int z = 0; z < kLoaders; ++z) mn = min(mn, lnd[z]);
                            landed_c = mn;
                        }
                        ensure_landed(b + 1);
                    }
                    bq[s_ % PF] = b_load(tbn, s_ + PF - S);
                }
            }
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                popv[cb] = __uint_as_float(pi[cb].x);
                locv[cb] = (int)pi[cb].y;
                if constexpr (HEAD == PDA_HEAD_RAW) {
                    // raw head on any prep: 1/pop := 1, constant := +8e-6 (a prep built with a popularity carries its pieces);
                    // null items keep their -3e38
                    const bool nul = !(popv[cb] == popv[cb]);
                    bx[cb][0] = h ? (nul ? 0x0000FF61u : bf16_up(8.0e-6f)) : 0x00003F80u;
                    bx[cb][1] = h ? bx[cb][1] : 0x00003F80u;
                    bx[cb][2] = h ? 0u : 0x3F800000u;
                    bx[cb][3] = 0u;
                }
#pragma unroll
                for (int u = 0; u < UA; ++u)
                    acc[u][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aex[u]), __builtin_bit_cast(bf16x8, bx[cb]), acc[u][cb], 0, 0, 0);
            }
        }
        // (the votes about this tile, read while the MFMAs run and BEFORE the release -- a wave that sees it may store its vote
        // four tiles on into the same words; the block is finished either way)
        if (ES && tile_first && it >= kVL && ((it - kVL) & (kVoteEvery - 1)) == kVoteEvery - 1) {
            const unsigned* v = &s_vote[((it - kVL) & 7) * kMainWaves];
            unsigned all = 1u;
#pragma unroll
            for (int z = 0; z < MW; ++z) all &= lds_ld(&v[z]);
            stopped = all != 0u;
        }
        PDA_CBAR();
        lds_st(&s_released[w], (unsigned)(b + 1));
        PDA_CBAR();
        // ---- the filter: "some register of the lane is negative" = the sign bit of the OR of the bit patterns.  (This stalls
        // the wave until its MFMAs are through; the other MFMA wave of the SIMD has the matrix pipe meanwhile.) ----
        if constexpr ((PDA_V4_ABL & 1) != 0) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int cb = 0; cb < NB; ++cb)
                for (int u = 0; u < UA; ++u) asm volatile("" ::"v"(acc[u][cb]));       // timing only: no filter, but the MFMAs stay
#endif
        } else {
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
#pragma unroll
                for (int u = 0; u < UA; ++u) {
                    uint32_t mo = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) mo |= (uint32_t)__float_as_int(acc[u][cb][r]);
                    bool clampy = false;
                    if constexpr (HEAD == PDA_HEAD_POP) clampy = __any(popv[cb] > thr_min[u]);     // s~ + eps < 0: head <= pop; rare once the lists are warm
                    if (__any((int)mo < 0) || clampy) {
                        PROF_T0(ts);
                        PROF_INC(4, 1);
                        uint32_t mcb = 0;
                        if (__any((int)mo < 0)) {
                            // the exact mask: bit 15 - r <-> register r is negative (v_alignbit shifts the sign bit in)
#pragma unroll
                            for (int r = 0; r < 16; ++r) mcb = __builtin_amdgcn_alignbit(mcb, (uint32_t)__float_as_int(acc[u][cb][r]), 31);
                        }
                        if (clampy) {
                            int hv = h;
#if defined(__HIP_DEVICE_COMPILE__)
                            asm volatile("" : "+v"(hv));
#endif
#pragma unroll
                            for (int r = 0; r < 16; ++r) mcb |= (popv[cb] > thr_of(r, hv, u)) ? (1u << (15 - r)) : 0u;
                        }
                        push_mask(mcb, locv[cb], u);
                        PDA_CBAR();
                        publish_tails();
                        PROF_T1(ts, 3);
                    }
                }
            }
        }
        if (pr_tv != tver_seen) {
            tver_seen = pr_tv;
            refresh_thr();
            PROF_INC(5, 1);
        }
        if (NB == 1 && (b & 1) == 0) continue;
        ++n_done;                              // a 64-item tile is complete
    }
    PROF_T1(tm0, 0);
    PROF_INC(13, 1);
    PROF_INC(15, tail[0] + tail[RPW - 1] * (RPW - 1));
    PROF_FLUSH(0, 5);
    PROF_FLUSH(13, 15);
    PDA_CBAR();
    publish_tails();
    PDA_CBAR();
    lds_st(&s_done[w], 1u);
    if (stopped) lds_st(s_stop, 1u);
    if (lane == 0 && w == 0) atomicAdd(reinterpret_cast<unsigned long long*>(g.stats + 2), (unsigned long long)(2 * n_done * (UT / kUserTile)));
    }
    PDA_STAMP(st4);
    // ================================== all waves: sort and emit ==================================
    // Behind this barrier every candidate has been rescored and nobody appends any more (the rescoring waves arrive last).  The
    // lists are exact; what is left is their order -- rows the rescoring waves did not get to in their idle time -- and the
    // copy out: shared by all waves (64 rows per rescoring wave, one after the other, were the tail of every workgroup).
    __syncthreads();
    PDA_STAMP(st5);
    // (The sort is bound by the compares -- ~57 x 3 issue slots per row, four waves per SIMD at it: 20 - 35 us of a workgroup's
    // ~130 in an early-terminating sweep; ranking four rows at a time, on the 32-bit score halves, measured no faster.)
    constexpr int EB = 8;                  // (the rows' places in out_keys: eight loads of the permutation in flight)
    for (int r0 = wave; r0 < UT; r0 += EB * G::WAVES) {
        int rov[EB];
#pragma unroll
        for (int q = 0; q < EB; ++q) {
            const int rb = utile * UT + r0 + q * G::WAVES;
            rov[q] = (r0 + q * G::WAVES < UT && rb < g.n_users_blk) ? orig_row(rb) : -1;
        }
#pragma unroll
        for (int q = 0; q < EB; ++q) {
            const int rr = r0 + q * G::WAVES;
            if (rr >= UT) break;
            uint64_t* buf = lists + (size_t)rr * CAPL;
            compact_list<CAPL, GL>(buf, &cntl[rr], &taul[rr], K, lane, &s_uns[rr >> 5], 1u << (rr & 31));
            const int c = cntl[rr];
            if (rov[q] >= 0 && lane < K) {
                const uint64_t k = lane < c ? buf[lane] : 0ull;
                g.out_keys[((size_t)split * g.n_users_blk + rov[q]) * K + lane] = k;
            }
        }
    }
#ifdef PDA_V4_STAMP
    if (ES && lane == 0 && (wave == 0 || wave == G::WAVES - 1) && (blockIdx.x % 100) == 7) {
        const unsigned long long st6 = __builtin_amdgcn_s_memrealtime();
        printf("wg %4d wave %2d start %llu: prologue %llu  A rows %llu  barrier %llu  sweep %llu  wait-all %llu  epilogue %llu (x10 ns) tiles %d\n", (int)blockIdx.x, wave,
               st0, st1 - st0, st2 - st1, st3 - st2, st4 - st3, st5 - st4, st6 - st5, n_it);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Regrouping the users of an early-terminating sweep.  A workgroup sweeps until its LAST user is done; with the users as they
// come, the slowest of 256 decides (116 tiles on average at config 3 where a user needs ~50).  After the warm-up a user's
// stopping tile is predictable -- the first tile whose suffix bound falls below the K-th value of its warm-up list --, so the
// sweep takes the users sorted by that prediction (longest first): workgroups of alike users.  A counting sort on 1024 bins;
// the sweep reads and writes every per-user array through row_perm, the caller sees nothing of it.
// ---------------------------------------------------------------------------------------------------------------------
template <int D, bool BF>
__global__ void __launch_bounds__(256) stop_predict4_kernel(Args4 g, int* __restrict__ bin_of) {
    const int tid = threadIdx.x, sub = tid & 7;
    const int u = (int)blockIdx.x * 32 + (tid >> 3);
    if (u >= g.n_users_blk) return;
    if (g.pred_ws != nullptr) {           // the warm-up of this call left both numbers: 8 bytes per user instead of the row and K keys
        if (sub != 0) return;
        float tau = g.pred_ws[2 * (size_t)u];
        if (g.seed != nullptr) tau = fmaxf(tau, g.seed[u]);
        const float nu = g.pred_ws[2 * (size_t)u + 1];
        int lo = min(g.warm_tiles, g.n_tiles), hi = g.n_tiles;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (__builtin_fmaf(nu, g.sufB[mid], g.sufA[mid]) * 1.000002f < tau) hi = mid; else lo = mid + 1;
        }
        bin_of[u] = 1023 - min(1023, (int)((long long)lo * 1024 / (g.n_tiles + 1)));
        return;
    }
    const int uid = g.users[u];
    float ss = 0.f;
    for (int c = sub * 4; c < D; c += 32) {
        const f32x4 x = pda_load4<BF>(g.U, (size_t)uid * D + c);
        ss += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
    }
    uint32_t mn = 0xFFFFFFFFu;
    int cnt = 0;
    for (int k = sub; k < g.K; k += 8) {
        const uint64_t key = g.lists_empty ? 0ull : g.out_keys[(size_t)u * g.K + k];
        if (key != 0ull) { mn = min(mn, (uint32_t)(key >> 32)); ++cnt; }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        ss += __shfl_xor(ss, o, 64);
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
        cnt += __shfl_xor(cnt, o, 64);
    }
    if (sub != 0) return;
    float tau = cnt >= g.K ? pda_unordf(mn) : -INFINITY;
    if (g.seed != nullptr) tau = fmaxf(tau, g.seed[u]);
    const float nu = sqrtf(ss) * 1.0009765625f * 1.0001f;
    int lo = min(g.warm_tiles, g.n_tiles), hi = g.n_tiles;          // first tile t in [lo, hi) with bound(t) < tau; hi: never
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (__builtin_fmaf(nu, g.sufB[mid], g.sufA[mid]) * 1.000002f < tau) hi = mid; else lo = mid + 1;
    }
    const int bin = 1023 - min(1023, (int)((long long)lo * 1024 / (g.n_tiles + 1)));      // longest first
    bin_of[u] = bin;
}
// the users pile up in a few dozen bins: counted in LDS per 4096 users, one global add per workgroup and bin in use
__global__ void __launch_bounds__(1024) stop_hist4_kernel(const int* __restrict__ bin_of, int n, int* __restrict__ bins) {
    __shared__ int sh[1024];
    const int t = threadIdx.x;
    sh[t] = 0;
    __syncthreads();
    for (int i = 0; i < 4; ++i) {
        const int u = (int)blockIdx.x * 4096 + i * 1024 + t;
        if (u < n) atomicAdd(&sh[bin_of[u]], 1);
    }
    __syncthreads();
    if (sh[t]) atomicAdd(&bins[t], sh[t]);
}
__global__ void __launch_bounds__(1024) stop_scan4_kernel(int* __restrict__ bins) {      // counts -> exclusive starts, in place
    __shared__ int sh[1024];
    const int t = threadIdx.x;
    const int v = bins[t];
    sh[t] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int a = t >= o ? sh[t - o] : 0;
        __syncthreads();
        sh[t] += a;
        __syncthreads();
    }
    bins[t] = sh[t] - v;
}
__global__ void __launch_bounds__(1024) stop_scatter4_kernel(const int* __restrict__ bin_of, int* __restrict__ bins, int n, int32_t* __restrict__ perm) {
    __shared__ int sh[1024];
    const int t = threadIdx.x;
    sh[t] = 0;
    __syncthreads();
    int b[4], r[4];
    for (int i = 0; i < 4; ++i) {
        const int u = (int)blockIdx.x * 4096 + i * 1024 + t;
        b[i] = u < n ? bin_of[u] : -1;
        r[i] = b[i] >= 0 ? atomicAdd(&sh[b[i]], 1) : 0;          // rank within the workgroup's share of the bin
    }
    __syncthreads();
    const int c = sh[t];
    __syncthreads();
    sh[t] = c ? atomicAdd(&bins[t], c) : 0;                      // where that share starts
    __syncthreads();
    for (int i = 0; i < 4; ++i)
        if (b[i] >= 0) perm[sh[b[i]] + r[i]] = (int)blockIdx.x * 4096 + i * 1024 + t;
}

#include "pda_v5_sweep.h"

template <int D, int HEAD, bool BF, int GM>
int launch_sweep4(const Args4& g, hipStream_t stream) {
    using G = Geo4<D, GM>;
    constexpr bool kHasES = true;
    static int attr_set = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep4_kernel<D, HEAD, BF, false, GM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)G::lds_total) != hipSuccess)
            return PDA_ERR_LAUNCH;
        if constexpr (kHasES) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep4_kernel<D, HEAD, BF, true, GM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)G::lds_total) != hipSuccess)
                return PDA_ERR_LAUNCH;
        }
        attr_set = 1;
    }
    const int utiles = (g.n_users_blk + G::UT - 1) / G::UT;
    if constexpr (kHasES) {
        if (g.sufA != nullptr && g.regroup_ws != nullptr) {
            Args4 gp = g;
            int* bins = g.regroup_ws;
            int* bin_of = bins + 1024;
            int32_t* perm = bin_of + g.n_users_blk;
            if (hipMemsetAsync(bins, 0, 1024 * sizeof(int), stream) != hipSuccess) return PDA_ERR_LAUNCH;
            hipLaunchKernelGGL((stop_predict4_kernel<D, BF>), dim3((unsigned)((g.n_users_blk + 31) / 32)), dim3(256), 0, stream, g, bin_of);
            const unsigned hb = (unsigned)((g.n_users_blk + 4095) / 4096);
            hipLaunchKernelGGL(stop_hist4_kernel, dim3(hb), dim3(1024), 0, stream, bin_of, g.n_users_blk, bins);
            hipLaunchKernelGGL(stop_scan4_kernel, dim3(1), dim3(1024), 0, stream, bins);
            hipLaunchKernelGGL(stop_scatter4_kernel, dim3(hb), dim3(1024), 0, stream, bin_of, bins, g.n_users_blk, perm);
            PDA_CHECK_LAUNCH();
            gp.row_perm = perm;
            hipLaunchKernelGGL((sweep4_kernel<D, HEAD, BF, true, GM>), dim3((unsigned)(utiles * g.n_splits)), dim3(64 * G::WAVES), G::lds_total, stream, gp);
            PDA_CHECK_LAUNCH();
            return PDA_OK;
        }
        if (g.sufA != nullptr) {
            hipLaunchKernelGGL((sweep4_kernel<D, HEAD, BF, true, GM>), dim3((unsigned)(utiles * g.n_splits)), dim3(64 * G::WAVES), G::lds_total, stream, g);
            PDA_CHECK_LAUNCH();
            return PDA_OK;
        }
    } else if (g.sufA != nullptr) {
        return PDA_ERR_ARG;
    }
    hipLaunchKernelGGL((sweep4_kernel<D, HEAD, BF, false, GM>), dim3((unsigned)(utiles * g.n_splits)), dim3(64 * G::WAVES), G::lds_total, stream, g);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

template <int D, int HEAD, bool BF>
int launch4_sweep(const Args4& g, hipStream_t stream, int geometry);

template <int D, int HEAD, bool BF>
int launch4(const Args4& g, int phase, hipStream_t stream, int geometry) {      // phase: 1 = warm-up only, 2 = sweep only, 3 = both
    if (phase & 1) {
        constexpr int CAP = kCap4;
        const size_t smem = 32 * D * 4 + (size_t)kUserTile * (CAP * 8 + 8) + kUserTile * 2 * kWarmTiles * 4 + 256 + kUserTile * 8;
        static int attr_set = 0;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&warm4_kernel<D, HEAD, BF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem) != hipSuccess)
                return PDA_ERR_LAUNCH;
            attr_set = 1;
        }
        const int utiles = (g.n_users_blk + kUserTile - 1) / kUserTile;
        Args4 gw = g;
        if (g.warm_shared) {             // ONE warm-up per user: "split 0 of 1" scores tiles 0 .. warm_tiles - 1 of the whole order
            gw.n_splits = 1;
            gw.warm_shared = 0;
        }
        hipLaunchKernelGGL((warm4_kernel<D, HEAD, BF>), dim3((unsigned)(utiles * gw.n_splits)), dim3(kThreads), smem, stream, gw);
        PDA_CHECK_LAUNCH();
    }
    if ((g.n_tiles + g.n_splits - 1) / g.n_splits <= g.warm_tiles) return PDA_OK;      // every split ends inside its warm-up
    if (phase & 2) {
        Args4 gs = g;
        if (g.warm_shared) gs.seed = g.seed_out;      // every split prunes against the shared warm-up's K-th value
        return launch4_sweep<D, HEAD, BF>(gs, stream, geometry);
    }
    return PDA_OK;
}

template <int D, int HEAD, bool BF>
int launch4_sweep(const Args4& g, hipStream_t stream, int geometry) {
    {
        // the huge geometry (pda_v5_sweep.h): dense sweeps of the popularity head on a prep built WITH that popularity -- 1 024-user workgroups,
        // or (d = 256, config 5) 512-user ones: 128 users per wave fill the 256 AGPRs.  Anything else asked for with that hint: the default.
        if constexpr (HEAD == PDA_HEAD_POP) {
            if (geometry == 4 && g.sufA == nullptr && g.prep_hdr_pop != 0) return launch_sweep5<D, BF, true, D == 256 ? 128 : 256>(g, stream);
        }
        if constexpr (D <= 128) {
            if (geometry == 3) return launch_sweep4<D, HEAD, BF, 3>(g, stream);
        }
        return launch_sweep4<D, HEAD, BF, 0>(g, stream);
    }
    return PDA_OK;
}

int user_tile4(int d) { return d == 64 ? Geo4<64>::UT : d == 128 ? Geo4<128>::UT : Geo4<256>::UT; }
static bool lists_in_hbm4(int) { return true; }       // (every d may run with its lists in the workspace since round 3: Geo4<D, true>)
// the workspace of the pda_score_topk4_* calls: the counters of pda_score_topk_workspace_bytes, then (workgroups of 512 users:
// d <= 128) the list slots of every workgroup
// the workspace of the pda_score_topk4_* calls: [counters of pda_score_topk_workspace_bytes | list slots of every workgroup when the
// lists live in HBM | Bloom filters, 128 B per user | warm-position train-item masks, 32 B per user and split | regrouping: 1024 bins, bin and sweep row of every user]
struct Ws4 {
    size_t lists, bloom, hmask, regroup, handover, ufrag, unorm, seed, total;
};
static Ws4 ws4_layout(int n_users_blk, int d, int n_splits) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    Ws4 w{};
    w.lists = al(pda_score_topk_workspace_bytes(n_users_blk));
    size_t b = w.lists;
    if (lists_in_hbm4(d)) {
        const size_t ut = 1024;         // (the widest user tile of any geometry: the huge one's)
        b += ((size_t)n_users_blk + ut - 1) / ut * (size_t)n_splits * ut * kCap4 * 8 + 256;
    }
    w.bloom = al(b);
    w.hmask = al(w.bloom + (size_t)n_users_blk * 128);
    w.regroup = al(w.hmask + ((size_t)n_users_blk + kUserTile - 1) / kUserTile * (size_t)n_splits * kUserTile * 2 * kWarmTiles * 4);
    w.handover = al(w.regroup + (1024 + 4 * (size_t)n_users_blk) * 4);       // (behind: bins, bin_of, row_perm, pred_ws)
    w.total = w.handover + (size_t)n_splits * (size_t)n_users_blk * kCap4 * 8;
    // the huge geometry (d <= 128): the user block as bf16 MFMA operands (whole 1 024-user workgroups) and the users' padded norms
    w.ufrag = al(w.total);
    w.unorm = w.ufrag;
    {
        const size_t n_pad = ((size_t)n_users_blk + kUT5 - 1) / kUT5 * kUT5;
        w.unorm = w.ufrag + al(n_pad * 2 * (size_t)d);
        w.total = w.unorm + al(n_pad * 4);
    }
    w.seed = w.total;                                   // the shared warm-up's seed: one float per user
    w.total = w.seed + al((size_t)n_users_blk * 4);
    return w;
}
extern "C" size_t pda_score_topk4_workspace_bytes(int n_users_blk, int n_items_local, int d, int n_splits) {
    if (n_users_blk <= 0 || n_items_local <= 0 || (d != 64 && d != 128 && d != 256)) return 0;
    if (n_splits <= 0) n_splits = pda_score_topk4_auto_splits(n_users_blk, n_items_local, d);
    return ws4_layout(n_users_blk, d, n_splits).total;
}

int run_score4(const void* U, const void* I_shard, bool bf16, const void* prep, const float* pop_shard, const int32_t* users,
               int n_users_blk, int item_offset, int n_items_local, int d, const int64_t* hist_indptr, const int32_t* hist_indices,
               int hist_row_mode, int K, int head, int early_stop, int n_splits, uint64_t* out_keys, void* workspace, hipStream_t s,
               int phase = 3, const float* seed = nullptr, int warm_tiles = 0, const int* n_users_dev = nullptr) {
    if (!U || !I_shard || !prep || !users || !out_keys || !workspace) return PDA_ERR_ARG;
    if (n_users_blk <= 0 || n_items_local <= 0 || item_offset < 0) return PDA_ERR_ARG;
    if (K < 1 || K > PDA_MAX_K) return PDA_ERR_ARG;
    if (early_stop < 0 || (early_stop & ~0x7FF) != 0) return PDA_ERR_ARG;
    const bool warm_per_split = (early_stop & PDA_SWEEP_WARM_PER_SPLIT) != 0;
    // geometry hints (Geo4<D, 3>, sweep5_kernel): results do not depend on them, and every geometry takes any n_splits and any user count
    // (tests/test_gpu_score_topk.py runs each with 1 / 2 / 3 / 8 splits and ragged blocks)
    // (PDA_SWEEP_FEW_CANDIDATES, PDA_SWEEP_WIDE and the two PDA_SWEEP_HUGE_* sub-variants named geometries that round 5 removed -- each
    // measured slower than the huge geometry wherever it applied, profiles/README.md; the bits stay valid and select nothing)
    int geometry = (early_stop & PDA_SWEEP_HUGE) ? 4 : (early_stop & PDA_SWEEP_MANY_CANDIDATES) ? 3 : 0;
    if (warm_tiles == 0) warm_tiles = (early_stop >> 4) & 7;                       // PDA_SWEEP_WARM_TILES(n)
    early_stop &= 1;
    if (head != PDA_HEAD_RAW && head != PDA_HEAD_POP) return PDA_ERR_ARG;
    if (head == PDA_HEAD_POP && !pop_shard) return PDA_ERR_ARG;
    if (hist_indptr && !hist_indices) return PDA_ERR_ARG;
    if (d != 64 && d != 128 && d != 256) return PDA_ERR_UNSUPPORTED;
    if (K > kCap4 - 3) return PDA_ERR_UNSUPPORTED;
    if ((uint64_t)n_items_local > (1ull << 26)) return PDA_ERR_UNSUPPORTED;          // ring words: 6-bit row, 26-bit local item id
    if (warm_tiles < 0 || warm_tiles > kWarmTiles) return PDA_ERR_ARG;
#ifndef PDA_V4_WARM_DEFAULT
#define PDA_V4_WARM_DEFAULT kWarmTiles
#endif
    if (warm_tiles == 0) warm_tiles = d <= 128 ? PDA_V4_WARM_DEFAULT : kWarmTiles;
    // phase 4: a sweep of the WHOLE shard from empty lists against the caller's seed (no warm-up on this catalogue)
    const bool from_empty = phase == 4;
    if (from_empty) {
        if (seed == nullptr) return PDA_ERR_ARG;
        phase = 2;
        warm_tiles = 0;
    } else if (phase < 1 || phase > 3) {
        return PDA_ERR_ARG;
    }
    if (n_splits <= 0) n_splits = pda_score_topk4_auto_splits(n_users_blk, n_items_local, d);
    const Prep4Layout L = prep4_layout(n_items_local, d);
    const unsigned char* pb = reinterpret_cast<const unsigned char*>(prep);
    if (((phase & 1) || from_empty) && hipMemsetAsync(workspace, 0, pda_score_topk_workspace_bytes(n_users_blk), s) != hipSuccess) return PDA_ERR_LAUNCH;
    const Ws4 W = ws4_layout(n_users_blk, d, n_splits);
    unsigned char* wsb = reinterpret_cast<unsigned char*>(workspace);
    const float* sA = reinterpret_cast<const float*>(pb + (head == PDA_HEAD_POP ? L.sufA : L.sufB));
    const float* sB = reinterpret_cast<const float*>(pb + (head == PDA_HEAD_POP ? L.sufB : L.sufR));
    // raw head: bound = ||u|| max ||i||: sufA := 0 is not stored -- the raw-head vote uses (sufB' = sufR, sufA' = 0) through
    // the same fma; a zero array is the front of sufA of a prep WITHOUT popularity (tile_bound4_kernel, has_pop = 0)
    Args4 g{U, I_shard, pop_shard, users, hist_indptr, hist_indices, out_keys, pb + L.rows,
            early_stop ? sA : nullptr, early_stop ? sB : nullptr, reinterpret_cast<const int*>(pb + L.pos_of),
            reinterpret_cast<unsigned*>(workspace), reinterpret_cast<uint64_t*>(wsb + W.lists), nullptr, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<const int*>(pb + L.hdr), seed, n_users_blk, item_offset, n_items_local, hist_row_mode, K, n_splits, L.n_tiles, warm_tiles,
            // sorted hand-over when nobody sorts behind the warm-up: phase 1 alone, or a catalogue that ends inside the warm-up
#ifdef PDA_V4_WARM_SORTED
            1, nullptr, pb + L.rows5, reinterpret_cast<const float*>(pb + L.meta5), nullptr, nullptr, 0};
#else
            (phase == 1 || (L.n_tiles + n_splits - 1) / n_splits <= warm_tiles) ? 1 : 0, nullptr,
            pb + L.rows5, reinterpret_cast<const float*>(pb + L.meta5), nullptr, nullptr, 0};
#endif
    // warm-up and sweep in one call: K .. kCap4 keys per row through the workspace (see warm4_kernel)
    if (phase == 3 && !g.warm_sorted && PDA_V4_HANDOVER_WS) g.handover = reinterpret_cast<uint64_t*>(wsb + W.handover);
    g.ufrag = wsb + W.ufrag;
    g.unorm = reinterpret_cast<const float*>(wsb + W.unorm);
    // (the prep is the caller's, built by pda_item_prep4_* with the SAME pop_shard it passes here for the popularity head: ops.item_prep4
    // keys its cache on it; a prep built without one carries an unscaled image and the huge geometry falls back to the default one)
    g.prep_hdr_pop = (head == PDA_HEAD_POP && pop_shard != nullptr) ? 1 : 0;
    g.warm_final = (geometry >= 4 && phase == 3 && g.handover != nullptr && g.prep_hdr_pop && !early_stop) ? 1 : 0;
    g.lists_empty = from_empty ? 1 : 0;
    g.n_users_dev = n_users_dev;
    // one call over several item splits: ONE exact warm-up per user instead of one per split (warm_tiles_of; PDA_SWEEP_WARM_PER_SPLIT)
    if (phase == 3 && n_splits > 1 && seed == nullptr && !warm_per_split && L.n_tiles > n_splits * warm_tiles) {
        g.warm_shared = 1;
        g.seed_out = reinterpret_cast<float*>(wsb + W.seed);
    }
    if (hist_indptr && (phase & 2)) {
        uint32_t* bloom = reinterpret_cast<uint32_t*>(wsb + W.bloom);
        hipLaunchKernelGGL(hist_bloom4_kernel, dim3((unsigned)((n_users_blk + 31) / 32)), dim3(256), 0, s, users, hist_indptr, hist_indices,
                           hist_row_mode, n_users_blk, bloom, reinterpret_cast<const int*>(pb + L.hdr), head == PDA_HEAD_POP ? 1 : 0);
        PDA_CHECK_LAUNCH();
        g.bloom = bloom;
    }
    if (hist_indptr && (phase & 1) && n_users_blk >= 98304) {
        // the train-item bits of the warm positions, by a kernel of its own (see warm_mask4_kernel) -- where it pays for its launch:
        // 1.35 -> 1.29 ms per early-terminating sweep of 262 144 users, but 0.267 -> 0.277 ms at 50 000
        uint32_t* hm = reinterpret_cast<uint32_t*>(wsb + W.hmask);
        const int utiles = (n_users_blk + kUserTile - 1) / kUserTile;
        Args4 gm = g;
        if (g.warm_shared) {
            gm.n_splits = 1;
            gm.warm_shared = 0;
        }
        hipLaunchKernelGGL(warm_mask4_kernel, dim3((unsigned)(utiles * gm.n_splits)), dim3(kThreads), 0, s, gm, hm);
        PDA_CHECK_LAUNCH();
        g.hmask_ws = hm;
    }
#ifndef PDA_V4_REGROUP_MIN
#define PDA_V4_REGROUP_MIN 98304
#endif
    // users regrouped by predicted stopping tile (stop_predict4_kernel): where the workgroups come in more than one round
    if (early_stop && (phase & 2) && n_splits == 1 && n_users_blk >= PDA_V4_REGROUP_MIN) {
        g.regroup_ws = reinterpret_cast<int32_t*>(wsb + W.regroup);
        if (phase & 1) g.pred_ws = reinterpret_cast<float*>(g.regroup_ws + 1024 + 2 * (size_t)n_users_blk);       // (a sweep of its own reads the lists)
    }
    if (head == PDA_HEAD_RAW) {
        g.sufA = early_stop ? reinterpret_cast<const float*>(pb + L.sufA) : nullptr;     // all zero for a raw prep
        g.sufB = early_stop ? reinterpret_cast<const float*>(pb + L.sufR) : nullptr;
    }
#define PDA_V4_(DD, BFV) (head == PDA_HEAD_POP ? launch4<DD, PDA_HEAD_POP, BFV>(g, phase, s, geometry) : launch4<DD, PDA_HEAD_RAW, BFV>(g, phase, s, geometry))
    switch (d) {
        case 64: return bf16 ? PDA_V4_(64, true) : PDA_V4_(64, false);
        case 128: return bf16 ? PDA_V4_(128, true) : PDA_V4_(128, false);
        case 256: return bf16 ? PDA_V4_(256, true) : PDA_V4_(256, false);
        default: return PDA_ERR_UNSUPPORTED;
    }
#undef PDA_V4_
}

// the value at rank `pos` (0-based) of every user's (sorted, best first) partial lists: the largest over the item splits,
// -inf when no list is that long
__global__ void __launch_bounds__(256) kth_value_kernel(const uint64_t* __restrict__ keys, int S, int n_users, int K, int pos, float* __restrict__ out) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n_users) return;
    float best = -INFINITY;
    for (int sp = 0; sp < S; ++sp) {
        const uint64_t k = keys[((size_t)sp * n_users + u) * K + pos];
        if (k != 0ull) best = fmaxf(best, pda_key_val(k));
    }
    out[u] = best;
}

// One round of the cross-shard bisection that tightens the seed (pda_topk_seed_refine): [mode >= 1] the all-reduced count of
// the previous threshold decides which half survives -- K entries at or above it in the shards' warm-up lists together
// make it a lower bound of the final K-th value --, [mode <= 1] the next threshold is set and this shard's entries at or
// above it are counted.
__global__ void __launch_bounds__(256) seed_refine_kernel(const uint64_t* __restrict__ keys, int S, int n_users, int K, float* __restrict__ lo,
                                                          float* __restrict__ hi, float* __restrict__ mid, int32_t* __restrict__ counts, int mode) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n_users) return;
    float l = lo[u], h = hi[u];
    if (mode >= 1) {
        if (counts[u] >= K) l = mid[u];
        else h = mid[u];
        lo[u] = l;
        hi[u] = h;
    }
    if (mode <= 1) {
        const bool open = l > -INFINITY && h < INFINITY && h > l;
        const float m = open ? l + 0.5f * (h - l) : l;
        int c = 0;
        for (int sp = 0; sp < S; ++sp) {
            const uint64_t* row = keys + ((size_t)sp * n_users + u) * K;
            for (int q = 0; q < K; ++q) {
                const uint64_t k = row[q];
                if (k == 0ull || pda_key_val(k) < m) break;          // (sorted, best first)
                ++c;
            }
        }
        mid[u] = m;
        counts[u] = c;
    }
}


// ---- the seed exchange in TWO collectives (round 3; MAX, MIN and three sequential SUM rounds before) ----------------------
// bounds [3][n_users]: row 0 = the value at rank K - 1 of the user's warm-up lists (max over the splits), row 1 = the value at
// rank m - 1 (m = ceil(K / R)), row 2 = MINUS row 1.  ONE all-reduce MAX over the 3 n_users floats leaves: the largest K-th
// value, the largest m-th value (an upper bound of the K-th value of the merged warm-up lists) and minus the smallest m-th value
// (min x = -max -x: R shards with m entries above it hold K entries above it).
__global__ void __launch_bounds__(256) seed_bounds_kernel(const uint64_t* __restrict__ keys, int S, int n_users, int K, int m,
                                                          float* __restrict__ out) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n_users) return;
    float a = -INFINITY, b = -INFINITY;
    for (int sp = 0; sp < S; ++sp) {
        const uint64_t* row = keys + ((size_t)sp * n_users + u) * K;
        const uint64_t ka = row[K - 1], kb = row[m - 1];
        if (ka != 0ull) a = fmaxf(a, pda_key_val(ka));
        if (kb != 0ull) b = fmaxf(b, pda_key_val(kb));
    }
    out[u] = a;
    out[(size_t)n_users + u] = b;
    out[2 * (size_t)n_users + u] = -b;
}
// the common thresholds between the seed lo = max(bounds 0, -bounds 2) and the upper bound hi = bounds 1: the grid the three
// bisection rounds used to walk one round trip at a time (n_thr = 2^rounds - 1 interior points), computed identically on
// every rank and in both kernels below (-ffp-contract=off; the reduced bounds are the same bits everywhere)
__device__ __forceinline__ float seed_thr(float lo, float hi, int j, int n_thr) {
    const bool open = lo > -INFINITY && hi < INFINITY && hi > lo;
    return open ? lo + (hi - lo) * ((float)(j + 1) / (float)(n_thr + 1)) : lo;
}
constexpr int kSeedThrMax = 15;
// counts [n_thr][n_users]: this shard's warm-up entries at or above threshold j (sorted lists, best first)
__global__ void __launch_bounds__(256) seed_counts_kernel(const uint64_t* __restrict__ keys, int S, int n_users, int K,
                                                          const float* __restrict__ bounds, int n_thr, int32_t* __restrict__ counts) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n_users) return;
    const float lo = fmaxf(bounds[u], -bounds[2 * (size_t)n_users + u]), hi = bounds[(size_t)n_users + u];
    float thr[kSeedThrMax];
    int c[kSeedThrMax];
#pragma unroll
    for (int j = 0; j < kSeedThrMax; ++j) { thr[j] = seed_thr(lo, hi, j < n_thr ? j : 0, n_thr); c[j] = 0; }
    for (int sp = 0; sp < S; ++sp) {
        const uint64_t* row = keys + ((size_t)sp * n_users + u) * K;
        for (int q = 0; q < K; ++q) {
            const uint64_t k = row[q];
            if (k == 0ull) break;
            const float v = pda_key_val(k);
            if (v < thr[0]) break;                 // (thr[0] is the smallest threshold; the list is sorted)
#pragma unroll
            for (int j = 0; j < kSeedThrMax; ++j) c[j] += (v >= thr[j]) ? 1 : 0;
        }
    }
#pragma unroll
    for (int j = 0; j < kSeedThrMax; ++j)
        if (j < n_thr) counts[(size_t)j * n_users + u] = c[j];
}
// seed = the largest threshold at or above which the shards hold K warm-up entries between them (summed counts), else lo
__global__ void __launch_bounds__(256) seed_pick_kernel(const float* __restrict__ bounds, const int32_t* __restrict__ counts, int n_thr,
                                                        int n_users, int K, float* __restrict__ seed) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n_users) return;
    const float lo = fmaxf(bounds[u], -bounds[2 * (size_t)n_users + u]), hi = bounds[(size_t)n_users + u];
    float s = lo;
    for (int j = 0; j < n_thr; ++j)
        if (counts[(size_t)j * n_users + u] >= K) s = fmaxf(s, seed_thr(lo, hi, j, n_thr));
    seed[u] = s;
}

}  // namespace

// the funnel's exact fallback (pda_score_funnel.hip): generation 4 on a block whose size is a device-side count
int pda_v4_run_score4_dev(const void* U, const void* I_shard, bool bf16, const void* prep, const float* pop_shard, const int32_t* users, int n_users_blk,
                          const int* n_users_dev, int item_offset, int n_items_local, int d, const int64_t* hist_indptr, const int32_t* hist_indices,
                          int hist_row_mode, int K, int head, int early_stop, int n_splits, const float* seed, uint64_t* out_keys, void* workspace, hipStream_t s) {
    // seed: a lower bound of every row's final K-th value (the funnel's tk: K unmasked items reach it), or NULL.  With a seed the sweep runs from EMPTY
    // lists (phase 4): the exact warm-up would cost its ~100 us per 128-user workgroup to collect what the seed already prunes to ~K pairs per row
    // (round 6: configs 1 / 2, where one row of 50 000 goes through here in every call).
    return run_score4(U, I_shard, bf16, prep, pop_shard, users, n_users_blk, item_offset, n_items_local, d, hist_indptr, hist_indices, hist_row_mode, K, head,
                      early_stop, n_splits, out_keys, workspace, s, seed != nullptr ? 4 : 3, seed, 0, n_users_dev);
}

extern "C" int pda_topk_seed_bounds(const uint64_t* keys, int n_splits, int n_users_blk, int K, int m, float* bounds, void* stream) {
    if (!keys || !bounds || n_splits < 1 || n_users_blk < 1 || K < 1 || m < 1 || m > K) return PDA_ERR_ARG;
    hipLaunchKernelGGL(seed_bounds_kernel, dim3((unsigned)((n_users_blk + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), keys,
                       n_splits, n_users_blk, K, m, bounds);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}
extern "C" int pda_topk_seed_counts(const uint64_t* keys, int n_splits, int n_users_blk, int K, const float* bounds, int n_thr, int32_t* counts,
                                    void* stream) {
    if (!keys || !bounds || !counts || n_splits < 1 || n_users_blk < 1 || K < 1 || n_thr < 1 || n_thr > kSeedThrMax) return PDA_ERR_ARG;
    hipLaunchKernelGGL(seed_counts_kernel, dim3((unsigned)((n_users_blk + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), keys,
                       n_splits, n_users_blk, K, bounds, n_thr, counts);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}
extern "C" int pda_topk_seed_pick(const float* bounds, const int32_t* counts, int n_thr, int n_users_blk, int K, float* seed, void* stream) {
    if (!bounds || !seed || n_users_blk < 1 || K < 1 || n_thr < 0 || n_thr > kSeedThrMax || (n_thr > 0 && !counts)) return PDA_ERR_ARG;
    hipLaunchKernelGGL(seed_pick_kernel, dim3((unsigned)((n_users_blk + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), bounds,
                       counts, n_thr, n_users_blk, K, seed);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

#ifdef PDA_V5_LOG

extern "C" int pda_debug_v5_log(unsigned* out, int n_words, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pda_v5_log), (size_t)n_words * 4) != hipSuccess) return PDA_ERR_LAUNCH;
    if (reset) { unsigned z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(pda_v5_log), &z, 4) != hipSuccess) return PDA_ERR_LAUNCH; }
    return PDA_OK;
}
#endif
#ifdef PDA_V4_PROF
extern "C" int pda_debug_prof4(unsigned long long* out16, int reset) {     /* 24 words */
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(pda_prof4), sizeof(unsigned long long) * 24) != hipSuccess) return PDA_ERR_LAUNCH;
    if (reset) { unsigned long long z[24] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(pda_prof4), z, sizeof(z)) != hipSuccess) return PDA_ERR_LAUNCH; }
    return PDA_OK;
}
#endif

extern "C" int pda_score_topk4_auto_splits(int n_users_blk, int n_items_local, int d) {
    // Item splits only make up for too few user tiles (one workgroup per CU): every split pays its own exact warm-up on
    // 256 items, so a launch that already fills half the chip stays unsplit.  (A rule that jumped to 8 splits at 188 user
    // tiles cost 2.7 instead of 0.7 ms on the Douban-shaped config.)
    if (n_users_blk <= 0 || n_items_local <= 0) return 1;
    const int ut = user_tile4(d);
    const int utiles = (n_users_blk + ut - 1) / ut;
    const int n_tiles = (n_items_local + 63) / 64;
    int s = 1;
    while (utiles * s * 2 <= 256 && s < 64 && n_tiles / (s * 2) >= 3 * kWarmTiles) s *= 2;
    return s;
}

extern "C" size_t pda_item_prep4_bytes(int n_items_local, int d) { return n_items_local > 0 ? prep4_layout(n_items_local, d).total : 0; }

extern "C" int pda_item_prep4_f32(const float* I_shard, const float* pop_shard, const int32_t* order, int n_items_local, int d, void* prep,
                                  void* stream) {
    if (!I_shard || !prep || n_items_local <= 0) return PDA_ERR_ARG;
    return run_prep4(I_shard, false, pop_shard, order, n_items_local, d, prep, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pda_item_prep4_bf16(const uint16_t* I_shard, const float* pop_shard, const int32_t* order, int n_items_local, int d, void* prep,
                                   void* stream) {
    if (!I_shard || !prep || n_items_local <= 0) return PDA_ERR_ARG;
    return run_prep4(I_shard, true, pop_shard, order, n_items_local, d, prep, reinterpret_cast<hipStream_t>(stream));
}
// the funnel's prep (pda_score_topk7_*): as pda_item_prep4_* without a popularity, the half-tile image in fp16 (header word 3)
extern "C" int pda_item_prep7_f32(const float* I_shard, const int32_t* order, int n_items_local, int d, void* prep, void* stream) {
    if (!I_shard || !prep || !order || n_items_local <= 0) return PDA_ERR_ARG;
    return run_prep4(I_shard, false, nullptr, order, n_items_local, d, prep, reinterpret_cast<hipStream_t>(stream), 1);
}
extern "C" int pda_item_prep7_bf16(const uint16_t* I_shard, const int32_t* order, int n_items_local, int d, void* prep, void* stream) {
    if (!I_shard || !prep || !order || n_items_local <= 0) return PDA_ERR_ARG;
    return run_prep4(I_shard, true, nullptr, order, n_items_local, d, prep, reinterpret_cast<hipStream_t>(stream), 1);
}
extern "C" int pda_item_prep4_check(const void* prep, int n_items_local, int d, void* stream) {
    if (!prep || n_items_local <= 0) return PDA_ERR_ARG;
    int bad = 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(&bad, prep, 4, hipMemcpyDeviceToHost, s) != hipSuccess) return PDA_ERR_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return PDA_ERR_LAUNCH;
    return bad ? PDA_ERR_ARG : PDA_OK;
}

extern "C" int pda_topk_kth_value(const uint64_t* keys, int n_splits, int n_users_blk, int K, int pos, float* out, void* stream) {
    if (!keys || !out || n_splits < 1 || n_users_blk < 1 || K < 1 || pos < 0 || pos >= K) return PDA_ERR_ARG;
    hipLaunchKernelGGL(kth_value_kernel, dim3((unsigned)((n_users_blk + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), keys,
                       n_splits, n_users_blk, K, pos, out);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

extern "C" int pda_topk_seed_refine(const uint64_t* keys, int n_splits, int n_users_blk, int K, float* lo, float* hi, float* mid,
                                    int32_t* counts, int mode, void* stream) {
    if (!keys || !lo || !hi || !mid || !counts || n_splits < 1 || n_users_blk < 1 || K < 1 || mode < 0 || mode > 2) return PDA_ERR_ARG;
    hipLaunchKernelGGL(seed_refine_kernel, dim3((unsigned)((n_users_blk + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), keys,
                       n_splits, n_users_blk, K, lo, hi, mid, counts, mode);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

// The two phases of pda_score_topk4_* as separate calls, for the item-sharded evaluation with exact early termination: every
// rank runs the warm-up, the ranks exchange two values per user of their warm-up lists (pda_topk_kth_value: the K-th -- the
// maximum over the ranks bounds the final K-th value from below -- and the ceil(K / R)-th -- R ranks with ceil(K / R) items
// above the MINIMUM over the ranks make K items above it: with the popular items spread over the shards this is the bound
// of a single GPU's warm-up), and the sweep takes the larger of the two as `seed`.
extern "C" int pda_score_topk4_phase_f32(const float* U, const float* I_shard, const void* prep, const float* pop_shard, const int32_t* users,
                                         int n_users_blk, int item_offset, int n_items_local, int d, const int64_t* hist_indptr,
                                         const int32_t* hist_indices, int hist_row_mode, int K, int head, int early_stop, int n_splits,
                                         int phase, int warm_tiles, const float* seed, uint64_t* out_keys, void* workspace, void* stream) {
    if (phase != 1 && phase != 2 && phase != 4) return PDA_ERR_ARG;
    return run_score4(U, I_shard, false, prep, pop_shard, users, n_users_blk, item_offset, n_items_local, d, hist_indptr, hist_indices,
                      hist_row_mode, K, head, early_stop, n_splits, out_keys, workspace, reinterpret_cast<hipStream_t>(stream), phase, seed, warm_tiles);
}
extern "C" int pda_score_topk4_phase_bf16(const uint16_t* U, const uint16_t* I_shard, const void* prep, const float* pop_shard,
                                          const int32_t* users, int n_users_blk, int item_offset, int n_items_local, int d,
                                          const int64_t* hist_indptr, const int32_t* hist_indices, int hist_row_mode, int K, int head,
                                          int early_stop, int n_splits, int phase, int warm_tiles, const float* seed, uint64_t* out_keys,
                                          void* workspace, void* stream) {
    if (phase != 1 && phase != 2 && phase != 4) return PDA_ERR_ARG;
    return run_score4(U, I_shard, true, prep, pop_shard, users, n_users_blk, item_offset, n_items_local, d, hist_indptr, hist_indices,
                      hist_row_mode, K, head, early_stop, n_splits, out_keys, workspace, reinterpret_cast<hipStream_t>(stream), phase, seed, warm_tiles);
}

extern "C" int pda_score_topk4_f32(const float* U, const float* I_shard, const void* prep, const float* pop_shard, const int32_t* users,
                                   int n_users_blk, int item_offset, int n_items_local, int d, const int64_t* hist_indptr,
                                   const int32_t* hist_indices, int hist_row_mode, int K, int head, int early_stop, int n_splits,
                                   uint64_t* out_keys, void* workspace, void* stream) {
    return run_score4(U, I_shard, false, prep, pop_shard, users, n_users_blk, item_offset, n_items_local, d, hist_indptr, hist_indices,
                      hist_row_mode, K, head, early_stop, n_splits, out_keys, workspace, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pda_score_topk4_bf16(const uint16_t* U, const uint16_t* I_shard, const void* prep, const float* pop_shard, const int32_t* users,
                                    int n_users_blk, int item_offset, int n_items_local, int d, const int64_t* hist_indptr,
                                    const int32_t* hist_indices, int hist_row_mode, int K, int head, int early_stop, int n_splits,
                                    uint64_t* out_keys, void* workspace, void* stream) {
    return run_score4(U, I_shard, true, prep, pop_shard, users, n_users_blk, item_offset, n_items_local, d, hist_indptr, hist_indices,
                      hist_row_mode, K, head, early_stop, n_splits, out_keys, workspace, reinterpret_cast<hipStream_t>(stream));
}
