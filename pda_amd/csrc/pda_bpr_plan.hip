// The exact mini-batch SGD step WITHOUT atomics (round 3): plan-driven, two launches.
//
// Reference semantics (MF/model_api.py:83,102-121; [TF-ext] IndexedSlices): the gradients of a batch are computed from the
// tables as they stand, duplicates of a row are SUMMED, then the sum is applied.  The fused step of pda_bpr_step.hip applies
// every triplet's gradient with fp32 atomics the moment it is known -- 192 scalar atomics per triplet at d = 64, serialising on
// hot rows, and a workgroup may gather a row another workgroup has already moved (hogwild inside a batch).  No single launch
// can be exact without a grid-wide "every gather is done" point (measured in round 2: a device-scope barrier costs more than a
// launch on eight XCDs with incoherent L2s), so the exact step is two launches built around a PLAN of the batch:
//
//   plan     (pda_triplet_plan; by the sampler, batches ahead of the step, one workgroup per batch, all in LDS)
//            the 2B item references pos[0..B) ++ neg[0..B) sorted by item -> segments of equal item:
//            seg_item[s], seg_start[s..s+1], entries[i] = index into pos ++ neg; per triplet two bits "my positive / negative
//            is referenced once in this batch"; a flag "a user occurs twice" (the sampler never does that: rd.sample,
//            MF/train_new_api.py:380-381 -- such a batch is REJECTED: loss := NaN, tables untouched).
//   launch A (per triplet, d/4 lanes each: the forward pass of pda_bpr_step.hip) gathers the three rows, computes loss and the
//            two scalar coefficients g a_p, g a_n of the triplet, moves the USER row with one plain 16-byte store per lane
//            (users are distinct: nobody else reads or writes that row), and leaves the old user row (contiguous [B, d]) and
//            the coefficients in a scratch buffer.  No item row is written: every gather of the launch sees the old tables.
//   launch B (per segment = per distinct item row) sums coefficient x old user row over the segment's entries in plan order --
//            a fixed order: the step is bit-reproducible --, adds the L2 term (regs / batch_size) count row, and moves the row with one
//            plain store.  Segments longer than 8 entries (hot positives: ~190 of 2048 for the top item of a Zipf catalogue)
//            are summed by all lane groups of the workgroup through LDS.
//
// Traffic per triplet: A reads 3 rows, writes 2 (user row, scratch copy); B reads 2 scratch rows + (distinct rows / B) x (1 read
// + 1 write) -- about 9 row transfers against the algorithmic 6, none of them atomic.  bf16 tables (config 5): forward pass on
// the bf16 rows, update on the fp32 masters, and the touched bf16 rows are re-rounded by the same two launches (the fused path
// needs three more launches for that).
//
// exact = 0 (pda_bpr_step_plan_f32 only): ONE launch -- user rows and the item rows the plan marks as referenced once take plain
// stores, the shared item rows keep the atomics of the fused step (hogwild on those rows only).
#include <cstdlib>
#include "pda_common.h"
#include "pda_plan_common.h"

namespace {

#ifndef PDA_PLAN_LONGU
#define PDA_PLAN_LONGU 8      // launch B, long segments: entries per lane group and round trip
#endif
constexpr int kPlanMaxB = 4096;
__host__ __device__ inline int xl_max(int B) { return (2 * B) / kXlMin + 1; }
// (rounded up to whole f32x4: the multi-workgroup segments' partial sums behind it are stored and read as f32x4)
__host__ __device__ inline size_t scratch_floats_base(int B, int d) { return ((size_t)B * (size_t)(d + 2) + 2 * ((size_t)B / 8 + 8) + 3) & ~(size_t)3; }
__host__ __device__ inline int xl_cnt_floats(int B) { return (xl_max(B) + 3) & ~3; }        // the arrival counters in front of xl_part, 16-byte granular

struct PlanView {
    int* hdr;              // [4]: segments, "a user occurs twice", 2B, B
    int* seg_item;         // [2B]
    int* seg_start;        // [2B + 2]
    int* entries;          // [2B]  index into pos ++ neg, ascending inside a segment
    unsigned char* flags;  // [B]   bit 0: the positive is referenced once in the batch, bit 1: the negative
};
__host__ __device__ inline size_t plan_bytes(int B) {
    const size_t raw = 16 + (size_t)4 * (2 * B) + (size_t)4 * (2 * B + 2) + (size_t)4 * (2 * B) + (size_t)B;
    return (raw + 15) & ~(size_t)15;
}
__host__ __device__ inline PlanView plan_view(void* p, int B) {
    unsigned char* b = reinterpret_cast<unsigned char*>(p);
    PlanView v;
    v.hdr = reinterpret_cast<int*>(b);
    v.seg_item = v.hdr + 4;
    v.seg_start = v.seg_item + 2 * B;
    v.entries = v.seg_start + 2 * B + 2;
    v.flags = reinterpret_cast<unsigned char*>(v.entries + 2 * B);
    return v;
}

// One workgroup per batch.  Counting sort on a 12-bit hash of the item id (LDS atomics), then every reference ranks itself
// inside its bucket by (item, index) -- the scheme of group_by_pos_kernel (pda_bpr_step.hip), on pos ++ neg.
__global__ void __launch_bounds__(1024) plan_lds_kernel(const int32_t* __restrict__ users, const int32_t* __restrict__ pos,
                                                        const int32_t* __restrict__ neg, int B, unsigned char* __restrict__ plans,
                                                        size_t plan_stride) {
    constexpr int NBIN = 4096, NMAX = 2 * kPlanMaxB;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    int* cnt = reinterpret_cast<int*>(sm);                                   // [NBIN + 1] counts -> exclusive bases; later the flag words [B]
    uint64_t* member = reinterpret_cast<uint64_t*>(sm + 16448);             // [NMAX] (item << 32 | index) bucket by bucket
    uint64_t* sorted = member + NMAX;                                        // [NMAX]; before that: the users' hash set
    int* wsum = reinterpret_cast<int*>(sorted + NMAX);                       // [16]
    int* misc = wsum + 16;                                                   // [0] "a user occurs twice"
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t off = (size_t)blockIdx.x * B;
    users += off;
    pos += off;
    neg += off;
    PlanView pv = plan_view(plans + (size_t)blockIdx.x * plan_stride, B);
    const int N = 2 * B;

    // -- repeated users?  open addressing into 16384 slots (the `sorted` region)
    int* tab = reinterpret_cast<int*>(sorted);
    for (int i = tid; i < 16384; i += 1024) tab[i] = -1;
    for (int i = tid; i <= NBIN; i += 1024) cnt[i] = 0;
    if (tid == 0) misc[0] = 0;
    __syncthreads();
    for (int i = tid; i < B; i += 1024) {
        const int u = users[i];
        unsigned slot = ((unsigned)u * 2654435761u) >> 18;
        for (int probe = 0; probe < 16384; ++probe) {
            const int old = atomicCAS(&tab[slot], -1, u);
            if (old == -1) break;
            if (old == u) { misc[0] = 1; break; }
            slot = (slot + 1) & 16383u;
        }
    }
    // -- references into their buckets
    int bkt[8], arr[8];
    uint64_t mykey[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int e = tid + 1024 * k;
        bkt[k] = arr[k] = 0;
        mykey[k] = 0;
        if (e < N) {
            const int item = e < B ? pos[e] : neg[e - B];
            mykey[k] = ((uint64_t)(uint32_t)item << 32) | (uint32_t)e;
            bkt[k] = (int)(((uint32_t)item * 2654435761u) >> 20);
            arr[k] = atomicAdd(&cnt[bkt[k]], 1);
        }
    }
    __syncthreads();
    // exclusive scan of the 4096 counts: 4 consecutive bins per thread, wave scan, 16 wave totals
    int c4[4], run = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { c4[q] = cnt[4 * tid + q]; run += c4[q]; }
    int inc = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w];
    int ex = wbase + inc - run;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) { cnt[4 * tid + q] = ex; ex += c4[q]; }
    if (tid == 1023) cnt[NBIN] = ex;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (tid + 1024 * k < N) member[cnt[bkt[k]] + arr[k]] = mykey[k];
    __syncthreads();                     // (the users' hash set in `sorted` is dead from here on)
    int dst[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        dst[k] = -1;
        if (tid + 1024 * k < N) {
            const int lo = cnt[bkt[k]], hi = cnt[bkt[k] + 1];
            int rank = 0;
#pragma unroll 8
            for (int m = lo; m < hi; ++m) rank += member[m] < mykey[k] ? 1 : 0;
            dst[k] = lo + rank;
        }
    }
    __syncthreads();                     // every bucket base has been read: cnt becomes the flag words
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (dst[k] >= 0) sorted[dst[k]] = mykey[k];
    for (int i = tid; i < B; i += 1024) cnt[i] = 0;
    __syncthreads();
    // -- segments: thread t owns positions 8 t .. 8 t + 7
    int heads = 0;
    uint32_t hm = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = 8 * tid + k;
        if (i < N) {
            const uint32_t it = (uint32_t)(sorted[i] >> 32);
            const bool head = i == 0 || (uint32_t)(sorted[i - 1] >> 32) != it;
            if (head) { hm |= 1u << k; ++heads; }
        }
    }
    int hinc = heads;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(hinc, o, 64);
        if (lane >= o) hinc += v;
    }
    if (lane == 63) wsum[wave] = hinc;
    __syncthreads();
    int hbase = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) hbase += wsum[w];
        total += wsum[w];
    }
    int seg = hbase + hinc - heads;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = 8 * tid + k;
        if (i < N) {
            const uint64_t key = sorted[i];
            const uint32_t it = (uint32_t)(key >> 32), e = (uint32_t)key;
            pv.entries[i] = (int)e;
            if (hm & (1u << k)) {
                pv.seg_item[seg] = (int)it;
                pv.seg_start[seg] = i;
                ++seg;
                const bool single = i + 1 == N || (uint32_t)(sorted[i + 1] >> 32) != it;
                if (single) atomicOr(&cnt[e < (uint32_t)B ? e : e - B], e < (uint32_t)B ? 1 : 2);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < B; i += 1024) pv.flags[i] = (unsigned char)cnt[i];
    if (tid == 0) {
        pv.seg_start[total] = N;
        pv.hdr[0] = total;
        pv.hdr[1] = misc[0];
        pv.hdr[2] = N;
        pv.hdr[3] = B;
    }
}

struct PlanStepArgs {
    float* U;              // fp32 tables that take the update (the masters of bf16 tables)
    float* I;
    const void* Ufwd;      // tables of the forward pass: U / I themselves, or the bf16 rows
    const void* Ifwd;
    uint16_t* Ush;         // bf16 rows to re-round (bf16 tables), else NULL
    uint16_t* Ish;
    const int32_t* users;
    const int32_t* pos;
    const int32_t* neg;
    const float* pos_pop;
    const float* neg_pop;
    const int* hdr;
    const int* seg_item;
    const int* seg_start;
    const int* entries;
    const unsigned char* flags;
    float* scratch;        // [B][D] old user rows, then [B][2] coefficients (g a_p, g a_n)
    float* loss_acc;
    int B;
    float inv_B, reg_c, lr;
    int exact;
    int strided;           // launch B: lane group g of workgroup w takes segment g W + w (W workgroups with work) instead of w G + g
};

__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
__device__ __forceinline__ void atomic_add4(float* p, f32x4 v) {
    unsafeAtomicAdd(p + 0, v[0]);
    unsafeAtomicAdd(p + 1, v[1]);
    unsafeAtomicAdd(p + 2, v[2]);
    unsafeAtomicAdd(p + 3, v[3]);
}
__device__ __forceinline__ uint32_t rne16(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void store_bf16x4(uint16_t* p, f32x4 v) {
    uint2 o;
    o.x = rne16(v[0]) | (rne16(v[1]) << 16);
    o.y = rne16(v[2]) | (rne16(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = o;
}

// launch A: one triplet per D/4 lanes
template <int D, bool BF>
__global__ void __launch_bounds__(512) plan_triplets_kernel(PlanStepArgs a) {
    constexpr int L = D / 4, TPB = 512 / L;
    __shared__ float red[2][8];
    __shared__ int s_pos[TPB];
    __shared__ __attribute__((aligned(16))) float s_dpe[TPB * D];
    const int tid = threadIdx.x, g = tid / L, e = tid % L;
    const int t = (int)blockIdx.x * TPB + g;
    const bool active = t < a.B;
    const bool with_pop = a.pos_pop != nullptr;
    const bool rejected = a.hdr[1] != 0;             // a user occurs twice: nothing is written, the loss becomes NaN
    float maxi = 0.f, sq = 0.f;
    int p = -1;
    bool shared_pos = false;
    float* ptarget = nullptr;
    if (active && !rejected) {
        const int u = a.users[t], n = a.neg[t];
        p = a.pos[t];
        const f32x4 ue = pda_load4<BF>(a.Ufwd, (size_t)u * D + 4 * e);
        const f32x4 pe = pda_load4<BF>(a.Ifwd, (size_t)p * D + 4 * e);
        const f32x4 ne = pda_load4<BF>(a.Ifwd, (size_t)n * D + 4 * e);
        f32x4 um = ue, pm = pe, nm = ne;            // the rows that take the update
        const unsigned fl = a.flags[t];
        if constexpr (BF) {
            um = *reinterpret_cast<const f32x4*>(a.U + (size_t)u * D + 4 * e);
            if (!a.exact) {
                pm = *reinterpret_cast<const f32x4*>(a.I + (size_t)p * D + 4 * e);
                nm = *reinterpret_cast<const f32x4*>(a.I + (size_t)n * D + 4 * e);
            }
        }
        float ps = dot4(ue, pe), ns = dot4(ue, ne);
        sq = dot4(ue, ue) + dot4(pe, pe) + dot4(ne, ne);
#pragma unroll
        for (int o = L / 2; o > 0; o >>= 1) {
            ps += __shfl_xor(ps, o, 64);
            ns += __shfl_xor(ns, o, 64);
        }
        float ap = 1.f, an = 1.f, psw = ps, nsw = ns;
        if (with_pop) {
            const float qp = a.pos_pop[t], qn = a.neg_pop[t];
            const float ep = ps > 0.f ? 1.f : expf(ps);   // d(elu+1)/dx  [TF-ext EluGrad]
            const float en = ns > 0.f ? 1.f : expf(ns);
            psw = (ps > 0.f ? ps + 1.f : ep) * qp;        // (elu(ps)+1)*pos_pop   MF/model_api.py:107,109
            nsw = (ns > 0.f ? ns + 1.f : en) * qn;        // :108,110
            ap = qp * ep;
            an = qn * en;
        }
        const float x = psw - nsw;
        const float sg = 1.f / (1.f + expf(-x));
        if (e == 0) maxi = logf(sg + 1e-10f);             // :112 / :702
        const float gg = -a.inv_B * sg * (1.f - sg) / (sg + 1e-10f);
        const float gp = gg * ap, gn = gg * an, c = a.reg_c, nlr = -a.lr;
        f32x4 due, dpe, dne;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            due[k] = gp * pe[k] - gn * ne[k] + c * ue[k];
            dpe[k] = gp * ue[k] + c * pe[k];
            dne[k] = -gn * ue[k] + c * ne[k];
        }
        // the user row: distinct inside the batch, one plain store
        const f32x4 un = um + due * nlr;
        *reinterpret_cast<f32x4*>(a.U + (size_t)u * D + 4 * e) = un;
        if constexpr (BF) store_bf16x4(a.Ush + (size_t)u * D + 4 * e, un);
        if (a.exact) {
            *reinterpret_cast<f32x4*>(a.scratch + (size_t)t * D + 4 * e) = ue;
            if (e == 0) *reinterpret_cast<float2*>(a.scratch + (size_t)a.B * D + 2 * (size_t)t) = make_float2(gp, gn);
        } else {
            // one launch: rows referenced once take plain stores, shared rows the atomics of the fused step
            if (fl & 2u) {
                const f32x4 nn = nm + dne * nlr;
                *reinterpret_cast<f32x4*>(a.I + (size_t)n * D + 4 * e) = nn;
            } else {
                atomic_add4(a.I + (size_t)n * D + 4 * e, dne * nlr);
            }
            if (fl & 1u) {
                const f32x4 pn = pm + dpe * nlr;
                *reinterpret_cast<f32x4*>(a.I + (size_t)p * D + 4 * e) = pn;
            } else {
                shared_pos = true;
                ptarget = a.I + (size_t)p * D + 4 * e;
                *reinterpret_cast<f32x4*>(s_dpe + g * D + 4 * e) = dpe * nlr;
            }
        }
    }
    if (!a.exact) {
        if (e == 0) s_pos[g] = shared_pos ? p : -1 - g;       // (distinct dummies: never equal to each other or to an item)
        __syncthreads();
        if (shared_pos) {
            // the first triplet of the workgroup with this positive sums all the workgroup's contributions to it
            bool leader = true;
            for (int k = 0; k < g; ++k) leader = leader && (s_pos[k] != p);
            if (leader) {
                f32x4 sum = *reinterpret_cast<const f32x4*>(s_dpe + g * D + 4 * e);
                for (int k = g + 1; k < TPB; ++k)
                    if (s_pos[k] == p) sum += *reinterpret_cast<const f32x4*>(s_dpe + k * D + 4 * e);
                atomic_add4(ptarget, sum);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        maxi += __shfl_xor(maxi, o, 64);
        sq += __shfl_xor(sq, o, 64);
    }
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) {
        red[0][wave] = maxi;
        red[1][wave] = sq;
    }
    __syncthreads();
    if (tid == 0 && (a.loss_acc || a.exact)) {
        float sm = 0.f, ss = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            sm += red[0][w];
            ss += red[1][w];
        }
        float mf = -sm * a.inv_B;                   // -mean(maxi)          :114 / :704
        float rg = a.reg_c * 0.5f * ss;             // regs * l2 / batch    :117-120
        if (rejected) mf = rg = __int_as_float(0x7FC00000);
        if (a.exact && blockIdx.x == 0 && a.strided) {          // (launch B's arrival counters of the very long segments)
            int* xc = reinterpret_cast<int*>(a.scratch + scratch_floats_base(a.B, D));
            for (int q = 0; q < xl_max(a.B); ++q) xc[q] = 0;
        }
        if (a.exact) {
            // two plain stores per workgroup; launch B adds them up (64 workgroups x 3 float atomics on the same three words were
            // a visible share of this latency-bound launch)
            float* part = a.scratch + (size_t)a.B * (D + 2) + 2 * (size_t)blockIdx.x;
            part[0] = mf;
            part[1] = rg;
        } else {
            unsafeAtomicAdd(a.loss_acc + 0, mf + rg);
            unsafeAtomicAdd(a.loss_acc + 1, mf);
            unsafeAtomicAdd(a.loss_acc + 2, rg);
        }
    }
}

// launch B, extra workgroups (large batches): workgroup j sums piece j % kXlPieces of very long segment j / kXlPieces; the last of a
// segment's kXlPieces workgroups to arrive adds the partial sums in piece order and stores the row.
template <int D, bool BF>
__device__ __forceinline__ void plan_items_xl(const PlanStepArgs& a, int j, float* s_part) {
    constexpr int L = D / 4, G = 256 / L, LONGU = PDA_PLAN_LONGU, STEP = G * LONGU;
    __shared__ int s_last;
    const int tid = threadIdx.x, g = tid / L, e = tid % L;
    const int B = a.B;
    const int k = j / kXlPieces, piece = j % kXlPieces;
    if (a.hdr[1] != 0) return;                                      // rejected batch
    const int n_xl = a.seg_start[2 * B + 1];
    if (k >= n_xl) return;
    const int sidx = a.seg_item[2 * B - 1 - k];                       // (the list of very long segments grows from the end of seg_item)
    const int x = a.seg_item[sidx], b0 = a.seg_start[sidx], b1 = a.seg_start[sidx + 1];
    const int len = b1 - b0;
    const int r0 = b0 + (int)((long long)len * piece / kXlPieces), r1 = b0 + (int)((long long)len * (piece + 1) / kXlPieces);
    const float* __restrict__ coef = a.scratch + (size_t)B * D;
    const float* __restrict__ rows_old = a.scratch;
    const int* __restrict__ entries = a.entries;
    int* xl_cnt = reinterpret_cast<int*>(a.scratch + scratch_floats_base(B, D));
    float* xl_part = a.scratch + scratch_floats_base(B, D) + xl_cnt_floats(B);
    f32x4 part = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = r0 + g; i0 < r1; i0 += STEP) {
        int en2[LONGU];
#pragma unroll
        for (int q = 0; q < LONGU; ++q) en2[q] = entries[min(i0 + q * G, 2 * B - 1)];
        float w2[LONGU];
        f32x4 u2[LONGU];
#pragma unroll
        for (int q = 0; q < LONGU; ++q) {
            const bool on = i0 + q * G < r1;
            const int t = en2[q] < B ? en2[q] : en2[q] - B;
            const float2 co = *reinterpret_cast<const float2*>(coef + 2 * (size_t)t);
            w2[q] = on ? (en2[q] < B ? co.x : -co.y) : 0.f;
            u2[q] = *reinterpret_cast<const f32x4*>(rows_old + (size_t)t * D + 4 * e);
        }
#pragma unroll
        for (int q = 0; q < LONGU; ++q) part += u2[q] * w2[q];
    }
    *reinterpret_cast<f32x4*>(s_part + g * D + 4 * e) = part;
    __syncthreads();
    if (g == 0) {
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < G; ++q) sum += *reinterpret_cast<const f32x4*>(s_part + q * D + 4 * e);       // group order
        *reinterpret_cast<f32x4*>(xl_part + ((size_t)k * kXlPieces + piece) * D + 4 * e) = sum;
    }
    __threadfence();                                                // the partial row is visible device-wide before the count
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&xl_cnt[k], 1) == kXlPieces - 1 ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    __threadfence();                                                // (acquire: the other workgroups' partial rows)
    if (g == 0) {
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < kXlPieces; ++q)
            sum += *reinterpret_cast<const volatile f32x4*>(xl_part + ((size_t)k * kXlPieces + q) * D + 4 * e);      // piece order
        const f32x4 row = pda_load4<BF>(a.Ifwd, (size_t)x * D + 4 * e);
        f32x4 old = row;
        if constexpr (BF) old = *reinterpret_cast<const f32x4*>(a.I + (size_t)x * D + 4 * e);
        const float cc = a.reg_c * (float)len;
        const f32x4 nw = old - (sum + row * cc) * a.lr;
        *reinterpret_cast<f32x4*>(a.I + (size_t)x * D + 4 * e) = nw;
        if constexpr (BF) store_bf16x4(a.Ish + (size_t)x * D + 4 * e, nw);
    }
}

// launch B: one segment (one distinct item row) per D/4 lanes.  The launch is a chain of dependent loads -- segment -> entries ->
// coefficients and old user rows -> the row's store --, so every level is issued for ALL of a lane group's entries at once
// (branch-free, clamped indices): three round trips for a segment of up to 8 entries, one more per 4 further entries and group.
template <int D, bool BF>
__global__ void __launch_bounds__(256) plan_items_kernel(PlanStepArgs a, int n_parts, int n_normal) {
    constexpr int L = D / 4, G = 256 / L, SHORT = 8, LONGU = PDA_PLAN_LONGU;
    __shared__ __attribute__((aligned(16))) float s_part[G * D];
    __shared__ int s_long[G];
    __shared__ int s_nlong;
    const int tid = threadIdx.x, g = tid / L, e = tid % L;
    if ((int)blockIdx.x >= n_normal) {
        plan_items_xl<D, BF>(a, (int)blockIdx.x - n_normal, s_part);
        return;
    }
    // Which segment.  Small batches (LDS plan: segments in hash-bucket order): w G + g, known without a load.  Large batches (plan
    // sorted by item id: the hot items -- neighbours wherever ids follow popularity -- would all fall to the first workgroups, each
    // long segment a serial loop of its workgroup: launch B 80 of a 91 us step at B = 32 768): g W + w, neighbours apart.
    const int B = a.B;
    int s = (int)blockIdx.x * G + g, n_wg = 0;
    if (a.strided) {
        n_wg = (a.hdr[0] + G - 1) / G;
        s = g * n_wg + (int)blockIdx.x;
        if (s >= 2 * B) s = 2 * B - 1;             // (the loads below are in bounds whatever they return)
    }
    const float* __restrict__ coef = a.scratch + (size_t)B * D;
    const float* __restrict__ rows_old = a.scratch;
    const int* __restrict__ entries = a.entries;
    // level 1: header, segment bounds (in-bounds for every s < 2B whatever they hold)
    const int rejected = a.hdr[1], n_seg = a.hdr[0];
    const int x_raw = a.seg_item[s], b0_raw = a.seg_start[s], b1_raw = a.seg_start[s + 1];
    if (blockIdx.x == 0 && a.loss_acc != nullptr && tid < 64) {
        // the loss of the batch: launch A left one (mf, reg) pair per workgroup
        float mf = 0.f, rg = 0.f;
        const float* part = a.scratch + (size_t)B * (D + 2);
        for (int q = tid; q < n_parts; q += 64) {
            mf += part[2 * q];
            rg += part[2 * q + 1];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mf += __shfl_xor(mf, o, 64);
            rg += __shfl_xor(rg, o, 64);
        }
        if (tid == 0) {           // (one writer per launch, launches of a stream in order: plain adds would do; atomics keep other streams safe)
            unsafeAtomicAdd(a.loss_acc + 0, mf + rg);
            unsafeAtomicAdd(a.loss_acc + 1, mf);
            unsafeAtomicAdd(a.loss_acc + 2, rg);
        }
    }
    if (rejected != 0) return;                        // rejected batch (uniform over the grid)
    if (a.strided ? (int)blockIdx.x >= n_wg : (int)blockIdx.x * G >= n_seg) return;
    if (tid == 0) s_nlong = 0;
    __syncthreads();
    const bool have = (a.strided ? g * n_wg + (int)blockIdx.x : s) < n_seg;
    const int x = have ? x_raw : 0, b0 = have ? b0_raw : 0, b1 = have ? b1_raw : 0;
    // level 2: the first SHORT entries and the row itself
    int en[SHORT];
#pragma unroll
    for (int i = 0; i < SHORT; ++i) en[i] = entries[min(b0 + i, 2 * B - 1)];
    const f32x4 row = pda_load4<BF>(a.Ifwd, (size_t)x * D + 4 * e);
    f32x4 old = row;
    if constexpr (BF) old = *reinterpret_cast<const f32x4*>(a.I + (size_t)x * D + 4 * e);
    // level 3: their coefficients and old user rows
    float w[SHORT];
    f32x4 ur[SHORT];
#pragma unroll
    for (int i = 0; i < SHORT; ++i) {
        const bool on = b0 + i < b1;
        const int t = en[i] < B ? en[i] : en[i] - B;
        const float2 co = *reinterpret_cast<const float2*>(coef + 2 * (size_t)t);
        w[i] = on ? (en[i] < B ? co.x : -co.y) : 0.f;
        ur[i] = *reinterpret_cast<const f32x4*>(rows_old + (size_t)t * D + 4 * e);
    }
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < SHORT; ++i) sum += ur[i] * w[i];          // plan order: bit-reproducible
    const bool is_xl = a.strided != 0 && have && b1 - b0 >= kXlMin;          // (summed and stored by the extra workgroups of the launch)
    if (have && !is_xl && b1 - b0 > SHORT && e == 0) s_long[atomicAdd(&s_nlong, 1)] = g;
    __syncthreads();
    const int nl = s_nlong;
    for (int k = 0; k < nl; ++k) {
        // a long segment: every lane group takes every G-th of its remaining entries, LONGU of them per round trip; the owner
        // adds the G partial sums in group order (which long segment comes first does not matter: each sum is formed the same way)
        const int og = s_long[k];
        const int so = a.strided ? og * n_wg + (int)blockIdx.x : (int)blockIdx.x * G + og;
        const int ob0 = a.seg_start[so] + SHORT, ob1 = a.seg_start[so + 1];
        f32x4 part = {0.f, 0.f, 0.f, 0.f};
        // Two dependent loads per round (the entry, then its coefficient and old user row): pipelined over the rounds -- the rows
        // of round r + 1 and the entries of round r + 2 are requested before round r is summed.  (Unpipelined, the one 3 000-entry
        // segment of a Zipf batch of 32 768 triplets was 70 of the step's 90 us: 24 rounds of two round trips in one workgroup.)
        constexpr int STEP = G * LONGU;
        auto load_entries = [&](int i0, int (&en2)[LONGU]) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < LONGU; ++q) en2[q] = entries[min(i0 + q * G, 2 * B - 1)];
        };
        auto load_rows = [&](int i0, const int (&en2)[LONGU], float (&w2)[LONGU], f32x4 (&u2)[LONGU]) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < LONGU; ++q) {
                const bool on = i0 + q * G < ob1;
                const int t = en2[q] < B ? en2[q] : en2[q] - B;
                const float2 co = *reinterpret_cast<const float2*>(coef + 2 * (size_t)t);
                w2[q] = on ? (en2[q] < B ? co.x : -co.y) : 0.f;
                u2[q] = *reinterpret_cast<const f32x4*>(rows_old + (size_t)t * D + 4 * e);
            }
        };
        {
            // (two buffer sets used in turn: a register copy from one to the other would wait for the very loads it is meant to hide)
            int i0 = ob0 + g;
            int enA[LONGU], enB[LONGU];
            float wA[LONGU], wB[LONGU];
            f32x4 uA[LONGU], uB[LONGU];
            if (i0 < ob1) {
                load_entries(i0, enA);
                load_rows(i0, enA, wA, uA);                          // round 0
                load_entries(i0 + STEP, enB);                        // entries of round 1
            }
            while (i0 < ob1) {
                if (i0 + STEP < ob1) {
                    load_rows(i0 + STEP, enB, wB, uB);               // rows of the next round
                    load_entries(i0 + 2 * STEP, enA);                // entries of the one after
                }
#pragma unroll
                for (int q = 0; q < LONGU; ++q) part += uA[q] * wA[q];      // (round order, as before: the sum is unchanged)
                i0 += STEP;
                if (i0 >= ob1) break;
                if (i0 + STEP < ob1) {
                    load_rows(i0 + STEP, enA, wA, uA);
                    load_entries(i0 + 2 * STEP, enB);
                }
#pragma unroll
                for (int q = 0; q < LONGU; ++q) part += uB[q] * wB[q];
                i0 += STEP;
            }
        }
        *reinterpret_cast<f32x4*>(s_part + g * D + 4 * e) = part;
        __syncthreads();
        if (g == og) {
            for (int q = 0; q < G; ++q) sum += *reinterpret_cast<const f32x4*>(s_part + q * D + 4 * e);
        }
        __syncthreads();
    }
    if (have && !is_xl) {
        const float cc = a.reg_c * (float)(b1 - b0);
        const f32x4 nw = old - (sum + row * cc) * a.lr;
        *reinterpret_cast<f32x4*>(a.I + (size_t)x * D + 4 * e) = nw;
        if constexpr (BF) store_bf16x4(a.Ish + (size_t)x * D + 4 * e, nw);
    }
}

template <int D, bool BF>
int launch_plan_step(const PlanStepArgs& a, hipStream_t s) {
    constexpr int TPB = 512 / (D / 4), G = 256 / (D / 4);
#ifndef PDA_PLAN_ONLY
#define PDA_PLAN_ONLY 0       // timing experiments: 1 = launch A only, 2 = launch B only (results are wrong)
#endif
    if (PDA_PLAN_ONLY != 2) hipLaunchKernelGGL((plan_triplets_kernel<D, BF>), dim3((unsigned)((a.B + TPB - 1) / TPB)), dim3(512), 0, s, a);
    PDA_CHECK_LAUNCH();
    if (a.exact && PDA_PLAN_ONLY != 1) {
        const int n_normal = (2 * a.B + G - 1) / G, n_xl_wg = a.strided ? xl_max(a.B) * kXlPieces : 0;
        hipLaunchKernelGGL((plan_items_kernel<D, BF>), dim3((unsigned)(n_normal + n_xl_wg)), dim3(256), 0, s, a, (a.B + TPB - 1) / TPB, n_normal);
        PDA_CHECK_LAUNCH();
    }
    return PDA_OK;
}

int run_plan_step(float* U, float* I, const void* Ufwd, const void* Ifwd, uint16_t* Ush, uint16_t* Ish, bool bf, const int32_t* users,
                  const int32_t* pos, const int32_t* neg, const float* pos_pop, const float* neg_pop, int B, int d, float regs, float reg_div,
                  float lr, const void* plan, float* scratch, int exact, float* loss_acc, hipStream_t s) {
    if (!U || !I || !users || !pos || !neg || !plan || B <= 0 || B > (1 << 24) || reg_div <= 0.f) return PDA_ERR_ARG;     // (plans of B > 4096: pda_triplet_plan_large)
    if ((pos_pop == nullptr) != (neg_pop == nullptr)) return PDA_ERR_ARG;
    if (exact && !scratch) return PDA_ERR_ARG;
    if (!exact && bf) return PDA_ERR_UNSUPPORTED;
    const PlanView pv = plan_view(const_cast<void*>(plan), B);
    const PlanStepArgs a{U, I, Ufwd, Ifwd, Ush, Ish, users, pos, neg, pos_pop, neg_pop, pv.hdr, pv.seg_item, pv.seg_start, pv.entries, pv.flags,
                         scratch, loss_acc, B, 1.0f / (float)B, regs / reg_div, lr, exact ? 1 : 0, B > kPlanMaxB ? 1 : 0};
    switch (d) {
        case 32: return bf ? launch_plan_step<32, true>(a, s) : launch_plan_step<32, false>(a, s);
        case 64: return bf ? launch_plan_step<64, true>(a, s) : launch_plan_step<64, false>(a, s);
        case 128: return bf ? launch_plan_step<128, true>(a, s) : launch_plan_step<128, false>(a, s);
        case 256: return bf ? launch_plan_step<256, true>(a, s) : launch_plan_step<256, false>(a, s);
        default: return PDA_ERR_UNSUPPORTED;
    }
}

}  // namespace

extern "C" size_t pda_triplet_plan_bytes(int B) { return B > 0 ? plan_bytes(B) : 0; }
// old user rows [B][d], coefficients [B][2], one (mf, reg) pair per workgroup of launch A (at most B / 8 + 1 of them: d = 256)
extern "C" size_t pda_bpr_step_plan_scratch_bytes(int B, int d) {
    // ... then (large batches) one counter and kXlPieces partial rows per very long segment
    return (B > 0 && d > 0) ? (scratch_floats_base(B, d) + (size_t)xl_cnt_floats(B) + (size_t)xl_max(B) * (size_t)kXlPieces * d) * 4 : 0;
}

extern "C" int pda_triplet_plan(const int32_t* users, const int32_t* pos, const int32_t* neg, int B, int n_batches, void* plans, void* stream) {
    if (!users || !pos || !neg || !plans || B <= 0 || n_batches <= 0) return PDA_ERR_ARG;
    if (B > kPlanMaxB) return PDA_ERR_UNSUPPORTED;
    constexpr int smem = 16448 + 2 * (2 * kPlanMaxB) * 8 + 64 + 64;
    static int attr_set = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&plan_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return PDA_ERR_LAUNCH;
        attr_set = 1;
    }
    hipLaunchKernelGGL(plan_lds_kernel, dim3((unsigned)n_batches), dim3(1024), smem, reinterpret_cast<hipStream_t>(stream), users, pos, neg, B,
                       reinterpret_cast<unsigned char*>(plans), plan_bytes(B));
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

extern "C" int pda_bpr_step_plan_f32(float* U, float* I, const int32_t* users, const int32_t* pos, const int32_t* neg, const float* pos_pop,
                                     const float* neg_pop, int B, int d, float regs, float reg_div, float lr, const void* plan, float* scratch,
                                     int exact, float* loss_acc, void* stream) {
    return run_plan_step(U, I, U, I, nullptr, nullptr, false, users, pos, neg, pos_pop, neg_pop, B, d, regs, reg_div, lr, plan, scratch, exact,
                         loss_acc, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pda_bpr_step_plan_bf16(uint16_t* U_bf16, uint16_t* I_bf16, float* U_master, float* I_master, const int32_t* users,
                                      const int32_t* pos, const int32_t* neg, const float* pos_pop, const float* neg_pop, int B, int d,
                                      float regs, float reg_div, float lr, const void* plan, float* scratch, float* loss_acc, void* stream) {
    if (!U_bf16 || !I_bf16) return PDA_ERR_ARG;
    return run_plan_step(U_master, I_master, U_bf16, I_bf16, U_bf16, I_bf16, true, users, pos, neg, pos_pop, neg_pop, B, d, regs, reg_div, lr, plan,
                         scratch, 1, loss_acc, reinterpret_cast<hipStream_t>(stream));
}
