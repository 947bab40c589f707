// Fused BPR triplet step for MI355X (gfx950): gather -> dots -> (ELU+1)*pop -> log-sigmoid loss + L2 ->
// closed-form gradients -> update, in one launch.
//
// Replaces the per-step TF graph of MF/model_api.py:51-53 (gather), :102-110 / :695-697 (scores),
// :112-121 / :699-705 (loss), the implicit gradient of `minimize` (:83, :471) and -- in the SGD mode
// the north_star asks for -- the parameter update.  The reference optimiser is Adam with dense decay
// [TF-ext]; that mode is PDA_UPD_DENSE_GRAD followed by pda_adam_dense_sweep_f32 on both tables.
//
// This path is HBM/latency bound (about 6*d*4 B per triplet, ~10*d flop): no MFMA.  Layout: d/4 lanes
// per triplet, each lane owns one float4 of the three gathered rows (a 4*d-byte row is one fully
// coalesced segment), dots by xor-shuffle inside the lane group, per-block loss reduction -> 3 atomics.
#include <cmath>
#include <cstdlib>
#include "pda_common.h"
#include "pda_sample.h"

namespace {

struct StepArgs {
    float* U;
    float* I;
    const int32_t* users;
    const int32_t* pos;
    const int32_t* neg;
    const float* pos_pop;
    const float* neg_pop;
    float* g_user;
    float* g_pos;
    float* g_neg;
    float* gU;
    float* gI;
    float* loss_acc;
    int B;
    float inv_B;
    float reg_c;  // regs / reg_div
    float lr;
    int mode;
    int item_offset;   // pos / neg are global item ids; row = id - item_offset in `I` (an item shard)
    int g_stride;      // floats between consecutive g_user rows (>= D)
    const void* Ufwd;  // tables the forward pass gathers from: U / I themselves (fp32) or their bf16 shadows
    const void* Ifwd;  // (pda_bpr_step_bf16: U / I are then the fp32 masters that take the update)
    int any_order;     // PDA_UPD_ANY_ORDER: equal positives are combined wherever they sit in the workgroup
    int users_distinct;  // PDA_UPD_USERS_DISTINCT: no user id occurs twice in the batch -- its row takes a plain store
    int32_t* tag_u = nullptr;   // pda_adam_step_f32: "row touched by step `tag`" words (one int32 per table row) that replace the bitmaps +
    int32_t* tag_i = nullptr;   // their memsets of pda_adam_mark_rows / pda_adam_dense_sweep3_f32: the step kernel itself leaves the tag
    int tag = 0;
};

__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

__device__ __forceinline__ void atomic_add4(float* p, f32x4 v) {
    unsafeAtomicAdd(p + 0, v[0]);
    unsafeAtomicAdd(p + 1, v[1]);
    unsafeAtomicAdd(p + 2, v[2]);
    unsafeAtomicAdd(p + 3, v[3]);
}

// 512 threads = 512/(D/4) triplets per block.  Positive items are popularity-skewed (the whole point of PDA): with Zipf
// data one batch holds the hottest item ~170 times, and 170 x 64 float atomics on the same two cache lines serialise in
// L2 (measured 25 us per 2048-triplet step).  The positives' contributions therefore go through LDS first: runs of
// equal `pos` inside the block are summed by their first triplet and leave as ONE atomic per element.  Any batch order
// is correct; a batch sorted by `pos` (pda_sort_triplets_by_pos, done by the device sampler) makes the runs long.
// COH (pda_bpr_train_steps_f32): the batch and the table rows were written by OTHER workgroups of this launch -- by the sampler
// and by the previous step's atomics, both at device scope -- and are read with device-scope loads (they miss the caches that
// are not coherent across the XCDs), so that a grid barrier needs no cache invalidation.
template <bool COH, typename T>
__device__ __forceinline__ T in_load(const T* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH, bool BF>
__device__ __forceinline__ f32x4 row_load4(const void* base, size_t idx) {
    if constexpr (COH && !BF) {
        const uint64_t* q = reinterpret_cast<const uint64_t*>(reinterpret_cast<const float*>(base) + idx);
        const uint64_t lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f32x4 v = {__uint_as_float((uint32_t)lo), __uint_as_float((uint32_t)(lo >> 32)), __uint_as_float((uint32_t)hi),
                   __uint_as_float((uint32_t)(hi >> 32))};
        return v;
    } else {
        return pda_load4<BF>(base, idx);
    }
}

template <int D, bool BF, bool COH = false>
__device__ __forceinline__ void bpr_step_body(const StepArgs& a, const int bid) {
    constexpr int L = D / 4;        // lanes per triplet
    constexpr int TPB = 512 / L;    // triplets per block
    __shared__ float red[2][8];
    __shared__ int s_pos[TPB];
    __shared__ __attribute__((aligned(16))) float s_dpe[TPB * D];
    const int tid = threadIdx.x, g = tid / L, e = tid % L;
    const int t = bid * TPB + g;
    const bool active = t < a.B;
    const bool with_pop = a.pos_pop != nullptr;
    const bool scatter = a.mode == PDA_UPD_SGD_FUSED || a.mode == PDA_UPD_DENSE_GRAD || a.mode == PDA_UPD_SGD_ITEMS ||
                         a.mode == PDA_UPD_DENSE_ITEMS;

    float maxi = 0.f, sq = 0.f;
    int p = -1;
    float* ptarget = nullptr;
    if (active) {
        const int u = in_load<COH>(&a.users[t]), n = in_load<COH>(&a.neg[t]) - a.item_offset;
        p = in_load<COH>(&a.pos[t]) - a.item_offset;
        float* up = a.U + (size_t)u * D + 4 * e;
        float* pp = a.I + (size_t)p * D + 4 * e;
        float* np_ = a.I + (size_t)n * D + 4 * e;
        const f32x4 ue = row_load4<COH, BF>(a.Ufwd, (size_t)u * D + 4 * e);
        const f32x4 pe = row_load4<COH, BF>(a.Ifwd, (size_t)p * D + 4 * e);
        const f32x4 ne = row_load4<COH, BF>(a.Ifwd, (size_t)n * D + 4 * e);
        float ps = dot4(ue, pe), ns = dot4(ue, ne);
        sq = dot4(ue, ue) + dot4(pe, pe) + dot4(ne, ne);
#pragma unroll
        for (int o = L / 2; o > 0; o >>= 1) {
            ps += __shfl_xor(ps, o, 64);
            ns += __shfl_xor(ns, o, 64);
        }
        float ap = 1.f, an = 1.f, psw = ps, nsw = ns;
        if (with_pop) {
            const float qp = in_load<COH>(&a.pos_pop[t]), qn = in_load<COH>(&a.neg_pop[t]);
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
        const float gp = gg * ap, gn = gg * an, c = a.reg_c;
        f32x4 due, dpe, dne;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            due[k] = gp * pe[k] - gn * ne[k] + c * ue[k];
            dpe[k] = gp * ue[k] + c * pe[k];
            dne[k] = -gn * ue[k] + c * ne[k];
        }
        const f32x4 dpe_raw = dpe;
        if (a.mode == PDA_UPD_SGD_FUSED) {
            const float nlr = -a.lr;
            // users are unique per batch in the reference's sampler (MF/train_new_api.py:380-381): when the caller says so
            // (PDA_UPD_USERS_DISTINCT) the row this triplet read is the row it writes, no atomics; otherwise atomics keep
            // B > n_users (and any other sampler) safe
            if (a.users_distinct && !BF && !COH) *reinterpret_cast<f32x4*>(up) = ue + due * nlr;
            else atomic_add4(up, due * nlr);
            atomic_add4(np_, dne * nlr);  // negatives are uniform over the catalogue: duplicates are rare
            dpe = dpe * nlr;
            ptarget = pp;
        } else if (a.mode == PDA_UPD_SGD_ITEMS) {
            // item-parallel training: this rank's item rows take their update here; the user gradient leaves through
            // g_user for the exchange (every rank then applies all of them: pda_apply_user_grads_f32)
            const float nlr = -a.lr;
            atomic_add4(np_, dne * nlr);
            dpe = dpe * nlr;
            ptarget = pp;
        } else if (a.mode == PDA_UPD_DENSE_ITEMS) {
            // item-parallel Adam: item gradients summed into the shard's dense accumulator, user gradient out
            atomic_add4(a.gI + (size_t)n * D + 4 * e, dne);
            ptarget = a.gI + (size_t)p * D + 4 * e;
        } else if (a.mode == PDA_UPD_DENSE_GRAD) {
            // (tagged step + distinct users: gU is zero off the rows the sweep clears behind itself, the row has one writer -- a plain store)
            if (a.tag_u && a.users_distinct && !COH) *reinterpret_cast<f32x4*>(a.gU + (size_t)u * D + 4 * e) = due;
            else atomic_add4(a.gU + (size_t)u * D + 4 * e, due);
            atomic_add4(a.gI + (size_t)n * D + 4 * e, dne);
            ptarget = a.gI + (size_t)p * D + 4 * e;
            if (a.tag_u && e == 0) {            // same value from every writer of a row: plain stores
                a.tag_u[u] = a.tag;
                a.tag_i[p] = a.tag;
                a.tag_i[n] = a.tag;
            }
        }
        if (scatter) *reinterpret_cast<f32x4*>(s_dpe + g * D + 4 * e) = dpe;
        if (a.g_user) *reinterpret_cast<f32x4*>(a.g_user + (size_t)t * a.g_stride + 4 * e) = due;
        if (a.g_pos) {
            *reinterpret_cast<f32x4*>(a.g_pos + (size_t)t * D + 4 * e) = dpe_raw;
            *reinterpret_cast<f32x4*>(a.g_neg + (size_t)t * D + 4 * e) = dne;
        }
    }
    if (e == 0) s_pos[g] = p;
    __syncthreads();
    if (scatter && active && a.any_order) {
        // batch in sampling order: the first triplet of the workgroup with this positive sums ALL the workgroup's
        // contributions to it -- adjacent or not -- and leaves as one atomic per element (hot item: one atomic per
        // workgroup instead of one per occurrence; 24.8 -> 12.9 us per 2048-triplet step at C2 without any sort)
        bool leader = true;
        for (int k = 0; k < g; ++k) leader = leader && (s_pos[k] != p);
        if (leader) {
            f32x4 sum = *reinterpret_cast<const f32x4*>(s_dpe + g * D + 4 * e);
            for (int k = g + 1; k < TPB; ++k)
                if (s_pos[k] == p) sum += *reinterpret_cast<const f32x4*>(s_dpe + k * D + 4 * e);
            atomic_add4(ptarget, sum);
        }
    } else if (scatter && active && (g == 0 || s_pos[g - 1] != p)) {   // grouped batch: first triplet of a run of equal positives
        f32x4 sum = *reinterpret_cast<const f32x4*>(s_dpe + g * D + 4 * e);
        for (int k = g + 1; k < TPB && s_pos[k] == p; ++k) sum += *reinterpret_cast<const f32x4*>(s_dpe + k * D + 4 * e);
        atomic_add4(ptarget, sum);
    }
    // block reduction of sum(log(.)) and sum of squares -> (loss, mf, reg)
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
    if (tid == 0 && a.loss_acc) {
        float sm = 0.f, ss = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            sm += red[0][w];
            ss += red[1][w];
        }
        const float mf = -sm * a.inv_B;             // -mean(maxi)          :114 / :704
        const float rg = a.reg_c * 0.5f * ss;       // regs * l2 / batch    :117-120
        unsafeAtomicAdd(a.loss_acc + 0, mf + rg);
        unsafeAtomicAdd(a.loss_acc + 1, mf);
        unsafeAtomicAdd(a.loss_acc + 2, rg);
    }
}

template <int D, bool BF>
__global__ void __launch_bounds__(512) bpr_step_kernel(StepArgs a) { bpr_step_body<D, BF>(a, (int)blockIdx.x); }

// The step of batch t and the sampler of batch t + 1 in ONE launch: workgroups [0, n_step_blocks) run the step, the ones
// behind them draw the next batch into the OTHER set of batch buffers.  The two are independent (the sampler never reads
// the tables) and both latency-bound on a handful of CUs; as consecutive graph nodes -- also on two captured streams --
// they ran back to back (23 us per step against 13 us for the slower of the two).
#ifndef PDA_SAMP_PER_BLOCK
#define PDA_SAMP_PER_BLOCK 64
#endif
constexpr int kSampPerBlock = PDA_SAMP_PER_BLOCK;   // one wave per sampler workgroup: 32 workgroups for a 2048-triplet batch
template <int D, bool BF>
__global__ void __launch_bounds__(512) bpr_step_sample_kernel(StepArgs a, SampleArgs sa, int n_sample_blocks) {
    // sampler workgroups first (they are dispatched first and have the longer dependent-load chains), kSampPerBlock triplets each
    if ((int)blockIdx.x >= n_sample_blocks) bpr_step_body<D, BF>(a, (int)blockIdx.x - n_sample_blocks);
    else if (threadIdx.x < kSampPerBlock) sample_one(sa, (int)blockIdx.x * kSampPerBlock + (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// n_steps training steps in ONE launch: a resident grid loops over the steps on the device.  Iteration i: the step workgroups
// run the fused SGD step on the batch in buffer set i & 1 while the sampler workgroups draw the batch of the next step into
// set (i + 1) & 1; a grid barrier (one device-scope counter) separates the iterations -- every update of step i is visible to
// the gathers of step i + 1, like two launches on one stream, without the launch (MF/train_new_api.py:1078-1096: the
// session.run loop, with the generator thread sampling ahead).
// ---------------------------------------------------------------------------------------------------------------------
struct TrainLoopArgs {
    StepArgs a[2];
    SampleArgs sa[2];
    uint64_t* step_ctr;        // device: the sampler step of the first batch drawn here; receives + n_steps at the end
    unsigned* bar;             // [0] arrivals (zeroed by the host call), [1] error flag (a barrier gave up)
    float* loss_steps;         // [n_steps][3] or NULL (then a[.].loss_acc accumulates over the steps)
    int n_steps, n_sample_blocks;      // n_sample_blocks: sampler workgroups of the grid (the rest step)
    int step_tiles, sample_tiles;       // work items of one iteration: each kind of workgroup strides over its own
};

constexpr unsigned kGridSpinMax = 1u << 22;

// Everything that crosses workgroups inside the loop -- the sampled batch, the table rows, the loss words -- is written and
// read at device scope (COH above), so the barrier is: my wave's stores and atomics are acknowledged (vmcnt(0)), count,
// wait for the count.  (With plain accesses it needs a write-back and an invalidation of the XCD's L2 per step: 30 us.)
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned target) {
    __shared__ int s_ok;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);                               // vmcnt(0) expcnt(0) lgkmcnt(0)
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spin = 0;
        int ok = 1;
        while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spin > kGridSpinMax) {
                ok = 0;                                          // not resident together
                break;
            }
        }
        if (!ok) __hip_atomic_store(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

template <int D>
__global__ void __launch_bounds__(512) bpr_train_loop_kernel(TrainLoopArgs t) {
    const bool sampler = (int)blockIdx.x < t.n_sample_blocks;
    const uint64_t step0 = *t.step_ctr;
    for (int i = 0; i < t.n_steps; ++i) {
        if (!sampler) {
            StepArgs a = t.a[i & 1];
            if (t.loss_steps) a.loss_acc = t.loss_steps + 3 * (size_t)i;
            const int n_step_wgs = (int)gridDim.x - t.n_sample_blocks;
            for (int tile = (int)blockIdx.x - t.n_sample_blocks; tile < t.step_tiles; tile += n_step_wgs) {
                bpr_step_body<D, false, true>(a, tile);
                __syncthreads();                                   // (the body's LDS arrays are reused by the next tile)
            }
        } else if (threadIdx.x < kSampPerBlock) {
            SampleArgs sa = t.sa[(i + 1) & 1];
            sa.step = step0 + (uint64_t)i;
            sa.step_dev = nullptr;
            sa.step_next = nullptr;
            for (int tile = (int)blockIdx.x; tile < t.sample_tiles; tile += t.n_sample_blocks)
                sample_one<true>(sa, tile * kSampPerBlock + (int)threadIdx.x);
        }
        if (!grid_barrier(t.bar, (unsigned)(i + 1) * gridDim.x)) return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *t.step_ctr = step0 + (uint64_t)t.n_steps;
}

// Sort one batch by positive item inside a single workgroup (B <= 4096): bitonic sort of (pos, slot) in LDS, then all
// five arrays are permuted through LDS.  Order inside a batch has no meaning for the loss or the update.
__global__ void __launch_bounds__(1024) sort_by_pos_kernel(int32_t* users, int32_t* pos, int32_t* neg, float* pos_pop,
                                                           float* neg_pop, int B, int n_pow2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sort[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem_sort);                  // [n_pow2]  (pos << 32) | slot
    int32_t* buf = reinterpret_cast<int32_t*>(keys + n_pow2);                 // [B] staging for one array at a time
    const int tid = threadIdx.x;
    for (int i = tid; i < n_pow2; i += 1024) keys[i] = i < B ? (((uint64_t)(uint32_t)pos[i] << 32) | (uint32_t)i) : ~0ull;
    __syncthreads();
    // Bitonic network over compare-exchange PAIRS: pair q of a step with distance jj touches i = ((q & ~(jj-1)) << 1) | (q & (jj-1))
    // and i + jj.  A wave's 64 consecutive pairs stay inside one aligned 128-element chunk as long as jj <= 64, so those
    // steps (51 of the 66 at n = 2048) need no workgroup barrier -- the wave owns its chunk and LDS executes its accesses in
    // order; only the steps with jj >= 128 are fenced by __syncthreads().  (Keys in registers with ds_bpermute exchanges
    // for jj <= 64 were measured slower: four 32-bit permutes per step.)
    const int half = n_pow2 >> 1;
    bool fenced = true;                      // the last thing that happened was a workgroup barrier
    for (int k = 2; k <= n_pow2; k <<= 1)
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            const bool cross = jj >= 128;
            if (cross && !fenced) __syncthreads();
            for (int q = tid; q < half; q += 1024) {
                const int i = ((q & ~(jj - 1)) << 1) | (q & (jj - 1)), p2 = i + jj;
                const uint64_t x = keys[i], y = keys[p2];
                const bool up = (i & k) == 0;
                if ((x > y) == up) {
                    keys[i] = y;
                    keys[p2] = x;
                }
            }
            if (cross) { __syncthreads(); fenced = true; }
            else { pda_wave_sync(); fenced = false; }
        }
    __syncthreads();
    // permutation of the five arrays, in place: every thread first GATHERS all its values (all loads in flight at once),
    // the workgroup barrier separates the reads from the writes
    int32_t* arrs[5] = {users, pos, neg, reinterpret_cast<int32_t*>(pos_pop), reinterpret_cast<int32_t*>(neg_pop)};
    (void)buf;
    int32_t vals[4][5];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + 1024 * e;
        if (i < B) {
            const uint32_t src = (uint32_t)keys[i];
#pragma unroll
            for (int q = 0; q < 5; ++q) vals[e][q] = arrs[q] ? arrs[q][src] : 0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + 1024 * e;
        if (i < B) {
#pragma unroll
            for (int q = 0; q < 5; ++q)
                if (arrs[q]) arrs[q][i] = vals[e][q];
        }
    }
}

// Group one batch by positive item inside a single workgroup (B <= 4096) without a sorting network: counting sort on a
// 12-bit hash of `pos` (LDS atomics), then every element ranks itself inside its bucket by (pos, original index).  The
// result is deterministic, a permutation of whole triplets, and every run of equal positives is contiguous -- all that
// pda_bpr_step_f32's on-chip run combining needs -- in ~1/5 of the bitonic sort's time (66 dependent LDS steps there).
__global__ void __launch_bounds__(1024) group_by_pos_kernel(int32_t* users, int32_t* pos, int32_t* neg, float* pos_pop,
                                                            float* neg_pop, int B) {
    constexpr int NBIN = 4096;
    __shared__ int cnt[NBIN + 1];      // counts, then exclusive bases (cnt[NBIN] = B)
    __shared__ uint64_t member[4096];  // (pos << 32 | source index) bucket by bucket, arrival order
    __shared__ int perm[4096];         // final: perm[dst] = src
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // one workgroup per batch: row blockIdx.x of [n][B] buffers (a single batch: row 0)
        const size_t off = (size_t)blockIdx.x * B;
        users += off;
        pos += off;
        neg += off;
        if (pos_pop) { pos_pop += off; neg_pop += off; }
    }
    for (int i = tid; i <= NBIN; i += 1024) cnt[i] = 0;
    __syncthreads();
    int bkt[4], arr[4];
    uint64_t mykey[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + 1024 * e;
        bkt[e] = arr[e] = 0;
        mykey[e] = 0;
        if (i < B) {
            const int p = pos[i];
            mykey[e] = ((uint64_t)(uint32_t)p << 32) | (uint32_t)i;
            bkt[e] = (int)(((uint32_t)p * 2654435761u) >> 20);
            arr[e] = atomicAdd(&cnt[bkt[e]], 1);
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
    for (int e = 0; e < 4; ++e) {
        const int i = tid + 1024 * e;
        if (i < B) member[cnt[bkt[e]] + arr[e]] = mykey[e];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + 1024 * e;
        if (i < B) {
            const int lo = cnt[bkt[e]], hi = cnt[bkt[e] + 1];
            int rank = 0;
#pragma unroll 8
            for (int m = lo; m < hi; ++m) rank += member[m] < mykey[e] ? 1 : 0;    // independent LDS reads: pipelined
            perm[lo + rank] = i;
        }
    }
    __syncthreads();
    int32_t* arrs[5] = {users, pos, neg, reinterpret_cast<int32_t*>(pos_pop), reinterpret_cast<int32_t*>(neg_pop)};
    int32_t vals[4][5];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + 1024 * e;
        if (i < B) {
            const int src = perm[i];
#pragma unroll
            for (int q = 0; q < 5; ++q) vals[e][q] = arrs[q] ? arrs[q][src] : 0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = tid + 1024 * e;
        if (i < B) {
#pragma unroll
            for (int q = 0; q < 5; ++q)
                if (arrs[q]) arrs[q][i] = vals[e][q];
        }
    }
}

// TF-1.14 Adam with dense decay: one streaming pass over a whole table (4 reads + 4 writes of n floats).
__global__ void __launch_bounds__(256) adam_dense_sweep_kernel(float* __restrict__ var, float* __restrict__ m,
                                                               float* __restrict__ v, float* __restrict__ g, size_t n4,
                                                               float lr_t, float b1, float b2, float eps) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 gg = reinterpret_cast<f32x4*>(g)[i];
        f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
        f32x4 xx = reinterpret_cast<f32x4*>(var)[i];
        bool touched = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            touched |= gg[k] != 0.f;
            mm[k] = b1 * mm[k] + (1.f - b1) * gg[k];
            vv[k] = b2 * vv[k] + (1.f - b2) * gg[k] * gg[k];
            xx[k] = xx[k] - lr_t * mm[k] / (sqrtf(vv[k]) + eps);
        }
        reinterpret_cast<f32x4*>(m)[i] = mm;
        reinterpret_cast<f32x4*>(v)[i] = vv;
        reinterpret_cast<f32x4*>(var)[i] = xx;
        if (touched) reinterpret_cast<f32x4*>(g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// Both tables of the model in one launch (the same arithmetic element by element): workgroups [0, blocks_a) sweep the first
// table, the rest the second -- one launch and one tail less per training step.
__global__ void __launch_bounds__(256) adam_dense_sweep2_kernel(float* __restrict__ var_a, float* __restrict__ m_a, float* __restrict__ v_a,
                                                                float* __restrict__ g_a, size_t n4_a, float* __restrict__ var_b,
                                                                float* __restrict__ m_b, float* __restrict__ v_b, float* __restrict__ g_b,
                                                                size_t n4_b, unsigned blocks_a, float lr_t, float b1, float b2, float eps) {
    const bool first = blockIdx.x < blocks_a;
    float* var = first ? var_a : var_b;
    float* m = first ? m_a : m_b;
    float* v = first ? v_a : v_b;
    float* g = first ? g_a : g_b;
    const size_t n4 = first ? n4_a : n4_b;
    const size_t blk = first ? blockIdx.x : blockIdx.x - blocks_a, nblk = first ? blocks_a : gridDim.x - blocks_a;
    const size_t stride = nblk * blockDim.x;
    for (size_t i = blk * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 gg = reinterpret_cast<f32x4*>(g)[i];
        f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
        f32x4 xx = reinterpret_cast<f32x4*>(var)[i];
        bool touched = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            touched |= gg[k] != 0.f;
            mm[k] = b1 * mm[k] + (1.f - b1) * gg[k];
            vv[k] = b2 * vv[k] + (1.f - b2) * gg[k] * gg[k];
            xx[k] = xx[k] - lr_t * mm[k] / (sqrtf(vv[k]) + eps);
        }
        reinterpret_cast<f32x4*>(m)[i] = mm;
        reinterpret_cast<f32x4*>(v)[i] = vv;
        reinterpret_cast<f32x4*>(var)[i] = xx;
        if (touched) reinterpret_cast<f32x4*>(g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}


// ---- the dense-decay sweep as SIX streams (round 5): a step's gradient is zero on all but the batch's rows, so the dense gradient tables need not
// be read at all -- one bit per row says "touched by this step" (adam_mark_rows_kernel: the batch's users / positives / negatives), the sweep
// reads and clears g only where it is set (and its bits behind itself).  Same arithmetic as adam_dense_sweep2_kernel, operation for operation
// (an untouched row computes with g = 0): bit-identical tables.  4.3 -> 3.7 GB per step on config 3's tables.
__global__ void __launch_bounds__(256) adam_mark_rows_kernel(const int32_t* __restrict__ users, const int32_t* __restrict__ pos, const int32_t* __restrict__ neg,
                                                             int B, uint32_t* __restrict__ touched_u, uint32_t* __restrict__ touched_i) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const int u = users[i], p = pos[i], n = neg[i];
    atomicOr(&touched_u[u >> 5], 1u << (u & 31));
    atomicOr(&touched_i[p >> 5], 1u << (p & 31));
    atomicOr(&touched_i[n >> 5], 1u << (n & 31));
}
__global__ void __launch_bounds__(256) adam_dense_sweep3_kernel(float* __restrict__ var_a, float* __restrict__ m_a, float* __restrict__ v_a,
                                                                float* __restrict__ g_a, size_t n4_a, int sh_a, const uint32_t* __restrict__ t_a,
                                                                float* __restrict__ var_b, float* __restrict__ m_b, float* __restrict__ v_b,
                                                                float* __restrict__ g_b, size_t n4_b, int sh_b, const uint32_t* __restrict__ t_b,
                                                                unsigned blocks_a, float lr_t, float b1, float b2, float eps) {
    const bool first = blockIdx.x < blocks_a;
    float* var = first ? var_a : var_b;
    float* m = first ? m_a : m_b;
    float* v = first ? v_a : v_b;
    float* g = first ? g_a : g_b;
    const uint32_t* tb = first ? t_a : t_b;
    const int sh = first ? sh_a : sh_b;                       // log2 of the row's 16-byte chunks
    const size_t n4 = first ? n4_a : n4_b;
    const size_t blk = first ? blockIdx.x : blockIdx.x - blocks_a, nblk = first ? blocks_a : gridDim.x - blocks_a;
    const size_t stride = nblk * blockDim.x;
    auto ld = [&](float* p, size_t i) __attribute__((always_inline)) -> f32x4 {
#ifdef PDA_ADAM_PLAIN_STREAMS
        return reinterpret_cast<f32x4*>(p)[i];
#else
        return __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p) + i);
#endif
    };
    auto st = [&](float* p, size_t i, f32x4 x) __attribute__((always_inline)) {
#ifdef PDA_ADAM_PLAIN_STREAMS
        reinterpret_cast<f32x4*>(p)[i] = x;
#else
        __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(p) + i);
#endif
    };
    // (streams read once and written once per step: non-temporal, so that 3.7 GB of them do not push the batch's rows out of L2 / MALL: 755 -> 742 us;
    // PDA_ADAM_UNROLL chunks per thread and iteration in flight)
#ifndef PDA_ADAM_UNROLL
#define PDA_ADAM_UNROLL 2
#endif
    constexpr int UN = PDA_ADAM_UNROLL;
    for (size_t i0 = blk * blockDim.x + threadIdx.x; i0 < n4; i0 += UN * stride) {
        f32x4 gg[UN], mm[UN], vv[UN], xx[UN];
        bool touched[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const size_t i = i0 + q * stride;
            const bool in = i < n4;
            const size_t row = (in ? i : i0) >> sh;
            touched[q] = in && ((tb[row >> 5] >> (row & 31)) & 1u) != 0u;
            gg[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (touched[q]) gg[q] = reinterpret_cast<f32x4*>(g)[i];
            if (in) {
                mm[q] = ld(m, i);
                vv[q] = ld(v, i);
                xx[q] = ld(var, i);
            }
        }
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const size_t i = i0 + q * stride;
            if (i >= n4) break;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mm[q][k] = b1 * mm[q][k] + (1.f - b1) * gg[q][k];
                vv[q][k] = b2 * vv[q][k] + (1.f - b2) * gg[q][k] * gg[q][k];
                xx[q][k] = xx[q][k] - lr_t * mm[q][k] / (sqrtf(vv[q][k]) + eps);
            }
            st(m, i, mm[q]);
            st(v, i, vv[q]);
            st(var, i, xx[q]);
            if (touched[q]) reinterpret_cast<f32x4*>(g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// ---- round 6: the sweep keyed by per-row step TAGS instead of bitmaps (tag[row] == step  <=>  the step's batch touched the row; written by
// the step kernel itself, never cleared: no mark launch, no memsets -- five launches per reference step become two), and the cache policy by
// working set: NT = false reads and writes x, m, v with plain accesses, so tables that fit the 256 MiB Infinity Cache (C1 / C2: 54 MB)
// are swept out of it instead of out of HBM; NT = true is adam_dense_sweep3_kernel's streaming policy for the big tables.  Same
// arithmetic as adam_dense_sweep2_kernel / 3, operation for operation: bit-identical tables.
template <bool NT, int UN>
__global__ void __launch_bounds__(256) adam_dense_sweep4_kernel(float* __restrict__ var_a, float* __restrict__ m_a, float* __restrict__ v_a,
                                                                float* __restrict__ g_a, size_t n4_a, const int32_t* __restrict__ t_a,
                                                                float* __restrict__ var_b, float* __restrict__ m_b, float* __restrict__ v_b,
                                                                float* __restrict__ g_b, size_t n4_b, const int32_t* __restrict__ t_b,
                                                                int sh, int tag, unsigned blocks_a, float lr_t, float b1, float b2, float eps) {
    const bool first = blockIdx.x < blocks_a;
    float* var = first ? var_a : var_b;
    float* m = first ? m_a : m_b;
    float* v = first ? v_a : v_b;
    float* g = first ? g_a : g_b;
    const int32_t* tg = first ? t_a : t_b;
    const size_t n4 = first ? n4_a : n4_b;
    const size_t blk = first ? blockIdx.x : blockIdx.x - blocks_a, nblk = first ? blocks_a : gridDim.x - blocks_a;
    const size_t stride = nblk * blockDim.x;
    auto ld = [&](float* p, size_t i) __attribute__((always_inline)) -> f32x4 {
        if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p) + i);
        else return reinterpret_cast<f32x4*>(p)[i];
    };
    auto st = [&](float* p, size_t i, f32x4 x) __attribute__((always_inline)) {
        if constexpr (NT) __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(p) + i);
        else reinterpret_cast<f32x4*>(p)[i] = x;
    };
    for (size_t i0 = blk * blockDim.x + threadIdx.x; i0 < n4; i0 += UN * stride) {
        f32x4 gg[UN], mm[UN], vv[UN], xx[UN];
        bool touched[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const size_t i = i0 + q * stride;
            const bool in = i < n4;
            touched[q] = in && tg[i >> sh] == tag;
            gg[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (in) {
                mm[q] = ld(m, i);
                vv[q] = ld(v, i);
                xx[q] = ld(var, i);
            }
        }
#pragma unroll
        for (int q = 0; q < UN; ++q)
            if (touched[q]) gg[q] = reinterpret_cast<f32x4*>(g)[i0 + q * stride];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const size_t i = i0 + q * stride;
            if (i >= n4) break;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mm[q][k] = b1 * mm[q][k] + (1.f - b1) * gg[q][k];
                vv[q][k] = b2 * vv[q][k] + (1.f - b2) * gg[q][k] * gg[q][k];
                xx[q][k] = xx[q][k] - lr_t * mm[q][k] / (sqrtf(vv[q][k]) + eps);
            }
            st(m, i, mm[q]);
            st(v, i, vv[q]);
            st(var, i, xx[q]);
            if (touched[q]) reinterpret_cast<f32x4*>(g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

template <int D>
__global__ void __launch_bounds__(256) adam_rows_kernel(float* var, float* m, float* v, float* g, const int32_t* rows,
                                                        int n_rows, float lr_t, float b1, float b2, float eps) {
    constexpr int L = D / 4, RPB = 256 / L;
    const int r = blockIdx.x * RPB + threadIdx.x / L, e = threadIdx.x % L;
    if (r >= n_rows) return;
    const size_t off = (size_t)rows[r] * D + 4 * e;
    f32x4 gg = *reinterpret_cast<f32x4*>(g + off), mm = *reinterpret_cast<f32x4*>(m + off);
    f32x4 vv = *reinterpret_cast<f32x4*>(v + off), xx = *reinterpret_cast<f32x4*>(var + off);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mm[k] = b1 * mm[k] + (1.f - b1) * gg[k];
        vv[k] = b2 * vv[k] + (1.f - b2) * gg[k] * gg[k];
        xx[k] = xx[k] - lr_t * mm[k] / (sqrtf(vv[k]) + eps);
    }
    *reinterpret_cast<f32x4*>(m + off) = mm;
    *reinterpret_cast<f32x4*>(v + off) = vv;
    *reinterpret_cast<f32x4*>(var + off) = xx;
    *reinterpret_cast<f32x4*>(g + off) = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ---------------------------------------------------------------------------------------------------------------------
// Exact lazy dense-decay Adam.  A row whose gradient is zero at step k only decays: m <- b1 m, v <- b2 v,
// x <- x - lr_k m / (sqrt(v) + eps) -- a function of the row and of k alone (the arithmetic of adam_dense_sweep_kernel with
// g = 0, operation for operation).  So a row may skip its idle steps and replay them in registers when it is next needed:
// last[row] = the step the row is current for, lr_tab[k] = the bias-corrected rate of step k.  Per training step t:
//   phase 0  every row of the batch is brought to step t - 1 (the forward pass of step t reads it);
//   (pda_bpr_step_f32, PDA_UPD_DENSE_GRAD: gradients into the dense accumulators)
//   phase 1  every row of the batch takes step t with its summed gradient; the accumulator row is cleared.
// Rows may repeat inside the lists (an item that is positive for one user and negative for another): the first group of
// threads to raise last[row] owns the row, the others leave.  Traffic per step: the batch rows -- not the 3.7 GB (config 3)
// or 74 GB (config 5) of the dense sweep.
// ---------------------------------------------------------------------------------------------------------------------
struct LazyAdamArgs {
    float *U, *mU, *vU, *gU;
    int32_t* lastU;
    float *I, *mI, *vI, *gI;
    int32_t* lastI;
    const int32_t *users, *pos, *neg;
    const float* lr_tab;
    int B, t, phase;
    float b1, b2, eps;
    const int32_t* t_dev;      // NULL, or the step in device memory (pda_adam_lazy_dev_f32: HIP-graph replays advance it)
    int32_t* t_next;           // phase 1 stores t + 1 here (the OTHER of two counter slots: no block reads it in this launch)
    int n_tab;                 // entries of lr_tab (t_dev mode: a step beyond the table leaves the tables alone)
};

// Steps from .. upto of an idle row, in registers.  The update lr_k m / (sqrt(v) + eps) shrinks by at least 0.912 per step: m by
// 0.9, the denominator by no more than sqrt(0.999) = 0.9995, and lr_k = lr sqrt(1 - b2^k) / (1 - b1^k) -- NOT monotone: it falls
// until k ~ 10 and then rises, fastest around k = 20 .. 50, by at most 1.2 % per step (sqrt((k + 1) / k) while b2^k ~ 1 - 0.001 k)
// -- 0.9 x 1.012 / 0.9995 = 0.911.  So once a step leaves all four x unchanged -- the update is below half an ulp -- every later
// step does too: from there on only m and v decay (two multiplications per element and step instead of a square root and a
// division).  Same results, bit for bit (tests: idle gaps inside steps 1 .. 100, where lr_k is not monotone, included).
template <int D>
__device__ __forceinline__ void adam_replay(f32x4& xx, f32x4& mm, f32x4& vv, int from, int upto, const float* __restrict__ lr_tab, float b1,
                                            float b2, float eps) {
    int k = from;
    for (; k <= upto; ++k) {
        const float lr_k = lr_tab[k];
        bool moved = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mm[q] = b1 * mm[q] + (1.f - b1) * 0.f;
            vv[q] = b2 * vv[q] + (1.f - b2) * 0.f * 0.f;
            const float xn = xx[q] - lr_k * mm[q] / (sqrtf(vv[q]) + eps);
            moved |= xn != xx[q];
            xx[q] = xn;
        }
        if (!moved) { ++k; break; }
    }
    for (; k <= upto; ++k) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mm[q] = b1 * mm[q] + (1.f - b1) * 0.f;
            vv[q] = b2 * vv[q] + (1.f - b2) * 0.f * 0.f;
        }
    }
}

// The same catch-up to the north_star's tolerance instead of bit for bit (round 3).  What the exact replay pays per element and
// idle step is a correctly rounded square root and a division (~35 VALU); but sqrt(v_k) = sqrt(v_0) sqrt(b2)^k is a running
// product, the division a v_rcp_f32 (1 ulp), and the terms lr_k m_k / (sqrt(v_k) + eps) fall by >= 0.91 per step, so the loop
// ends when a term can no longer move x (the remaining geometric tail is below 1e-11 or 2^-28 |x|) -- a few dozen steps of 6
// VALU -- and m, v take their closed-form powers b^n (one v_exp_f32 each) whatever the length of the gap.  Against the exact
// replay / the dense sweep: x to 1e-6 absolute, m and v to 1e-4 relative (the sweep's own 3 000 roundings are ~1e-5 from the
// real-number value); tests/test_gpu_bpr_step.py.
struct FastConsts { float sqrt_b2, log2_b1, log2_b2; };
template <int D>
__device__ __forceinline__ void adam_replay_fast(f32x4& xx, f32x4& mm, f32x4& vv, int from, int upto, const float* __restrict__ lr_tab, float b1,
                                                 float eps, const FastConsts fc) {
    const int n = upto - from + 1;
    if (n <= 0) return;
    f32x4 s, m = mm;
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] = sqrtf(vv[q]);
    for (int k = from; k <= upto; ++k) {
        const float lr_k = lr_tab[k];
        bool live = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            m[q] *= b1;
            s[q] *= fc.sqrt_b2;
            const float t = lr_k * m[q] * __builtin_amdgcn_rcpf(s[q] + eps);
            xx[q] -= t;
            live |= fabsf(t) > fmaxf(fabsf(xx[q]) * 3.7e-9f, 1.0e-12f);
        }
        if (!live) break;
    }
    const float p1 = exp2f((float)n * fc.log2_b1), p2 = exp2f((float)n * fc.log2_b2);
    mm = mm * p1;
    vv = vv * p2;
}

template <int D, bool FAST = false>
__global__ void __launch_bounds__(256) adam_lazy_kernel(LazyAdamArgs a, FastConsts fc = FastConsts{}) {
    constexpr int L = D / 4, RPB = 256 / L;
    const int ridx = blockIdx.x * RPB + threadIdx.x / L, e = threadIdx.x % L;
    if (ridx >= 3 * a.B) return;
    const int which = ridx / a.B, i = ridx - which * a.B;
    const bool user = which == 0;
    if (a.t_dev != nullptr) {
        a.t = *a.t_dev;
        if (a.phase == 1 && ridx == 0 && e == 0) *a.t_next = a.t + 1;
        if (a.t < 1 || a.t >= a.n_tab) return;
    }
    const int row = user ? a.users[i] : (which == 1 ? a.pos[i] : a.neg[i]);
    float* var = user ? a.U : a.I;
    float* m = user ? a.mU : a.mI;
    float* v = user ? a.vU : a.vI;
    float* g = user ? a.gU : a.gI;
    int32_t* last = user ? a.lastU : a.lastI;
    const int target = a.phase == 0 ? a.t - 1 : a.t;
    int old = 0;
    if (e == 0) old = atomicMax(&last[row], target);
    old = __shfl(old, (int)((threadIdx.x & 63) / L) * L, 64);
    if (old >= target) return;                                       // current already, or another group of this launch owns the row
    const size_t off = (size_t)row * D + 4 * e;
    f32x4 mm = *reinterpret_cast<f32x4*>(m + off), vv = *reinterpret_cast<f32x4*>(v + off), xx = *reinterpret_cast<f32x4*>(var + off);
    if constexpr (FAST) adam_replay_fast<D>(xx, mm, vv, old + 1, a.t - 1, a.lr_tab, a.b1, a.eps, fc);
    else adam_replay<D>(xx, mm, vv, old + 1, a.t - 1, a.lr_tab, a.b1, a.b2, a.eps);
    if (a.phase == 1) {
        const f32x4 gg = *reinterpret_cast<f32x4*>(g + off);
        const float lr_t = a.lr_tab[a.t];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mm[q] = a.b1 * mm[q] + (1.f - a.b1) * gg[q];
            vv[q] = a.b2 * vv[q] + (1.f - a.b2) * gg[q] * gg[q];
            xx[q] = xx[q] - lr_t * mm[q] / (sqrtf(vv[q]) + a.eps);
        }
        *reinterpret_cast<f32x4*>(g + off) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    *reinterpret_cast<f32x4*>(m + off) = mm;
    *reinterpret_cast<f32x4*>(v + off) = vv;
    *reinterpret_cast<f32x4*>(var + off) = xx;
}

// every row of a table up to step t (before an evaluation, a checkpoint, a switch of optimiser)
template <int D, bool FAST = false>
__global__ void __launch_bounds__(256) adam_lazy_sync_kernel(float* var, float* m, float* v, int32_t* last, size_t n_rows, int t,
                                                             const float* __restrict__ lr_tab, float b1, float b2, float eps,
                                                             FastConsts fc = FastConsts{}) {
    constexpr int L = D / 4, RPB = 256 / L;
    const int e = threadIdx.x % L;
    for (size_t row = (size_t)blockIdx.x * RPB + threadIdx.x / L; row < n_rows; row += (size_t)gridDim.x * RPB) {
        const int old = last[row];
        if (old >= t) continue;
        const size_t off = row * D + 4 * e;
        f32x4 mm = *reinterpret_cast<f32x4*>(m + off), vv = *reinterpret_cast<f32x4*>(v + off), xx = *reinterpret_cast<f32x4*>(var + off);
        if constexpr (FAST) adam_replay_fast<D>(xx, mm, vv, old + 1, t, lr_tab, b1, eps, fc);
        else adam_replay<D>(xx, mm, vv, old + 1, t, lr_tab, b1, b2, eps);
        *reinterpret_cast<f32x4*>(m + off) = mm;
        *reinterpret_cast<f32x4*>(v + off) = vv;
        *reinterpret_cast<f32x4*>(var + off) = xx;
        __builtin_amdgcn_wave_barrier();
        if (e == 0) last[row] = t;
    }
}

template <int D, bool BF = false>
int launch_step(const StepArgs& a, hipStream_t s, const SampleArgs* next = nullptr) {
    constexpr int TPB = 512 / (D / 4);
    const int nsb = (a.B + TPB - 1) / TPB;
    const int nsamp = next ? (next->B + kSampPerBlock - 1) / kSampPerBlock : 0;
    if (next) hipLaunchKernelGGL((bpr_step_sample_kernel<D, BF>), dim3((unsigned)(nsb + nsamp)), dim3(512), 0, s, a, *next, nsamp);
    else hipLaunchKernelGGL((bpr_step_kernel<D, BF>), dim3((unsigned)nsb), dim3(512), 0, s, a);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

// shadow[row] = RNE_bf16(master[row]) for a list of rows (duplicates harmless); one 16-byte store per thread
__global__ void __launch_bounds__(256) refresh_rows_kernel(const float* __restrict__ master, uint16_t* __restrict__ shadow,
                                                          const int32_t* __restrict__ rows, int row_offset, size_t n8, int d8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const size_t r = rows ? (size_t)(rows[i / d8] - row_offset) : i / d8;
    const size_t off = r * (8 * (size_t)d8) + 8 * (i % d8);
    const f32x4 a = *reinterpret_cast<const f32x4*>(master + off), b = *reinterpret_cast<const f32x4*>(master + off + 4);
    auto rne = [](float x) -> uint32_t {
        const uint32_t u = __float_as_uint(x);
        return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    };
    uint4 o;
    o.x = rne(a[0]) | (rne(a[1]) << 16);
    o.y = rne(a[2]) | (rne(a[3]) << 16);
    o.z = rne(b[0]) | (rne(b[1]) << 16);
    o.w = rne(b[2]) | (rne(b[3]) << 16);
    *reinterpret_cast<uint4*>(shadow + off) = o;
}

}  // namespace

extern "C" int pda_bpr_step_f32(float* U, float* I, const int32_t* users, const int32_t* pos, const int32_t* neg,
                                const float* pos_pop, const float* neg_pop, int B, int d, float regs, float reg_div,
                                float lr, int update_mode, float* g_user, float* g_pos, float* g_neg, float* gU,
                                float* gI, float* loss_acc, void* stream) {
    if (!U || !I || !users || !pos || !neg || B <= 0 || reg_div <= 0.f) return PDA_ERR_ARG;
    if ((pos_pop == nullptr) != (neg_pop == nullptr)) return PDA_ERR_ARG;
    const int any_order = (update_mode & PDA_UPD_ANY_ORDER) ? 1 : 0;
    [[maybe_unused]] const int users_distinct = (update_mode & PDA_UPD_USERS_DISTINCT) ? 1 : 0;
    update_mode &= ~(PDA_UPD_ANY_ORDER | PDA_UPD_USERS_DISTINCT);
    if (update_mode < PDA_UPD_NONE || update_mode > PDA_UPD_DENSE_GRAD) return PDA_ERR_ARG;
    if (update_mode == PDA_UPD_DENSE_GRAD && (!gU || !gI)) return PDA_ERR_ARG;
    if (g_user && (!g_pos || !g_neg)) return PDA_ERR_ARG;
    StepArgs a{U, I, users, pos, neg, pos_pop, neg_pop, g_user, g_pos, g_neg, gU, gI, loss_acc,
               B, 1.0f / (float)B, regs / reg_div, lr, update_mode, 0, d, U, I, any_order, users_distinct};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (d) {
        case 32: return launch_step<32>(a, s);
        case 64: return launch_step<64>(a, s);
        case 128: return launch_step<128>(a, s);
        case 256: return launch_step<256>(a, s);
        default: return PDA_ERR_UNSUPPORTED;
    }
}

extern "C" int pda_bpr_step_sample_f32(float* U, float* I, const int32_t* users, const int32_t* pos, const int32_t* neg,
                                       const float* pos_pop, const float* neg_pop, int B, int d, float regs, float reg_div,
                                       float lr, int update_mode, float* loss_acc, const pda_sample_job* next, void* stream) {
    if (!U || !I || !users || !pos || !neg || B <= 0 || reg_div <= 0.f || !next) return PDA_ERR_ARG;
    if ((pos_pop == nullptr) != (neg_pop == nullptr)) return PDA_ERR_ARG;
    const int any_order = (update_mode & PDA_UPD_ANY_ORDER) ? 1 : 0;
    [[maybe_unused]] const int users_distinct = (update_mode & PDA_UPD_USERS_DISTINCT) ? 1 : 0;
    update_mode &= ~(PDA_UPD_ANY_ORDER | PDA_UPD_USERS_DISTINCT);
    if (update_mode != PDA_UPD_SGD_FUSED && update_mode != PDA_UPD_NONE) return PDA_ERR_ARG;
    // the sampler job: validated like pda_sample_triplets_dev; its outputs must not be this step's inputs
    if (!next->users || !next->train_indptr || !next->train_indices || !next->pos || !next->neg || !next->step_dev || next->B <= 0 ||
        next->neg_hi <= next->neg_lo)
        return PDA_ERR_ARG;
    if (next->step_next == next->step_dev) return PDA_ERR_ARG;
    if (next->gen_users && next->n_pool <= 0) return PDA_ERR_ARG;
    if (next->pop_matrix && (!next->pos_pop || !next->neg_pop || next->n_slots <= 0)) return PDA_ERR_ARG;
    if (next->users == users || next->pos == pos || next->neg == neg) return PDA_ERR_ARG;
    StepArgs a{U, I, users, pos, neg, pos_pop, neg_pop, nullptr, nullptr, nullptr, nullptr, nullptr, loss_acc,
               B, 1.0f / (float)B, regs / reg_div, lr, update_mode, 0, d, U, I, any_order};
    SampleArgs sa{next->users, next->user_pool, next->train_indptr, next->train_indices, next->train_slots, next->pop_matrix,
                  next->pos, next->neg, next->pos_pop, next->neg_pop, next->seed, 0, next->B, next->n_pool, next->gen_users,
                  next->neg_lo, next->neg_hi, next->n_slots, next->step_dev, next->step_next};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (d) {
        case 32: return launch_step<32>(a, s, &sa);
        case 64: return launch_step<64>(a, s, &sa);
        case 128: return launch_step<128>(a, s, &sa);
        case 256: return launch_step<256>(a, s, &sa);
        default: return PDA_ERR_UNSUPPORTED;
    }
}

extern "C" int pda_bpr_train_steps_f32(float* U, float* I, int d, float regs, float reg_div, float lr, int update_mode,
                                       const pda_sample_job* set0, const pda_sample_job* set1, uint64_t* step_ctr, int n_steps,
                                       float* loss_acc, float* loss_steps, void* barrier_ws, void* stream) {
    if (!U || !I || !set0 || !set1 || !step_ctr || !barrier_ws || n_steps <= 0 || reg_div <= 0.f) return PDA_ERR_ARG;
    if (!loss_acc && !loss_steps) return PDA_ERR_ARG;
    const int any_order = (update_mode & PDA_UPD_ANY_ORDER) ? 1 : 0;
    [[maybe_unused]] const int users_distinct = (update_mode & PDA_UPD_USERS_DISTINCT) ? 1 : 0;
    update_mode &= ~(PDA_UPD_ANY_ORDER | PDA_UPD_USERS_DISTINCT);
    if (update_mode != PDA_UPD_SGD_FUSED) return PDA_ERR_ARG;
    TrainLoopArgs t{};
    const pda_sample_job* sets[2] = {set0, set1};
    const int B = set0->B;
    for (int q = 0; q < 2; ++q) {
        const pda_sample_job* j = sets[q];
        if (!j->users || !j->train_indptr || !j->train_indices || !j->pos || !j->neg || j->B != B || B <= 0 || j->neg_hi <= j->neg_lo)
            return PDA_ERR_ARG;
        if (j->gen_users && j->n_pool <= 0) return PDA_ERR_ARG;
        if (j->pop_matrix && (!j->pos_pop || !j->neg_pop || j->n_slots <= 0)) return PDA_ERR_ARG;
        if ((j->pos_pop == nullptr) != (j->neg_pop == nullptr)) return PDA_ERR_ARG;
        t.a[q] = StepArgs{U, I, j->users, j->pos, j->neg, j->pos_pop, j->neg_pop, nullptr, nullptr, nullptr, nullptr, nullptr, loss_acc,
                          B, 1.0f / (float)B, regs / reg_div, lr, update_mode, 0, d, U, I, any_order};
        t.sa[q] = SampleArgs{j->users, j->user_pool, j->train_indptr, j->train_indices, j->train_slots, j->pop_matrix, j->pos, j->neg,
                             j->pos_pop, j->neg_pop, j->seed, 0, B, j->n_pool, j->gen_users, j->neg_lo, j->neg_hi, j->n_slots, nullptr, nullptr};
    }
    if (set0->users == set1->users || set0->pos == set1->pos || set0->neg == set1->neg) return PDA_ERR_ARG;
    t.step_ctr = step_ctr;
    t.bar = reinterpret_cast<unsigned*>(barrier_ws);
    t.loss_steps = loss_steps;
    t.n_steps = n_steps;
    t.sample_tiles = (B + kSampPerBlock - 1) / kSampPerBlock;
    t.n_sample_blocks = t.sample_tiles < 64 ? t.sample_tiles : 64;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // only the arrival counter: bar[1] ("a barrier gave up: the loop was abandoned") is STICKY -- the caller zeroes it once and reads it
    // whenever it next synchronises, however many launches later (zeroing it here hid a failed launch behind the next one)
    if (hipMemsetAsync(barrier_ws, 0, 4, s) != hipSuccess) return PDA_ERR_LAUNCH;
#define PDA_LOOP(DD)                                                                                                     \
    case DD: {                                                                                                           \
        constexpr int TPB = 512 / (DD / 4);                                                                              \
        t.step_tiles = (B + TPB - 1) / TPB;                                                                              \
        /* every workgroup must be resident: at most 384 + 64 of them (two per CU fit), striding over the batch */      \
        const int grid = (t.step_tiles < 384 ? t.step_tiles : 384) + t.n_sample_blocks;                                  \
        hipLaunchKernelGGL(bpr_train_loop_kernel<DD>, dim3((unsigned)grid), dim3(512), 0, s, t);                         \
        break;                                                                                                           \
    }
    switch (d) {
        PDA_LOOP(32) PDA_LOOP(64) PDA_LOOP(128) PDA_LOOP(256)
        default: return PDA_ERR_UNSUPPORTED;
    }
#undef PDA_LOOP
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

extern "C" int pda_bpr_step_shard_f32(const float* U, float* I_shard, int item_offset, const int32_t* users, const int32_t* pos,
                                      const int32_t* neg, const float* pos_pop, const float* neg_pop, int B_local, int d,
                                      float regs, float reg_div, float mean_div, float lr, float* g_user, int g_stride,
                                      float* gI_shard, float* loss_acc, void* stream) {
    if (!U || !I_shard || !users || !pos || !neg || !g_user || B_local <= 0 || reg_div <= 0.f || mean_div <= 0.f || item_offset < 0)
        return PDA_ERR_ARG;
    if (g_stride < d || (g_stride & 3)) return PDA_ERR_ARG;
    if ((pos_pop == nullptr) != (neg_pop == nullptr)) return PDA_ERR_ARG;
    StepArgs a{const_cast<float*>(U), I_shard, users, pos, neg, pos_pop, neg_pop, g_user, nullptr, nullptr, nullptr, gI_shard,
               loss_acc, B_local, 1.0f / mean_div, regs / reg_div, lr, gI_shard ? PDA_UPD_DENSE_ITEMS : PDA_UPD_SGD_ITEMS,
               item_offset, g_stride, U, I_shard, 1};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (d) {
        case 32: return launch_step<32>(a, s);
        case 64: return launch_step<64>(a, s);
        case 128: return launch_step<128>(a, s);
        case 256: return launch_step<256>(a, s);
        default: return PDA_ERR_UNSUPPORTED;
    }
}

namespace {
// U[users[i]] -= lr * g[i]   (one float4 per thread; atomics: the same user may arrive from several ranks)
__global__ void __launch_bounds__(256) apply_user_grads_kernel(float* __restrict__ U, const int32_t* __restrict__ users,
                                                              const float* __restrict__ g, size_t n4, int d4, int g_stride, float nlr) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const size_t r = i / d4;
    const int e = (int)(i % d4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + r * g_stride + 4 * e);
    atomic_add4(U + (size_t)users[r] * (4 * d4) + 4 * e, v * nlr);
}
}  // namespace

extern "C" int pda_apply_user_grads_f32(float* U, const int32_t* users, const float* g, int n, int d, int g_stride, float lr,
                                        void* stream) {
    if (!U || !users || !g || n <= 0 || d <= 0 || (d & 3) || g_stride < d || (g_stride & 3)) return PDA_ERR_ARG;
    const size_t n4 = (size_t)n * (d / 4);
    hipLaunchKernelGGL(apply_user_grads_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       U, users, g, n4, d / 4, g_stride, -lr);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

namespace {
// The apply phase of the exact mini-batch SGD step: U[users[t]] -= lr g_user[t], I[pos[t]] -= lr g_pos[t], I[neg[t]] -= lr g_neg[t]
// (one float4 per thread, atomics: rows repeat inside a batch).  No table is READ for arithmetic here, so -- unlike the fused
// in-kernel update -- no triplet can see another triplet's update of the same batch.
__global__ void __launch_bounds__(256) sgd_apply_kernel(float* __restrict__ U, float* __restrict__ I, const int32_t* __restrict__ users,
                                                       const int32_t* __restrict__ pos, const int32_t* __restrict__ neg,
                                                       const float* __restrict__ gu, const float* __restrict__ gp,
                                                       const float* __restrict__ gn, size_t n4, int d4, float nlr) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * n4) return;
    const int which = (int)(i / n4);
    const size_t k = i % n4, r = k / d4;
    const int e = (int)(k % d4);
    const float* g = which == 0 ? gu : (which == 1 ? gp : gn);
    const int32_t row = which == 0 ? users[r] : (which == 1 ? pos[r] : neg[r]);
    float* tab = which == 0 ? U : I;
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + r * (size_t)(4 * d4) + 4 * e);
    atomic_add4(tab + (size_t)row * (4 * d4) + 4 * e, v * nlr);
}
}  // namespace

extern "C" int pda_sgd_apply_f32(float* U, float* I, const int32_t* users, const int32_t* pos, const int32_t* neg, const float* g_user,
                                 const float* g_pos, const float* g_neg, int B, int d, float lr, void* stream) {
    if (!U || !I || !users || !pos || !neg || !g_user || !g_pos || !g_neg || B <= 0 || d <= 0 || (d & 3)) return PDA_ERR_ARG;
    const size_t n4 = (size_t)B * (d / 4);
    hipLaunchKernelGGL(sgd_apply_kernel, dim3((unsigned)((3 * n4 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), U, I,
                       users, pos, neg, g_user, g_pos, g_neg, n4, d / 4, -lr);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

extern "C" int pda_sort_triplets_by_pos(int32_t* users, int32_t* pos, int32_t* neg, float* pos_pop, float* neg_pop, int B,
                                        void* stream) {
    if (!users || !pos || !neg || B <= 0) return PDA_ERR_ARG;
    if ((pos_pop == nullptr) != (neg_pop == nullptr)) return PDA_ERR_ARG;
    if (B > 4096) return PDA_ERR_UNSUPPORTED;
    int n2 = 1;
    while (n2 < B) n2 <<= 1;
    const size_t smem = (size_t)n2 * 8 + (size_t)B * 4;
    hipLaunchKernelGGL(sort_by_pos_kernel, dim3(1), dim3(1024), smem, reinterpret_cast<hipStream_t>(stream), users, pos, neg,
                       pos_pop, neg_pop, B, n2);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

extern "C" int pda_group_triplets_by_pos(int32_t* users, int32_t* pos, int32_t* neg, float* pos_pop, float* neg_pop, int B,
                                         void* stream) {
    if (!users || !pos || !neg || B <= 0) return PDA_ERR_ARG;
    if ((pos_pop == nullptr) != (neg_pop == nullptr)) return PDA_ERR_ARG;
    if (B > 4096) return PDA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(group_by_pos_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), users, pos, neg,
                       pos_pop, neg_pop, B);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

extern "C" int pda_group_triplets_by_pos_batches(int32_t* users, int32_t* pos, int32_t* neg, float* pos_pop, float* neg_pop, int B,
                                                 int n_batches, void* stream) {
    if (!users || !pos || !neg || B <= 0 || n_batches <= 0) return PDA_ERR_ARG;
    if ((pos_pop == nullptr) != (neg_pop == nullptr)) return PDA_ERR_ARG;
    if (B > 4096) return PDA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(group_by_pos_kernel, dim3((unsigned)n_batches), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), users, pos, neg,
                       pos_pop, neg_pop, B);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

extern "C" int pda_adam_dense_sweep_f32(float* var, float* m, float* v, float* g, size_t n, float lr_t, float beta1,
                                        float beta2, float eps, void* stream) {
    if (!var || !m || !v || !g || n == 0 || (n & 3)) return PDA_ERR_ARG;
    const size_t n4 = n / 4;
    const size_t want = (n4 + 255) / 256;
    const unsigned blocks = (unsigned)(want < 256u * 8u ? want : 256u * 8u);  // 8 blocks/CU, grid-stride
    hipLaunchKernelGGL(adam_dense_sweep_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), var, m,
                       v, g, n4, lr_t, beta1, beta2, eps);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

extern "C" int pda_adam_dense_sweep2_f32(float* var_a, float* m_a, float* v_a, float* g_a, size_t n_a, float* var_b, float* m_b,
                                         float* v_b, float* g_b, size_t n_b, float lr_t, float beta1, float beta2, float eps,
                                         void* stream) {
    if (!var_a || !m_a || !v_a || !g_a || !var_b || !m_b || !v_b || !g_b || n_a == 0 || n_b == 0 || ((n_a | n_b) & 3)) return PDA_ERR_ARG;
    const size_t n4a = n_a / 4, n4b = n_b / 4, total = 256u * 8u;               // 8 blocks / CU in all, split by size
    size_t ba = (size_t)((double)total * (double)n4a / (double)(n4a + n4b));
    ba = ba < 1 ? 1 : (ba > total - 1 ? total - 1 : ba);
    const size_t wa = (n4a + 255) / 256, wb = (n4b + 255) / 256;
    const unsigned blocks_a = (unsigned)(wa < ba ? wa : ba), blocks_b = (unsigned)(wb < total - ba ? wb : total - ba);
    hipLaunchKernelGGL(adam_dense_sweep2_kernel, dim3(blocks_a + blocks_b), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), var_a,
                       m_a, v_a, g_a, n4a, var_b, m_b, v_b, g_b, n4b, blocks_a, lr_t, beta1, beta2, eps);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}


extern "C" int pda_adam_mark_rows(const int32_t* users, const int32_t* pos, const int32_t* neg, int B, uint32_t* touched_u, uint32_t* touched_i, void* stream) {
    if (!users || !pos || !neg || !touched_u || !touched_i || B <= 0) return PDA_ERR_ARG;
    hipLaunchKernelGGL(adam_mark_rows_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), users, pos, neg, B, touched_u,
                       touched_i);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}
extern "C" int pda_adam_dense_sweep3_f32(float* var_a, float* m_a, float* v_a, float* g_a, size_t rows_a, uint32_t* touched_a, float* var_b, float* m_b,
                                         float* v_b, float* g_b, size_t rows_b, uint32_t* touched_b, int d, float lr_t, float beta1, float beta2, float eps,
                                         void* stream) {
    if (!var_a || !m_a || !v_a || !g_a || !touched_a || !var_b || !m_b || !v_b || !g_b || !touched_b || rows_a == 0 || rows_b == 0) return PDA_ERR_ARG;
    if (d < 4 || (d & (d - 1)) != 0) return PDA_ERR_UNSUPPORTED;                // (a row is a power of two of 16-byte chunks: element -> row by a shift)
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int sh = 0;
    while ((4 << sh) < d) ++sh;
    const size_t n4a = rows_a * (size_t)(d / 4), n4b = rows_b * (size_t)(d / 4), total = 256u * 8u;               // 8 blocks / CU in all, split by size
    size_t ba = (size_t)((double)total * (double)n4a / (double)(n4a + n4b));
    ba = ba < 1 ? 1 : (ba > total - 1 ? total - 1 : ba);
    const size_t wa = (n4a + 255) / 256, wb = (n4b + 255) / 256;
    const unsigned blocks_a = (unsigned)(wa < ba ? wa : ba), blocks_b = (unsigned)(wb < total - ba ? wb : total - ba);
    hipLaunchKernelGGL(adam_dense_sweep3_kernel, dim3(blocks_a + blocks_b), dim3(256), 0, s, var_a, m_a, v_a, g_a, n4a, sh, touched_a, var_b, m_b, v_b, g_b, n4b, sh,
                       touched_b, blocks_a, lr_t, beta1, beta2, eps);
    PDA_CHECK_LAUNCH();
    // the step's marks are spent
    if (hipMemsetAsync(touched_a, 0, ((rows_a + 31) / 32) * 4, s) != hipSuccess || hipMemsetAsync(touched_b, 0, ((rows_b + 31) / 32) * 4, s) != hipSuccess)
        return PDA_ERR_LAUNCH;
    return PDA_OK;
}

static int launch_sweep4(float* var_a, float* m_a, float* v_a, float* g_a, size_t rows_a, const int32_t* tag_a, float* var_b, float* m_b, float* v_b,
                         float* g_b, size_t rows_b, const int32_t* tag_b, int d, int tag, float lr_t, float beta1, float beta2, float eps, int cache_policy,
                         hipStream_t s) {
    int sh = 0;
    while ((4 << sh) < d) ++sh;
    const size_t n4a = rows_a * (size_t)(d / 4), n4b = rows_b * (size_t)(d / 4);
    // x, m, v of both tables, read and written: what has to stay in the Infinity Cache between two steps for the plain policy to pay
    const size_t working_set = 3 * (n4a + n4b) * 16;
    const bool nt = cache_policy == PDA_ADAM_CACHE_STREAM || (cache_policy == PDA_ADAM_CACHE_AUTO && working_set > PDA_ADAM_RESIDENT_BYTES);
    // 7 workgroups per CU (the kernel's 71 registers: all resident at once), two chunks per thread, grid-stride.  Measured at C2 (1.1 M chunks, the
    // sweep alone): 1 024 ... 4 608 workgroups x 1 / 2 chunks 15.4 - 17.1 us, four chunks per thread with every load in flight at once 18.3 us -- the
    // sweep sits at ~6.7 TB/s of its algorithmic bytes whatever the geometry (profiles/round6_adam_small.txt).
    const size_t total = 256u * 7u;
    size_t ba = (size_t)((double)total * (double)n4a / (double)(n4a + n4b));
    ba = ba < 1 ? 1 : (ba > total - 1 ? total - 1 : ba);
    const size_t wa = (n4a + 255) / 256, wb = (n4b + 255) / 256;
    const unsigned blocks_a = (unsigned)(wa < ba ? wa : ba), blocks_b = (unsigned)(wb < total - ba ? wb : total - ba);
#define PDA_SWEEP4(NTV, UNV)                                                                                                                          \
    hipLaunchKernelGGL((adam_dense_sweep4_kernel<NTV, UNV>), dim3(blocks_a + blocks_b), dim3(256), 0, s, var_a, m_a, v_a, g_a, n4a, tag_a, var_b, m_b, v_b, \
                       g_b, n4b, tag_b, sh, tag, blocks_a, lr_t, beta1, beta2, eps)
    if (nt) PDA_SWEEP4(true, 2);
    else PDA_SWEEP4(false, 2);
#undef PDA_SWEEP4
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

extern "C" int pda_adam_dense_sweep4_f32(float* var_a, float* m_a, float* v_a, float* g_a, size_t rows_a, const int32_t* tag_a, float* var_b, float* m_b,
                                         float* v_b, float* g_b, size_t rows_b, const int32_t* tag_b, int d, int tag, float lr_t, float beta1, float beta2,
                                         float eps, int cache_policy, void* stream) {
    if (!var_a || !m_a || !v_a || !g_a || !tag_a || !var_b || !m_b || !v_b || !g_b || !tag_b || rows_a == 0 || rows_b == 0) return PDA_ERR_ARG;
    if (d < 4 || (d & (d - 1)) != 0) return PDA_ERR_UNSUPPORTED;
    if (cache_policy < PDA_ADAM_CACHE_AUTO || cache_policy > PDA_ADAM_CACHE_STREAM) return PDA_ERR_ARG;
    return launch_sweep4(var_a, m_a, v_a, g_a, rows_a, tag_a, var_b, m_b, v_b, g_b, rows_b, tag_b, d, tag, lr_t, beta1, beta2, eps, cache_policy,
                         reinterpret_cast<hipStream_t>(stream));
}

// One reference train step (MF/model_api.py:83 minimize = gradients + TF-1.14 dense-decay Adam) in TWO launches: the step kernel sums the batch's
// gradients into gU / gI and tags the rows it touched, the sweep applies Adam to every row of both tables (g = 0 off the tagged rows) and zeroes
// g behind itself.
extern "C" int pda_adam_step_f32(float* U, float* mU, float* vU, float* gU, int32_t* tagU, size_t n_users, float* I, float* mI, float* vI, float* gI,
                                 int32_t* tagI, size_t n_items, const int32_t* users, const int32_t* pos, const int32_t* neg, const float* pos_pop,
                                 const float* neg_pop, int B, int d, float regs, float reg_div, int step_tag, float lr_t, float beta1, float beta2, float eps,
                                 int flags, int cache_policy, float* loss_acc, void* stream) {
    if (!U || !mU || !vU || !gU || !tagU || !I || !mI || !vI || !gI || !tagI || !users || !pos || !neg || B <= 0 || reg_div <= 0.f || n_users == 0 ||
        n_items == 0 || step_tag <= 0)
        return PDA_ERR_ARG;
    if ((pos_pop == nullptr) != (neg_pop == nullptr)) return PDA_ERR_ARG;
    if (flags & ~(PDA_UPD_ANY_ORDER | PDA_UPD_USERS_DISTINCT)) return PDA_ERR_ARG;
    if (cache_policy < PDA_ADAM_CACHE_AUTO || cache_policy > PDA_ADAM_CACHE_STREAM) return PDA_ERR_ARG;
    if (d != 32 && d != 64 && d != 128 && d != 256) return PDA_ERR_UNSUPPORTED;
    StepArgs a{U, I, users, pos, neg, pos_pop, neg_pop, nullptr, nullptr, nullptr, gU, gI, loss_acc,
               B, 1.0f / (float)B, regs / reg_div, 0.f, PDA_UPD_DENSE_GRAD, 0, d, U, I, (flags & PDA_UPD_ANY_ORDER) ? 1 : 0,
               (flags & PDA_UPD_USERS_DISTINCT) ? 1 : 0, tagU, tagI, step_tag};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc;
    switch (d) {
        case 32: rc = launch_step<32>(a, s); break;
        case 64: rc = launch_step<64>(a, s); break;
        case 128: rc = launch_step<128>(a, s); break;
        default: rc = launch_step<256>(a, s); break;
    }
    if (rc != PDA_OK) return rc;
    return launch_sweep4(U, mU, vU, gU, n_users, tagU, I, mI, vI, gI, n_items, tagI, d, step_tag, lr_t, beta1, beta2, eps, cache_policy, s);
}

extern "C" int pda_adam_rows_f32(float* var, float* m, float* v, float* g, const int32_t* rows, int n_rows, int d,
                                 float lr_t, float beta1, float beta2, float eps, void* stream) {
    if (!var || !m || !v || !g || !rows || n_rows <= 0) return PDA_ERR_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define PDA_ROWS(DD)                                                                                              \
    case DD: {                                                                                                    \
        constexpr int RPB = 256 / (DD / 4);                                                                       \
        hipLaunchKernelGGL(adam_rows_kernel<DD>, dim3((unsigned)((n_rows + RPB - 1) / RPB)), dim3(256), 0, s, var, m, v, \
                           g, rows, n_rows, lr_t, beta1, beta2, eps);                                             \
        break;                                                                                                    \
    }
    switch (d) {
        PDA_ROWS(32) PDA_ROWS(64) PDA_ROWS(128) PDA_ROWS(256)
        default: return PDA_ERR_UNSUPPORTED;
    }
#undef PDA_ROWS
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

static int run_adam_lazy(int phase, bool fast, float* U, float* mU, float* vU, float* gU, int32_t* lastU, float* I, float* mI, float* vI, float* gI,
                         int32_t* lastI, const int32_t* users, const int32_t* pos, const int32_t* neg, int B, int d, int t, const int32_t* t_dev,
                         int32_t* t_next, const float* lr_tab, int n_tab, float beta1, float beta2, float eps, hipStream_t s);

// The same with the step in DEVICE memory: t = t_dev[0] is read by the kernel, phase 1 stores t + 1 into t_next (the other of two
// counter slots; the caller alternates them from step to step), so a captured HIP graph of steps advances on every replay.
extern "C" int pda_adam_lazy_dev_f32(int phase, float* U, float* mU, float* vU, float* gU, int32_t* lastU, float* I, float* mI, float* vI,
                                     float* gI, int32_t* lastI, const int32_t* users, const int32_t* pos, const int32_t* neg, int B, int d,
                                     const int32_t* t_dev, int32_t* t_next, const float* lr_tab, int n_tab, float beta1, float beta2, float eps,
                                     void* stream) {
    if (!U || !mU || !vU || !gU || !lastU || !I || !mI || !vI || !gI || !lastI || !users || !pos || !neg || !lr_tab || !t_dev || !t_next) return PDA_ERR_ARG;
    const bool fast = (phase & PDA_ADAM_REPLAY_FAST) != 0;
    phase &= ~PDA_ADAM_REPLAY_FAST;
    if (B <= 0 || n_tab < 2 || t_dev == t_next || (phase != 0 && phase != 1)) return PDA_ERR_ARG;
    return run_adam_lazy(phase, fast, U, mU, vU, gU, lastU, I, mI, vI, gI, lastI, users, pos, neg, B, d, 1, t_dev, t_next, lr_tab, n_tab, beta1, beta2, eps,
                         reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pda_adam_lazy_f32(int phase, float* U, float* mU, float* vU, float* gU, int32_t* lastU, float* I, float* mI, float* vI,
                                 float* gI, int32_t* lastI, const int32_t* users, const int32_t* pos, const int32_t* neg, int B, int d, int t,
                                 const float* lr_tab, float beta1, float beta2, float eps, void* stream) {
    if (!U || !mU || !vU || !gU || !lastU || !I || !mI || !vI || !gI || !lastI || !users || !pos || !neg || !lr_tab) return PDA_ERR_ARG;
    const bool fast = (phase & PDA_ADAM_REPLAY_FAST) != 0;
    phase &= ~PDA_ADAM_REPLAY_FAST;
    if (B <= 0 || t < 1 || (phase != 0 && phase != 1)) return PDA_ERR_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return run_adam_lazy(phase, fast, U, mU, vU, gU, lastU, I, mI, vI, gI, lastI, users, pos, neg, B, d, t, nullptr, nullptr, lr_tab, 0, beta1, beta2, eps, s);
}

static int run_adam_lazy(int phase, bool fast, float* U, float* mU, float* vU, float* gU, int32_t* lastU, float* I, float* mI, float* vI, float* gI,
                         int32_t* lastI, const int32_t* users, const int32_t* pos, const int32_t* neg, int B, int d, int t, const int32_t* t_dev,
                         int32_t* t_next, const float* lr_tab, int n_tab, float beta1, float beta2, float eps, hipStream_t s) {
    const LazyAdamArgs a{U, mU, vU, gU, lastU, I, mI, vI, gI, lastI, users, pos, neg, lr_tab, B, t, phase, beta1, beta2, eps, t_dev, t_next, n_tab};
    const FastConsts fc{(float)sqrt((double)beta2), (float)log2((double)beta1), (float)log2((double)beta2)};
#define PDA_LAZY(DD)                                                                                                   \
    case DD: {                                                                                                         \
        constexpr int RPB = 256 / (DD / 4);                                                                            \
        const dim3 grid((unsigned)((3 * (size_t)B + RPB - 1) / RPB));                                                  \
        if (fast) hipLaunchKernelGGL((adam_lazy_kernel<DD, true>), grid, dim3(256), 0, s, a, fc);                      \
        else hipLaunchKernelGGL((adam_lazy_kernel<DD, false>), grid, dim3(256), 0, s, a, fc);                          \
        break;                                                                                                         \
    }
    switch (d) {
        PDA_LAZY(32) PDA_LAZY(64) PDA_LAZY(128) PDA_LAZY(256)
        default: return PDA_ERR_UNSUPPORTED;
    }
#undef PDA_LAZY
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

static int run_lazy_sync(float* var, float* m, float* v, int32_t* last, size_t n_rows, int d, int t, const float* lr_tab,
                         float beta1, float beta2, float eps, bool fast, void* stream) {
    if (!var || !m || !v || !last || !lr_tab || n_rows == 0 || t < 0) return PDA_ERR_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const FastConsts fc{(float)sqrt((double)beta2), (float)log2((double)beta1), (float)log2((double)beta2)};
#define PDA_LSYNC(DD)                                                                                                  \
    case DD: {                                                                                                         \
        constexpr int RPB = 256 / (DD / 4);                                                                            \
        const size_t nb = (n_rows + RPB - 1) / RPB;                                                                    \
        const dim3 grid((unsigned)(nb < 16384 ? nb : 16384));                                                          \
        if (fast) hipLaunchKernelGGL((adam_lazy_sync_kernel<DD, true>), grid, dim3(256), 0, s, var, m, v, last, n_rows, t, lr_tab, beta1, beta2, eps, fc); \
        else hipLaunchKernelGGL((adam_lazy_sync_kernel<DD, false>), grid, dim3(256), 0, s, var, m, v, last, n_rows, t, lr_tab, beta1, beta2, eps, fc);    \
        break;                                                                                                         \
    }
    switch (d) {
        PDA_LSYNC(32) PDA_LSYNC(64) PDA_LSYNC(128) PDA_LSYNC(256)
        default: return PDA_ERR_UNSUPPORTED;
    }
#undef PDA_LSYNC
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}
extern "C" int pda_adam_lazy_sync_f32(float* var, float* m, float* v, int32_t* last, size_t n_rows, int d, int t, const float* lr_tab,
                                      float beta1, float beta2, float eps, void* stream) {
    return run_lazy_sync(var, m, v, last, n_rows, d, t, lr_tab, beta1, beta2, eps, false, stream);
}
extern "C" int pda_adam_lazy_sync_fast_f32(float* var, float* m, float* v, int32_t* last, size_t n_rows, int d, int t, const float* lr_tab,
                                           float beta1, float beta2, float eps, void* stream) {
    return run_lazy_sync(var, m, v, last, n_rows, d, t, lr_tab, beta1, beta2, eps, true, stream);
}

extern "C" int pda_bpr_step_bf16(const uint16_t* U_bf16, const uint16_t* I_bf16, float* U_master, float* I_master,
                                 const int32_t* users, const int32_t* pos, const int32_t* neg, const float* pos_pop,
                                 const float* neg_pop, int B, int d, float regs, float reg_div, float lr, int update_mode,
                                 float* g_user, float* g_pos, float* g_neg, float* gU, float* gI, float* loss_acc, void* stream) {
    if (!U_bf16 || !I_bf16 || !users || !pos || !neg || B <= 0 || reg_div <= 0.f) return PDA_ERR_ARG;
    if ((pos_pop == nullptr) != (neg_pop == nullptr)) return PDA_ERR_ARG;
    const int any_order = (update_mode & PDA_UPD_ANY_ORDER) ? 1 : 0;
    [[maybe_unused]] const int users_distinct = (update_mode & PDA_UPD_USERS_DISTINCT) ? 1 : 0;
    update_mode &= ~(PDA_UPD_ANY_ORDER | PDA_UPD_USERS_DISTINCT);
    if (update_mode < PDA_UPD_NONE || update_mode > PDA_UPD_DENSE_GRAD) return PDA_ERR_ARG;
    if (update_mode == PDA_UPD_DENSE_GRAD && (!gU || !gI)) return PDA_ERR_ARG;
    if (update_mode == PDA_UPD_SGD_FUSED && (!U_master || !I_master)) return PDA_ERR_ARG;   // bf16 rows take no atomics
    if (g_user && (!g_pos || !g_neg)) return PDA_ERR_ARG;
    StepArgs a{U_master, I_master, users, pos, neg, pos_pop, neg_pop, g_user, g_pos, g_neg, gU, gI, loss_acc,
               B, 1.0f / (float)B, regs / reg_div, lr, update_mode, 0, d, U_bf16, I_bf16, any_order};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (d) {
        case 32: return launch_step<32, true>(a, s);
        case 64: return launch_step<64, true>(a, s);
        case 128: return launch_step<128, true>(a, s);
        case 256: return launch_step<256, true>(a, s);
        default: return PDA_ERR_UNSUPPORTED;
    }
}

extern "C" int pda_refresh_rows_bf16(const float* master, uint16_t* shadow, const int32_t* rows, int n_rows, int d, void* stream) {
    if (!master || !shadow || n_rows <= 0 || d <= 0 || (d & 7)) return PDA_ERR_ARG;
    const size_t n8 = (size_t)n_rows * (d / 8);
    hipLaunchKernelGGL(refresh_rows_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       master, shadow, rows, 0, n8, d / 8);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}
