/* pda_hip_experimental.h -- entry points of libpda_hip.so BEYOND the stable drop-in surface of pda_hip.h.
 *
 * pda_hip.h is what SURVEY.md section 8(b) asks a replacement of the reference's path to export: the plan, the item preps, the score + mask + top-K
 * calls the plan may name, the merge, the train step, the reference's optimiser, the metrics and the sampler.  A maintainer of the reference
 * binds THAT header only.
 *
 * This header declares what pda_amd itself uses on top of it and what the benchmarks measure -- the two phases and the seed exchange of the
 * item-sharded evaluation (pda_amd/dist.py), the split rules behind the plan, the planned / looped train steps, the six-stream sweep and the
 * lazy replay of Adam, the sampler's look-ahead, measured peaks.  Same conventions (device pointers, explicit stream, int return codes),
 * NO stability promise: signatures here may change with PDA_ABI_VERSION unchanged.
 */
#ifndef PDA_HIP_EXPERIMENTAL_H
#define PDA_HIP_EXPERIMENTAL_H

#include "pda_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- item splits: the rules behind pda_score_topk_plan ------------------------------------------------------------------------------- */
/* item splits of the huge geometry for a block (0: the block is too small for it) -- the rule behind PDA_PATH_GEN4 plans of dense sweeps */
int pda_score_topk_huge_splits(int n_users_blk, int n_items_local, int d);

/* Item-sharded evaluation with exact early termination: the two phases of pda_score_topk4_* as separate calls and a seed.
 *   phase 1  the exact warm-up only: out_keys holds every split's list of its first 64 warm_tiles visited items
 *            (warm_tiles 1 .. 4, 0 = 4; the same value in both phases: with R shards the R warm-ups together cover
 *            R x 64 warm_tiles items, so a sharded run wants fewer per shard)
 *   pda_topk_kth_value(out_keys, n_splits, n_users_blk, K, pos, tau, stream)   tau f32 [n_users_blk]: the value at rank pos
 *            (0-based) of a user's warm-up lists (max over the splits; -inf while no list is that long).  Two bounds of a
 *            user's FINAL K-th value over R item shards: the MAXIMUM over the shards of their values at pos = K - 1, and the
 *            MINIMUM over the shards of their values at pos = ceil(K / R) - 1 (R shards with ceil(K / R) items above it);
 *            two all-reduces of 4 bytes per user, seed = the larger of the two
 *   phase 2  the sweep, seed = that maximum (or NULL): pairs whose exact score is below the seed stay out of this shard's
 *            list -- they cannot be in the merged top K -- and the early termination prunes against max(own K-th value, seed).
 *            A shard's list may then end with fewer than K entries (empty slots = 0); the merge of the shards' lists is exactly
 *            the top K of the whole catalogue.
 *   pda_topk_seed_refine(out_keys, n_splits, n_users_blk, K, lo, hi, mid, counts, mode, stream)   optional, between the two
 *            phases: rounds of a bisection between the seed (lo) and the MAXIMUM over the shards of their ceil(K / R)-th
 *            warm-up value (hi: some shard holds ceil(K / R) of the merged top K, so this bounds their K-th value from above).  mode 0: mid := (lo + hi) / 2, counts[u] := this shard's warm-up entries >= mid[u]; the caller
 *            all-reduces counts (SUM); mode 1: lo/hi updated from the summed counts (>= K entries at or above mid make it
 *            a bound), next mid and counts; mode 2: the last update only.  Three rounds bring eight shards of config 3 from
 *            1.78 x to 1.03 x the tiles of one GPU (4 more bytes per user and round).
 *   phase 4  (round 4) the sweep of the WHOLE shard from EMPTY lists against the caller's seed (required; -inf = no bound for that
 *            user): no warm-up ran on this catalogue -- it ran elsewhere, on replicated hot items (pda_amd/dist.py: the 256 globally
 *            most popular rows live on every rank and are taken OUT of the shards; a rank warms up 1 / R of the users on them and
 *            the K-th values are all-gathered as the seed).  out_keys needs no initialisation; warm_tiles is ignored.
 * Without it every rank prunes against its own shard's K-th value only and scores 8 x 32 % instead of 3.9 % of the
 * catalogue (config 3, eight shards).  n_splits must be the same (> 0) in both phases. */
int pda_topk_kth_value(const uint64_t* keys, int n_splits, int n_users_blk, int K, int pos, float* out, void* stream);
/* Packed keys whose item field holds LOCAL row ids of a gathered table -> the same keys with gid[local] in it, in place (empty slots
 * stay 0; n_keys keys, gid int32 [n_gid]).  The replicated-hot-items path of pda_amd/dist.py scores a hot table and a cold shard
 * that are row subsets of the catalogue; gid ascends with the local id, so sorted lists stay sorted, ties included. */
int pda_topk_remap_items(uint64_t* keys, size_t n_keys, const int32_t* gid, int n_gid, void* stream);
/* The same exchange in TWO collectives per user block (round 3; replaces kth_value x 3 + MAX + MIN + three sequential SUM rounds):
 *   pda_topk_seed_bounds  bounds f32 [3][n_users_blk] := (value at rank K - 1, value at rank m - 1, MINUS the value at rank
 *            m - 1) of the shard's warm-up lists, m = ceil(K / R).  ONE all-reduce MAX over the 3 n_users_blk floats (min x =
 *            -max -x).  A rank without items contributes (-inf, -inf, +inf).
 *   pda_topk_seed_counts  counts i32 [n_thr][n_users_blk] := this shard's warm-up entries at or above the n_thr common thresholds
 *            lo + (hi - lo) (j + 1) / (n_thr + 1) between lo = max(bounds 0, -bounds 2) and hi = bounds 1 (the grid that
 *            `rounds` bisection rounds walk, n_thr = 2^rounds - 1 <= 15).  ONE all-reduce SUM.
 *   pda_topk_seed_pick    seed[u] := the largest threshold with a summed count >= K, else lo (n_thr = 0: lo).
 * Both collectives are a few bytes per user and depend on the warm-up only: a caller with several user blocks issues them for
 * block b + 1 on a side stream under the sweep of block b (pda_amd/dist.py). */
int pda_topk_seed_bounds(const uint64_t* keys, int n_splits, int n_users_blk, int K, int m, float* bounds, void* stream);
int pda_topk_seed_counts(const uint64_t* keys, int n_splits, int n_users_blk, int K, const float* bounds, int n_thr, int32_t* counts,
                         void* stream);
int pda_topk_seed_pick(const float* bounds, const int32_t* counts, int n_thr, int n_users_blk, int K, float* seed, void* stream);
int pda_topk_seed_refine(const uint64_t* keys, int n_splits, int n_users_blk, int K, float* lo, float* hi, float* mid, int32_t* counts,
                         int mode, void* stream);
int pda_score_topk4_phase_f32(const float* U, const float* I_shard, const void* prep, const float* pop_shard, const int32_t* users,
                              int n_users_blk, int item_offset, int n_items_local, int d, const int64_t* hist_indptr,
                              const int32_t* hist_indices, int hist_row_mode, int K, int head, int early_stop, int n_splits,
                              int phase, int warm_tiles, const float* seed, uint64_t* out_keys, void* workspace, void* stream);
int pda_score_topk4_phase_bf16(const uint16_t* U, const uint16_t* I_shard, const void* prep, const float* pop_shard,
                               const int32_t* users, int n_users_blk, int item_offset, int n_items_local, int d,
                               const int64_t* hist_indptr, const int32_t* hist_indices, int hist_row_mode, int K, int head,
                               int early_stop, int n_splits, int phase, int warm_tiles, const float* seed, uint64_t* out_keys,
                               void* workspace, void* stream);

/* ---- train step variants -------------------------------------------------------------------------------------------------------------- */
/* ---- The exact mini-batch SGD step without atomics (round 3; pda_bpr_plan.hip): plan + two launches -------------------------
 * The reference applies the SUM of a batch's gradients, all computed from the tables as they stood (MF/model_api.py:83,102-121;
 * IndexedSlices are summed per row [TF-ext]).  PDA_UPD_SGD_FUSED above is hogwild inside a batch and spends 3 d fp32 atomics per
 * triplet; this path is exact, bit-reproducible and writes every touched row ONCE with plain stores.
 *   pda_triplet_plan(users, pos, neg, B, n_batches, plans)   one workgroup per batch ([n_batches, B] arrays, B <= 4096): the 2B item
 *            references pos ++ neg sorted by item (LDS counting sort) -> segments of equal item, two bits per triplet ("my positive /
 *            negative is referenced once in the batch"), and a flag "a user occurs twice".  plans: n_batches x
 *            pda_triplet_plan_bytes(B) bytes.  The plan depends on the ids only: a sampler computes it batches ahead of the step.
 *   pda_bpr_step_plan_f32(..., plan, scratch, exact, loss_acc)
 *            exact = 1: launch A (per triplet) gathers, computes loss and the triplet's two coefficients, moves the USER row with
 *            a plain store (users are distinct inside a batch -- the sampler contract, rd.sample at MF/train_new_api.py:380-381)
 *            and leaves the old user row + coefficients in scratch (pda_bpr_step_plan_scratch_bytes(B, d) bytes); launch B (per
 *            distinct item row) sums coefficient x old user row over the row's references in plan order, adds the L2 term, and
 *            stores the row.  No gather of either launch can see a row of this batch already moved.
 *            exact = 0: ONE launch: user rows and once-referenced item rows take plain stores, shared item rows keep the atomics
 *            of PDA_UPD_SGD_FUSED (hogwild on those rows only).  fp32 tables only.
 *            A batch in which a user occurs twice is REJECTED by the plan: loss_acc receives NaN and no table is written
 *            (use PDA_UPD_NONE + pda_sgd_apply_f32 for such batches).
 *   pda_bpr_step_plan_bf16  the exact step on bf16 tables (config 5): forward pass on the bf16 rows, update on the fp32 masters,
 *            the touched bf16 rows re-rounded (RNE) by the same two launches (no pda_refresh_rows_bf16 afterwards). */
/*   pda_triplet_plan_large(users, pos, neg, B, plan, workspace)   the same plan (same layout, same bytes) of ONE batch of any size
 *            (pda_bpr_plan_large.hip: a device-wide stable radix sort of the 2B references instead of one workgroup's LDS) -- what
 *            lets the exact step run at batch sizes where the launch floor no longer matters (B = 32 768 at config 2: see
 *            profiles/round3_train_b_sweep.txt).  workspace: pda_triplet_plan_large_workspace_bytes(B) bytes. */
size_t pda_triplet_plan_bytes(int B);
size_t pda_triplet_plan_large_workspace_bytes(int B);
int pda_triplet_plan_large(const int32_t* users, const int32_t* pos, const int32_t* neg, int B, void* plan, void* workspace, void* stream);
size_t pda_bpr_step_plan_scratch_bytes(int B, int d);
int pda_triplet_plan(const int32_t* users, const int32_t* pos, const int32_t* neg, int B, int n_batches, void* plans, void* stream);
int pda_bpr_step_plan_f32(float* U, float* I, const int32_t* users, const int32_t* pos, const int32_t* neg, const float* pos_pop,
                          const float* neg_pop, int B, int d, float regs, float reg_div, float lr, const void* plan, float* scratch,
                          int exact, float* loss_acc, void* stream);
int pda_bpr_step_plan_bf16(uint16_t* U_bf16, uint16_t* I_bf16, float* U_master, float* I_master, const int32_t* users,
                           const int32_t* pos, const int32_t* neg, const float* pos_pop, const float* neg_pop, int B, int d,
                           float regs, float reg_div, float lr, const void* plan, float* scratch, float* loss_acc, void* stream);

/* Reorder one batch (all five arrays, in place) so that equal positives are adjacent: pda_bpr_step_f32 then sums each
 * run on chip before touching HBM.  Purely a performance aid (order inside a batch has no meaning); B <= 4096. */
int pda_sort_triplets_by_pos(int32_t* users, int32_t* pos, int32_t* neg, float* pos_pop, float* neg_pop, int B,
                             void* stream);

/* The same purpose, without a sorting network: the batch is permuted so that every run of equal positives is contiguous
 * (order: hash bucket of pos, then pos, then original index -- deterministic).  ~5x faster than the sort; what the device
 * sampler uses.  B <= 4096. */
int pda_group_triplets_by_pos(int32_t* users, int32_t* pos, int32_t* neg, float* pos_pop, float* neg_pop, int B,
                              void* stream);

/* ---- the reference's optimiser, other forms ------------------------------------------------------------------------------------------ */
/* The same dense-decay step as SIX streams instead of seven (round 5): the gradient tables are zero on all but the batch's rows, so they are read
 * (and cleared) only where a row's bit is set.  pda_adam_mark_rows sets the bits of a batch (users -> touched_u; pos, neg -> touched_i; bitmaps of
 * ceil(rows / 32) words, zero before the first step); pda_adam_dense_sweep3_f32 sweeps both tables (d a power of two) and clears the bits behind
 * itself.  Bit-identical tables to pda_adam_dense_sweep2_f32 (an untouched row computes with g = 0, operation for operation).  MF/model_api.py:83.
 * PRECONDITIONS: users / pos / neg hold B valid row ids each (the kernel does no bounds check: an id beyond the tables writes beyond the bitmaps);
 * g_a / g_b are ZERO on every row whose bit is not set -- a caller that accumulates gradients for rows outside (users, pos, neg) (micro-batches,
 * gradients reduced from other ranks) must mark those rows too, or use pda_adam_dense_sweep2_f32, which reads every row's gradient.
 * Superseded for single-GPU training by pda_adam_step_f32 (pda_hip.h: row tags instead of bitmaps, two launches instead of five). */
int pda_adam_mark_rows(const int32_t* users, const int32_t* pos, const int32_t* neg, int B, uint32_t* touched_u, uint32_t* touched_i, void* stream);
int pda_adam_dense_sweep3_f32(float* var_a, float* m_a, float* v_a, float* g_a, size_t rows_a, uint32_t* touched_a, float* var_b, float* m_b, float* v_b,
                              float* g_b, size_t rows_b, uint32_t* touched_b, int d, float lr_t, float beta1, float beta2, float eps, void* stream);

/* Lazy/sparse Adam on the touched rows only (declared deviation).  rows i32 [n_rows]
 * must be unique; g is the dense accumulator (reset on the touched rows). */
int pda_adam_rows_f32(float* var, float* m, float* v, float* g, const int32_t* rows, int n_rows, int d, float lr_t,
                      float beta1, float beta2, float eps, void* stream);

/* The SAME optimiser without the sweep (exact, not the lazy deviation above): a row whose gradient is zero at step k only
 * decays -- m <- b1 m, v <- b2 v, x <- x - lr_k m / (sqrt(v) + eps), the arithmetic of pda_adam_dense_sweep_f32 with g = 0 --
 * so it may skip its idle steps and replay them in registers when it is next needed.  last i32 [rows]: the step a row is
 * current for (0 at the start); lr_tab f32 [>= t + 1]: lr_tab[k] = the bias-corrected rate of step k (index 0 unused).
 * Per training step t (1-based), around pda_bpr_step_f32(PDA_UPD_DENSE_GRAD, gU, gI):
 *   pda_adam_lazy_f32(phase = 0, ...)   before it: every row of the batch (users, pos, neg; repeats allowed) is brought to t - 1
 *   pda_adam_lazy_f32(phase = 1, ...)   after it: every row of the batch takes step t with its summed gradient; its
 *                                       accumulator row is cleared
 *   pda_adam_lazy_sync_f32(table, t)    every row of a table up to step t: before an evaluation or a checkpoint
 * After the sync the tables and moments equal those of t dense sweeps bit for bit (tests/test_gpu_bpr_step.py).  Traffic
 * per step: the batch rows, instead of 3.7 GB (config 3) or 74 GB (config 5). */
int pda_adam_lazy_f32(int phase, float* U, float* mU, float* vU, float* gU, int32_t* lastU, float* I, float* mI, float* vI,
                      float* gI, int32_t* lastI, const int32_t* users, const int32_t* pos, const int32_t* neg, int B, int d, int t,
                      const float* lr_tab, float beta1, float beta2, float eps, void* stream);
int pda_adam_lazy_sync_f32(float* var, float* m, float* v, int32_t* last, size_t n_rows, int d, int t, const float* lr_tab,
                           float beta1, float beta2, float eps, void* stream);
/* The same catch-up to the north_star's tolerance instead of bit for bit (round 3): phase | PDA_ADAM_REPLAY_FAST resp.
 * pda_adam_lazy_sync_fast_f32.  sqrt(v_k) as a running product, the division a hardware reciprocal, the loop over a row's idle
 * steps ends when its geometrically falling terms can no longer move x, and m, v take their closed-form powers: x within 1e-6
 * of the exact replay / the dense sweep, m and v within 1e-4 relative, over idle gaps of thousands of steps
 * (tests/test_gpu_bpr_step.py); ~6 instead of ~35 VALU per element and replayed step, and a few dozen instead of ~180 steps. */
/* pda_adam_lazy_dev_f32: pda_adam_lazy_f32 with the step in DEVICE memory (t = t_dev[0]; phase 1 stores t + 1 into t_next, the
 * other of two int32 counter slots which the caller alternates from step to step; lr_tab holds n_tab entries, a step beyond it
 * leaves the tables alone): the three launches of a training step -- phase 0, pda_bpr_step_f32(PDA_UPD_DENSE_GRAD), phase 1 --
 * can be captured into a HIP graph whose replays advance the optimiser. */
int pda_adam_lazy_dev_f32(int phase, float* U, float* mU, float* vU, float* gU, int32_t* lastU, float* I, float* mI, float* vI,
                          float* gI, int32_t* lastI, const int32_t* users, const int32_t* pos, const int32_t* neg, int B, int d,
                          const int32_t* t_dev, int32_t* t_next, const float* lr_tab, int n_tab, float beta1, float beta2, float eps,
                          void* stream);
#define PDA_ADAM_REPLAY_FAST 0x10
int pda_adam_lazy_sync_fast_f32(float* var, float* m, float* v, int32_t* last, size_t n_rows, int d, int t, const float* lr_tab,
                                float beta1, float beta2, float eps, void* stream);

/* ---- sampler look-ahead, the train loop on the device --------------------------------------------------------------------------------- */
/* Train step of batch t and sampler of batch t + 1 in ONE launch (spare workgroups of the step kernel draw the next
 * batch): the two are independent and latency-bound, and as separate launches -- also on two captured streams -- they run
 * back to back.  `next` (host memory, read during the call) = the arguments of pda_sample_triplets_dev; its output arrays
 * must be a second set of batch buffers, not the ones this step reads.  Step arguments as pda_bpr_step_f32 restricted to
 * PDA_UPD_SGD_FUSED / PDA_UPD_NONE (| PDA_UPD_ANY_ORDER).  Equivalent to pda_bpr_step_f32 followed by
 * pda_sample_triplets_dev on the same stream.  Reference: the generator thread that samples while session.run trains
 * (MF/train_new_api.py:178-220, 260-288). */
typedef struct pda_sample_job {
    int32_t* users; int gen_users; const int32_t* user_pool; int n_pool; int B;
    const int64_t* train_indptr; const int32_t* train_indices; const int32_t* train_slots;
    int neg_lo, neg_hi; const float* pop_matrix; int n_slots; uint64_t seed;
    const uint64_t* step_dev; uint64_t* step_next;
    int32_t* pos; int32_t* neg; float* pos_pop; float* neg_pop;
} pda_sample_job;
int pda_bpr_step_sample_f32(float* U, float* I, const int32_t* users, const int32_t* pos, const int32_t* neg,
                            const float* pos_pop, const float* neg_pop, int B, int d, float regs, float reg_div, float lr,
                            int update_mode, float* loss_acc, const pda_sample_job* next, void* stream);

/* The sampler n batches ahead (the reference's generator thread keeps a queue of batches, MF/train_new_api.py:178-220): ONE
 * launch draws the batches of steps *step_dev .. *step_dev + n_batches - 1 into row j of [n_batches][B] buffers -- bit for bit
 * what n_batches pda_sample_triplets_dev calls draw -- and stores *step_dev + n_batches to step_next (!= step_dev);
 * group_by_pos != 0 (B <= 4096) groups every batch by positive item in a second launch, one workgroup per batch
 * (pda_group_triplets_by_pos_batches).  64 batches ahead: 9.6 instead of 13.9 us per 2048-triplet step with a fresh batch. */
int pda_sample_batches_dev(int32_t* users, int gen_users, const int32_t* user_pool, int n_pool, int B, int n_batches,
                           const int64_t* train_indptr, const int32_t* train_indices, const int32_t* train_slots, int neg_lo,
                           int neg_hi, const float* pop_matrix, int n_slots, uint64_t seed, const uint64_t* step_dev,
                           uint64_t* step_next, int32_t* pos, int32_t* neg, float* pos_pop, float* neg_pop, int group_by_pos,
                           void* stream);
int pda_group_triplets_by_pos_batches(int32_t* users, int32_t* pos, int32_t* neg, float* pos_pop, float* neg_pop, int B,
                                      int n_batches, void* stream);

/* n_steps fused SGD steps in ONE launch (the session.run loop of MF/train_new_api.py:1078-1096 with the generator thread
 * sampling ahead): a resident grid loops on the device; iteration i steps on the batch in buffer set i & 1 while spare
 * workgroups draw the next batch into set (i + 1) & 1; a grid barrier separates the iterations.  set0 / set1: the two sets of
 * batch buffers with the sampler's configuration (step_dev / step_next unused); set0 must hold the first batch on entry, and
 * after the call set (n_steps & 1) holds the batch of the next step.  step_ctr u64 (device): the sampler step of the first
 * batch drawn here, advanced by n_steps.  loss_steps f32 [n_steps][3] (NULL: everything is summed into loss_acc [3]).
 * barrier_ws: 8 device bytes, zeroed by the caller before the FIRST call; word 0 is the arrival counter (reset by every call),
 * word 1 is STICKY: != 0 means some launch since the caller last cleared it found the grid not resident as a whole (another
 * kernel held CUs) and abandoned its loop -- that launch trained nothing and did not advance step_ctr.  Read it whenever the
 * stream is next synchronised.  update_mode: PDA_UPD_SGD_FUSED (| PDA_UPD_ANY_ORDER).  Equals n_steps x
 * (pda_bpr_step_f32, pda_sample_triplets_dev) on one stream (fp32 atomics: 1e-6).  The grid is at most 384 + 64
 * workgroups, striding over larger batches. */
int pda_bpr_train_steps_f32(float* U, float* I, int d, float regs, float reg_div, float lr, int update_mode,
                            const pda_sample_job* set0, const pda_sample_job* set1, uint64_t* step_ctr, int n_steps,
                            float* loss_acc, float* loss_steps, void* barrier_ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Peaks measured on the box (no reference counterpart; BASELINE.md section 4 asks for measured roofs).
 *   pda_peak_mfma_bf16: one launch of iters x 16 v_mfma_f32_32x32x16_bf16 per wave on 8192 waves (two per SIMD,
 *        four accumulator chains each, register operands); the caller times it (HIP events) and divides
 *        pda_peak_mfma_flops_per_launch(iters) by the duration.  sink: any 4 device bytes (never written).
 *   pda_peak_copy: dst[0..n) = src[0..n), float4 grid-stride; 2 * 4 n bytes of HBM traffic per launch.
 * ------------------------------------------------------------------------------------------------ */
double pda_peak_mfma_flops_per_launch(int iters);
int pda_peak_mfma_bf16(float* sink, int iters, void* stream);
/* The same loop with every operand 1.0: the chip is power-limited, and a matrix pipe that toggles nothing clocks higher (the
 * micro-architecture guide's 2 495 TFLOP/s is this kind of figure; random mantissas reach ~2 050 on the same box). */
int pda_peak_mfma_bf16_const(float* sink, int iters, void* stream);
/* The roof of the sweep's own KIND of loop: the B operand of every MFMA read from the LDS (one ds_read_b128 per MFMA, the very
 * inline-asm block statement of the sweep: 18 MFMAs per 64-item block, 2 MFMA waves per SIMD), a VALU read of the block's
 * accumulators, and four loader waves streaming `rows` into two LDS slots by LDS-DMA -- no hand-over, no lists, no candidates.
 * rows: >= (n_blk + 1024) x 19 456 bytes of bf16 data of the caller's kind (random bf16: ~1 300 TFLOP/s executed; constant: ~1 700);
 * out: 4 words of scratch.  1024 workgroups x 8 MFMA waves x n_blk blocks x 18 MFMAs. */
double pda_peak_mfma_lds_flops_per_launch(int n_blk);
int pda_peak_mfma_lds_bf16(const void* rows, size_t n_bytes, void* out, int n_blk, void* stream);
int pda_peak_copy(const float* src, float* dst, size_t n_floats, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PDA_HIP_EXPERIMENTAL_H */
