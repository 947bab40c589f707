/* pda_hip.h -- C ABI of libpda_hip.so: the MI355X (gfx950) implementation of the PDA BPR-MF hot path.
 *
 * This is the drop-in boundary.  The reference (zyang1580/PDA) has no plugin registry: the path is a
 * TensorFlow-1.14 graph driven through `sess.run` from Python, plus two native helpers with a plain
 * C-style signature (caller-owned contiguous row-major buffers, blocking, void):
 *     void arg_top_k_2d(float*, int rating_len, int rows_num, int top_k, int thread_num, int* results)
 *                                                         util/cython/include/arg_topk.h:29
 *     void cpp_evaluate_matrix(float*, int, vector<unordered_set<int>>&, vector<int>, int, int, float*)
 *                                                         evaluator/backend/cpp/include/evaluate.h:53-54
 * The entry points below keep that shape (plain pointers + sizes, caller-owned buffers) and replace
 * the `sess.run([...])` fetches named next to each one.  Differences, by design:
 *   - pointers are DEVICE pointers (HBM) unless the name says `host`;
 *   - every call takes a hipStream_t (passed as void*) and is stream-ordered / asynchronous;
 *   - every call returns int: 0 = ok, <0 = PDA_ERR_* ; no exceptions cross the ABI;
 *   - no global mutable state: re-entrant across streams and threads.
 * Python binds this with ctypes (pda_amd/_lib.py); INTEGRATION.md shows the stub a reference
 * maintainer would add to MF/.
 *
 * THIS header is the STABLE surface: pda_score_topk_plan and every entry point a plan may name (item preps, score + mask +
 * top-K of each generation), pda_topk_merge, pda_bpr_step_f32 / _bf16 (+ pda_sgd_apply_f32, pda_refresh_rows_bf16, the item-parallel shard step),
 * the reference's optimiser (pda_adam_step_f32, pda_adam_dense_sweep*_f32), pda_metrics, pda_sample_triplets(_dev).  Everything else the library
 * exports -- phases and seeds of the item-sharded evaluation, planned / looped train steps, the lazy replay of Adam, sampler look-ahead, measured
 * peaks -- is declared in pda_hip_experimental.h, without a stability promise.
 */
#ifndef PDA_HIP_H
#define PDA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDA_ABI_VERSION 2   /* 2 (round 6): pda_adam_step_f32 / pda_adam_dense_sweep4_f32 added; the entry points beyond the drop-in surface moved to
                          * pda_hip_experimental.h (same symbols, same signatures) */

#define PDA_OK 0
#define PDA_ERR_ARG (-1)         /* null pointer, negative size, K out of range ...            */
#define PDA_ERR_UNSUPPORTED (-2) /* embed dim / K / mode combination without a compiled kernel */
#define PDA_ERR_LAUNCH (-3)      /* hipLaunch / hipGetLastError failure                        */
#define PDA_ERR_WORKSPACE (-4)   /* caller-provided workspace too small                        */

/* Recommendation heads of DatasetApi_Model.Create_Recommendation (MF/train_new_api.py:594-612). */
#define PDA_HEAD_RAW 0 /* 'main_branch': top_k(R + M)                              :597-598      */
#define PDA_HEAD_POP 1 /* 'main_with_pop' / 'condition': top_k((elu(R)+1)*pop + M) :601-604,608-609 */

/* How hist_indptr is indexed. */
#define PDA_HIST_BY_BLOCK_ROW 0 /* row r of this call's user block (the reference's COO rows, :736) */
#define PDA_HIST_BY_USER_ID 1   /* global user id users[r] (one resident CSR for all users)         */

/* Update modes of the fused triplet step. */
#define PDA_UPD_NONE 0       /* loss + per-occurrence gradients only (parity harness)                        */
#define PDA_UPD_SGD_FUSED 1  /* north_star: in-kernel row update, atomics on item rows.  ASYNCHRONOUS inside a launch:
                              * a workgroup may gather a row that another workgroup of the same launch has already
                              * updated (hot positives, repeated users), so the result is a hogwild-style step -- equal
                              * to the mini-batch step up to O(lr) cross terms, not bit-reproducible.  The exact
                              * mini-batch step is PDA_UPD_NONE + pda_sgd_apply_f32 (two launches).              */
#define PDA_UPD_DENSE_GRAD 2 /* atomically sum gradients into dense gU/gI (feeds pda_adam_dense_sweep_f32)  */
#define PDA_UPD_ANY_ORDER 0x100 /* OR into update_mode when the batch is NOT grouped by positive: equal positives are then
                                 * combined on chip wherever they sit in a workgroup (slightly slower on grouped batches) */
#define PDA_UPD_USERS_DISTINCT 0x200 /* OR into update_mode of pda_bpr_step_f32(PDA_UPD_SGD_FUSED): the caller asserts that no user id
                              * occurs twice in the batch (the reference's sampler draws users without replacement,
                              * MF/train_new_api.py:380-381; pda_sample_triplets does while B <= the pool) -- the user row then
                              * takes a plain store of (row read) - lr * gradient instead of d fp32 atomics.  With a repeated
                              * user one of its updates is lost. */
#define PDA_UPD_SGD_ITEMS 3  /* internal: what pda_bpr_step_shard_f32 runs (item rows updated, user grads out)   */
#define PDA_UPD_DENSE_ITEMS 4 /* internal: pda_bpr_step_shard_f32 with gI_shard (item grads accumulated, user grads out) */

#define PDA_MAX_K 64
#define PDA_TOPK_CAP 60 /* per-user on-chip candidate slots (>= K) */

int pda_abi_version(void);
const char* pda_error_string(int code);

/* ------------------------------------------------------------------------------------------------
 * Full-catalogue score + history mask + top-K   (A6; replaces sess.run(main_brach_topk_idx |
 * main_with_pop_topk_idx | condition_topk_idx), MF/train_new_api.py:632-637, i.e. the graph
 * Gather -> MatMul[Bu,I] -> Elu,+1,*pop -> SparseTensorDenseAdd(-inf) -> TopKV2 of :594-612
 * and MF/model_api.py:62,113).
 *
 *   U          f32 [n_users_total, d]      user table (replicated on every rank)
 *   I_shard    f32 [n_items_local, d]      this rank's item rows; local row j is global item item_offset+j
 *   pop_shard  f32 [n_items_local] or NULL popularity^gamma for the same rows (required for PDA_HEAD_POP)
 *   users      i32 [n_users_blk]           user ids of the block
 *   hist_*     CSR of train items to mask: indptr i64 [rows+1], indices i32 (GLOBAL item ids) which MUST
 *              be sorted ascending within each row (duplicates allowed); NULL indptr = no mask
 *   K          1..PDA_MAX_K (reference: 50), K <= n_items_local * is not required * (short lists are
 *              padded with empty keys and resolved by the merge)
 *   n_splits   item-range splits per user tile (>=1; 0 = choose automatically).  Each split yields a
 *              partial list; out_keys holds [n_splits, n_users_blk, K] packed keys, best first.
 * Packed key: (orderable_bits(score) << 32) | (0xFFFFFFFF - global_item); 0 = empty slot; larger = better,
 * so that ties on the score resolve to the lower item index exactly like tf.nn.top_k.
 * Scores are exact fp32: a k-ordered fmaf chain in the order documented in oracle/pda_oracle.c.
 * ------------------------------------------------------------------------------------------------ */
int pda_score_topk_auto_splits(int n_users_blk, int n_items_local);          /* the n_splits the calls below pick for n_splits <= 0 */

int pda_score_topk_f32(const float* U, const float* I_shard, const float* pop_shard, const int32_t* users,
                       int n_users_blk, int item_offset, int n_items_local, int d, const int64_t* hist_indptr,
                       const int32_t* hist_indices, int hist_row_mode, int K, int head, int n_splits,
                       uint64_t* out_keys, void* stream);

/* Pre-filtered path: a bf16 MFMA pass scores every pair approximately with a rigorous error bound, only the pairs that
 * can still beat a user's running threshold are rescored with the exact fp32 chain; returns exactly the keys of
 * pda_score_topk_f32.  Two kernels sit behind these entry points and the library picks per call (PDA_SCORE_KERNEL=v2|v3
 * forces one): v3 (pda_score_topk_v3.hip: ONE bf16 MFMA per k-step plus one k-step that carries the threshold test,
 * candidate ring, exact lists) everywhere except the early-terminating sweep over bf16 tables at d = 256, which runs v2
 * (pda_score_topk_v2.hip: hi/lo split = three MFMAs per k-step, approximate lists, exact finish, exact-kernel
 * recomputation of user tiles whose near-tie band overflows).
 * The item shard is pre-split once per weight version into bf16 planes, padded row norms and the 16 bf16 per item of the
 * folded threshold test (the prep of an ordered sweep also holds the pieces of 1/pop: redo it when pop changes):
 *   pda_item_prep_bytes(n, d)  -> size of the caller-owned `prep` buffer (device)
 *   pda_item_prep_f32(I_shard, n, d, prep, stream)
 *   pda_score_topk_workspace_bytes(n_users_blk) -> size of the per-call scratch `workspace` (device; the call
 *                                 clears it itself, stream-ordered; one workspace per concurrent call).  After the
 *                                 call, the u64 at byte offset 8 holds the number of 32-item tiles scored, summed
 *                                 over workgroups (statistics: (n_items/32) * ceil(n_users/128) when nothing was skipped),
 *                                 and the u32 at byte offset 4 the number of pairs rescored exactly (v3 only)
 *   pda_score_topk_prepped_f32(..arguments of pda_score_topk_f32 plus `prep` after I_shard and `workspace` before stream..)
 * d in {64,128,256}; K <= PDA_TOPK_CAP-4.  PDA_HEAD_POP: pop_shard must be >= 0 (NaN entries never rank; it is pop^gamma of a
 * normalised count, MF/train_new_api.py:952-959) -- the filter is derived for that; the exact kernel pda_score_topk_f32
 * has no such requirement. */
size_t pda_item_prep_bytes(int n_items_local, int d);
int pda_item_prep_f32(const float* I_shard, int n_items_local, int d, void* prep, void* stream);
size_t pda_score_topk_workspace_bytes(int n_users_blk);
int pda_score_topk_prepped_f32(const float* U, const float* I_shard, const void* prep, const float* pop_shard,
                               const int32_t* users, int n_users_blk, int item_offset, int n_items_local, int d,
                               const int64_t* hist_indptr, const int32_t* hist_indices, int hist_row_mode, int K,
                               int head, int n_splits, uint64_t* out_keys, void* workspace, void* stream);

/* Ordered sweep with exact early termination (same keys again).  PDA's head is popularity-weighted:
 *     head(s) = (elu(s)+1) pop  <=  pop + ||u|| (pop ||i||)          (PDA_HEAD_RAW:  s <= ||u|| ||i||)
 * so when the catalogue is visited strongest-bound-first, a user block can stop as soon as the bound of everything
 * not yet visited is below every user's running K-th value -- most of a popularity-skewed catalogue is never scored.
 * The caller chooses the visiting `order` (i32 [n_items_local], a permutation of the local item ids; ANY permutation
 * gives the exact result, `argsort(-pop * (1 + c ||i||))` makes the stop early); pop must be >= 0 for the bound to be
 * tight (|pop| is used).  History rows have to be in visiting positions too:
 *   pda_item_prep_ordered_bytes(n, d)
 *   pda_item_prep_ordered_f32(I_shard, pop_shard|NULL, order, n, d, prep, stream)     per (weights, pop, order)
 *   pda_item_prep_ordered_check(prep, n, d, stream)   optional, synchronises: PDA_ERR_ARG if `order` was not a permutation
 *   pda_hist_reorder(prep, n, d, item_offset, hist_indptr, hist_indices, n_rows, out_indices, stream)
 *        out_indices i32 [nnz]: in-shard ids replaced by item_offset + position, every row sorted again
 *   pda_score_topk_ordered_f32(.. as pda_score_topk_prepped_f32, with hist_indices_ord after hist_indices ..)
 *        hist_indices stays the ORIGINAL (item-id) history, ascending per row: v3 walks hist_indices_ord over its exact
 *        warm-up tiles only and masks later candidates by a binary search in hist_indices; v2's exact-kernel
 *        recomputation of overflowed tiles uses it too.
 *        early_stop = 1: stop as described.  early_stop = 0: every tile is scored (a dense sweep) but still in visiting
 *        order -- strong items first raise the running thresholds quickly, which alone removes most of the candidate
 *        handling (C3: 4.2 ms instead of 8.7 ms per 65 536 users).
 * With n_splits > 1 the splits take interleaved tiles of the visiting order.
 * No reference counterpart: the reference scores the full [Bu, I] matrix (MF/train_new_api.py:594-612). */
size_t pda_item_prep_ordered_bytes(int n_items_local, int d);
int pda_item_prep_ordered_f32(const float* I_shard, const float* pop_shard, const int32_t* order, int n_items_local, int d,
                              void* prep, void* stream);
int pda_item_prep_ordered_check(const void* prep, int n_items_local, int d, void* stream);
int pda_hist_reorder(const void* prep, int n_items_local, int d, int item_offset, const int64_t* hist_indptr,
                     const int32_t* hist_indices, int n_rows, int32_t* out_indices, void* stream);
int pda_score_topk_ordered_f32(const float* U, const float* I_shard, const void* prep, const float* pop_shard,
                               const int32_t* users, int n_users_blk, int item_offset, int n_items_local, int d,
                               const int64_t* hist_indptr, const int32_t* hist_indices, const int32_t* hist_indices_ord,
                               int hist_row_mode, int K, int head, int early_stop, int n_splits, uint64_t* out_keys,
                               void* workspace, void* stream);

/* bf16 tables (BASELINE config 5: 10M x 2M, d=256 bf16).  U and I_shard hold bf16 bit patterns (uint16, row-major).
 * The score of a pair is DEFINED as the same fp32 fmaf chain applied to the widened values -- i.e. these entry points
 * return exactly the keys pda_score_topk_f32 returns on the tables converted to fp32 (tests/test_gpu_score_topk.py).
 * Products of two bf16 are exact in fp32, so the pre-filter's error bound is 2^-14 instead of 2^-8 ||u|| ||i|| and there are
 * no hi/lo planes: the prep buffer holds the row norms and the test pieces (plus, ordered, the rows in visiting order).
 * d in {64,128,256}.  pda_item_prep_ordered_check / pda_hist_reorder serve both table types. */
size_t pda_item_prep_bf16_bytes(int n_items_local, int d);
int pda_item_prep_bf16(const uint16_t* I_shard, int n_items_local, int d, void* prep, void* stream);
int pda_score_topk_bf16(const uint16_t* U, const uint16_t* I_shard, const void* prep, const float* pop_shard,
                        const int32_t* users, int n_users_blk, int item_offset, int n_items_local, int d,
                        const int64_t* hist_indptr, const int32_t* hist_indices, int hist_row_mode, int K, int head,
                        int n_splits, uint64_t* out_keys, void* workspace, void* stream);
size_t pda_item_prep_ordered_bf16_bytes(int n_items_local, int d);
int pda_item_prep_ordered_bf16(const uint16_t* I_shard, const float* pop_shard, const int32_t* order, int n_items_local,
                               int d, void* prep, void* stream);
int pda_score_topk_ordered_bf16(const uint16_t* U, const uint16_t* I_shard, const void* prep, const float* pop_shard,
                                const int32_t* users, int n_users_blk, int item_offset, int n_items_local, int d,
                                const int64_t* hist_indptr, const int32_t* hist_indices, const int32_t* hist_indices_ord,
                                int hist_row_mode, int K, int head, int early_stop, int n_splits, uint64_t* out_keys,
                                void* workspace, void* stream);

/* Generation 4 of the pre-filtered path (pda_score_topk_v4.hip): the same filter + exact rescoring (same keys again), with
 * 64 user rows per matrix-core wave, separate rescoring waves and LDS-DMA tile streaming.  ONE entry point per table type
 * covers the three sweep modes of the entry points above:
 *   order == NULL in the prep            natural item order
 *   order given, early_stop == 0         visiting order, every tile scored
 *   order given, early_stop == 1         visiting order with exact early termination
 * The prep holds the item rows padded to 2 d + 48 bytes in visiting order (bf16 row, the 16 bf16 of the folded threshold
 * test, pop, local id, norm) plus the suffix bounds; redo it when the weights, pop or the order change:
 *   pda_item_prep4_bytes(n, d)
 *   pda_item_prep4_f32 / _bf16(I_shard, pop_shard|NULL, order|NULL, n, d, prep, stream)
 *        pop_shard NULL: a prep for PDA_HEAD_RAW only; with pop_shard: PDA_HEAD_POP, and PDA_HEAD_RAW without early_stop
 *   pda_item_prep4_check(prep, n, d, stream)   optional, synchronises: PDA_ERR_ARG if `order` was not a permutation
 *   pda_score_topk4_auto_splits(n_users_blk, n_items_local, d)   the n_splits the call picks for n_splits <= 0
 *   pda_score_topk4_f32 / _bf16(.. as pda_score_topk_ordered_*, without hist_indices_ord: train items are masked at the
 *        candidate stage by a binary search in hist_indices ..)
 *   pda_score_topk4_workspace_bytes(n_users_blk, n_items_local, d, n_splits)   the workspace of these calls (n_splits as
 *        passed to the call, <= 0 = automatic): the counters, and for d <= 128 -- workgroups of 512 users, whose exact lists
 *        do not fit the LDS -- 57 list slots of 8 bytes per user and item split (120 MB at 262 144 users), and 128 bytes per
 *        user for the Bloom filters of the train items (the history mask at the candidate stage)
 *        and 16 bytes per user + 4 KiB for regrouping the users of an early-terminating sweep by predicted stopping tile
 *        (blocks of 98 304 users and more with one item split; internal: rows of out_keys stay the caller's block rows)
 * d in {64,128,256}; K <= 54; n_items_local <= 2^26.  out_keys doubles as the
 * hand-over buffer between the exact warm-up kernel and the sweep.  PDA_ERR_UNSUPPORTED: use the entry points above.
 * early_stop: bit 0 = exact early termination; the other bits are GEOMETRY HINTS (results never depend on them).
 *        bits 4 .. 6 = PDA_SWEEP_WARM_TILES(n), n = 1 .. 4 (0 = 4): 64-item tiles per split scored by the exact warm-up kernel.
 *        An item shard of an R-rank job wants 4 / R: the R warm-ups cover R x 64 n items between them, and the warm-up is the
 *        per-rank cost that does not shrink with the shard.
 *        bit 1 = PDA_SWEEP_FEW_CANDIDATES and bit 2 = PDA_SWEEP_WIDE: accepted, select nothing.  (Rounds 3 and 4: the default geometry with
 *        its lists in the workspace, and 512-user workgroups with 64 user rows per MFMA wave -- the huge geometry, bit 7, took over every
 *        block they served and round 5 removed them; profiles/README.md keeps their measurements.) */
#define PDA_SWEEP_FEW_CANDIDATES 2
#define PDA_SWEEP_WIDE 4
/*        bit 3 = PDA_SWEEP_MANY_CANDIDATES, the geometry hint for the opposite kind of sweep (d <= 128): hundreds of list
 *        insertions per user -- the raw head, the popularity head in natural item order.  128 users per workgroup: four MFMA
 *        waves, and EIGHT rescoring waves (one per 16 user rows) with the lists of 128 users and four tile slots in the LDS.
 *        Such a sweep is bound by its rescoring waves (cycle counters: 98 % busy at four per 256 users).  Identical keys. */
#define PDA_SWEEP_MANY_CANDIDATES 8
/*        bit 7 = PDA_SWEEP_HUGE, the geometry hint for the dense sweep of the popularity head in visiting order on VERY large user
 *        blocks (a prep built with the popularity the call passes): 1 024 users per workgroup, four waves of 512
 *        registers, 256 users each -- the users' bf16 rows in AGPRs, every item fragment read from the LDS feeds EIGHT MFMAs, the
 *        product transposed so that the threshold test is a per-lane compare (no test k-step), the item image pre-scaled by the
 *        popularity (pda_v5_sweep.h).  d = 256: 512 users per workgroup, 128 per wave.  Smaller
 *        blocks fill the chip with item splits (pda_amd.ops.huge_splits chooses them; one shared warm-up, bit 10).  Identical keys.
 *        Dense sweeps only (with bit 0 set: ignored); other heads, or a prep built without the popularity: the default geometry. */
#define PDA_SWEEP_HUGE 128
/*        bits 8 and 9 = PDA_SWEEP_HUGE_32X32, PDA_SWEEP_HUGE_2WG: accepted, select nothing.  (Round 4's A/B variants of the huge loop -- on
 *        v_mfma_f32_32x32x16_bf16, 9.16 against 8.07 ms, and as two 512-user workgroups per CU, 8.38 ms -- removed in round 5.) */
#define PDA_SWEEP_HUGE_32X32 256
#define PDA_SWEEP_HUGE_2WG 512
/*        bit 10 = PDA_SWEEP_WARM_PER_SPLIT.  By default a one-call sweep (warm-up + sweep in one entry point) over n_splits > 1 item
 *        splits runs ONE exact warm-up per user -- on the first warm tiles of the whole visiting order, handed to split 0 -- and every
 *        other split starts with an empty list and that warm-up's K-th value as its seed (a lower bound of the user's final K-th
 *        value: pairs below it stay out of the split's list, which may end shorter than K; the merged lists are the same).  With this
 *        bit every split warms up on its own first tiles as before round 4 (A/B measurements, cross-checks). */
#define PDA_SWEEP_WARM_PER_SPLIT 1024
#define PDA_SWEEP_WARM_TILES(n) (((n) & 7) << 4)
size_t pda_item_prep4_bytes(int n_items_local, int d);
int pda_item_prep4_f32(const float* I_shard, const float* pop_shard, const int32_t* order, int n_items_local, int d, void* prep,
                       void* stream);
int pda_item_prep4_bf16(const uint16_t* I_shard, const float* pop_shard, const int32_t* order, int n_items_local, int d, void* prep,
                        void* stream);
int pda_item_prep4_check(const void* prep, int n_items_local, int d, void* stream);
int pda_score_topk4_auto_splits(int n_users_blk, int n_items_local, int d);
size_t pda_score_topk4_workspace_bytes(int n_users_blk, int n_items_local, int d, int n_splits);
int pda_score_topk4_f32(const float* U, const float* I_shard, const void* prep, const float* pop_shard, const int32_t* users,
                        int n_users_blk, int item_offset, int n_items_local, int d, const int64_t* hist_indptr,
                        const int32_t* hist_indices, int hist_row_mode, int K, int head, int early_stop, int n_splits,
                        uint64_t* out_keys, void* workspace, void* stream);
int pda_score_topk4_bf16(const uint16_t* U, const uint16_t* I_shard, const void* prep, const float* pop_shard, const int32_t* users,
                         int n_users_blk, int item_offset, int n_items_local, int d, const int64_t* hist_indptr,
                         const int32_t* hist_indices, int hist_row_mode, int K, int head, int early_stop, int n_splits,
                         uint64_t* out_keys, void* workspace, void* stream);

/* WHICH entry point serves a call, with which item splits, geometry hint, visiting order and workspace: the library's own policy (round 5;
 * before, a caller had to re-implement ~80 lines of pda_amd/ops.py).  Results never depend on the plan -- every path returns the same packed
 * keys -- only the time does.  A caller:
 *     pda_score_plan p;  pda_score_topk_plan(n_users_blk, n_items_local, d, K, head, PDA_SWEEP_MODE_DEFAULT, 0, hist_row_mode, &p);
 *     build the item prep p.path wants with the visiting order p.order (pda_item_prep4_* for PDA_PATH_GEN4 / PDA_PATH_FUNNEL), allocate
 *     p.workspace_bytes and keys of p.keys_rows x K, call the entry point of p.path with p.n_splits and p.early_stop, merge the splits
 *     (pda_topk_merge) when p.n_splits > 1.
 * Reference precedent for "one native call does the job": util/cython/include/arg_topk.h:29-45. */
#define PDA_SWEEP_MODE_DEFAULT (-1)        /* the library's default for the head */
#define PDA_SWEEP_MODE_NATURAL 0           /* every tile, natural item order */
#define PDA_SWEEP_MODE_EARLY_STOP 1        /* visiting order + exact early termination */
#define PDA_SWEEP_MODE_VISITING_ORDER 2    /* every tile, in visiting order (thresholds rise early) */
#define PDA_PATH_EXACT_F32 1               /* pda_score_topk_f32 */
#define PDA_PATH_GEN3 2                    /* pda_score_topk_prepped_f32 / pda_score_topk_bf16 */
#define PDA_PATH_GEN3_ORDERED 3            /* pda_score_topk_ordered_f32 / _bf16 */
#define PDA_PATH_GEN4 4                    /* pda_score_topk4_f32 / _bf16 */
#define PDA_PATH_FUNNEL 7                  /* pda_score_topk7_f32 / _bf16 */
#define PDA_FUNNEL_WORKSPACE_BUDGET ((size_t)24 << 30)   /* a block whose funnel workspace (~27 KB per user) would pass this is planned on PDA_PATH_GEN4 */
#define PDA_ORDER_NATURAL 0
#define PDA_ORDER_BY_POPULARITY 1          /* most popular first (stable) */
#define PDA_ORDER_BY_NORM 2                /* largest ||i|| first (stable) */
#define PDA_ORDER_RANDOM 3                 /* any fixed random permutation */
typedef struct pda_score_plan {
    int path;                 /* PDA_PATH_* */
    int sweep_mode;           /* PDA_SWEEP_MODE_* the call will run (the default resolved) */
    int n_splits;             /* item splits = leading dimension of out_keys */
    int early_stop;           /* the early_stop argument of pda_score_topk4_* / pda_score_topk_ordered_*: sweep mode bit | geometry hint */
    int order;                /* PDA_ORDER_*: the visiting order the item prep is built with */
    int prep_with_pop;        /* the item prep is built WITH the popularity vector */
    size_t workspace_bytes;
    size_t keys_rows;         /* n_splits x n_users_blk rows of K keys */
} pda_score_plan;
/* hist_row_mode: PDA_HIST_BY_BLOCK_ROW / PDA_HIST_BY_USER_ID, or -1 for a call without a train-item mask */
int pda_score_topk_plan(int n_users_blk, int n_items_local, int d, int K, int head, int sweep_mode, int table_bf16, int hist_row_mode,
                        pda_score_plan* plan);
/* The funnel (round 5; pda_score_funnel.hip, pda_v7_funnel.h): score + mask + top-K for the RAW head -- the ranking the reference evaluates
 * in every epoch and the only one of --train normal (MF/train_new_api.py:597-598,1139-1141,1160-1165) -- on large user blocks.  Same packed keys
 * as every other generation, ONE list per user (out_keys [n_users_blk, K]; the item splits are merged inside).  A running exact list takes
 * K ln(n / n0) insertions per user in any order that does not sort by score, each an exact rescoring with a gathered item row; the funnel sweeps
 * growing parts of the catalogue against FIXED per-user thresholds on the huge geometry's machine mapping, writes the bf16 bounds of what beats
 * them to per-lane lists, tightens the thresholds between the parts from the r-th largest lower bound seen, and rescores only the final K + the
 * pairs inside the bound's band exactly.  Rows whose estimate was too bold (probability ~1e-6 per launch) or whose lists overflowed are served by
 * generation 4's exact lists inside the same call.
 *   prep       pda_item_prep7_f32 / _bf16 (below: pda_item_prep4_* without a popularity and with the half-tile image in FP16 -- the funnel's sweeps run
 *              v_mfma_f32_16x16x32_f16: eleven significant bits instead of bf16's eight, the band of the rigorous bound 8 x narrower on fp32 tables), with a
 *              visiting order that is a RANDOM permutation of the shard (the thresholds' ranks assume that the items seen so far are a uniform sample; any
 *              order gives exact results, a sorted one more fallbacks).  A prep of pda_item_prep4_* is refused: error word 7, every row served by the
 *              exact fallback.
 *   hist_*     optional; PDA_HIST_BY_USER_ID (the exact fallback re-blocks the failed rows: its cost follows their number) or PDA_HIST_BY_BLOCK_ROW (the
 *              reference's per-block COO mask: the rows keep their places and the fallback sweeps the WHOLE block again if a row failed -- pda_score_topk_plan
 *              sends such calls here up to 16 384 users)
 *   head       PDA_HEAD_RAW (PDA_ERR_UNSUPPORTED otherwise: the popularity head in visiting order is the huge geometry's)
 *   d          64 / 128 / 256 (d = 256: 512-user workgroups, as the huge geometry);  K <= 54;  4 096 <= n_items_local <= 2^26
 *   workspace  pda_score_topk7_workspace_bytes(n_users_blk, n_items_local, d) bytes; +0 error word, +4 pairs rescored exactly, +16 kernel identity
 * Reference counterpart: MF/model_api.py:62 (the raw ratings) + tf.nn.top_k(..., 50) behind the -inf mask, MF/train_new_api.py:594-612. */
int pda_item_prep7_f32(const float* I_shard, const int32_t* order, int n_items_local, int d, void* prep, void* stream);      /* pda_item_prep4_bytes(n, d) bytes */
int pda_item_prep7_bf16(const uint16_t* I_shard, const int32_t* order, int n_items_local, int d, void* prep, void* stream);
size_t pda_score_topk7_workspace_bytes(int n_users_blk, int n_items_local, int d);
int pda_score_topk7_f32(const float* U, const float* I_shard, const void* prep, const int32_t* users, int n_users_blk, int item_offset,
                        int n_items_local, int d, const int64_t* hist_indptr, const int32_t* hist_indices, int hist_row_mode, int K, int head,
                        uint64_t* out_keys, void* workspace, void* stream);
int pda_score_topk7_bf16(const uint16_t* U, const uint16_t* I_shard, const void* prep, const int32_t* users, int n_users_blk, int item_offset,
                         int n_items_local, int d, const int64_t* hist_indptr, const int32_t* hist_indices, int hist_row_mode, int K, int head,
                         uint64_t* out_keys, void* workspace, void* stream);

/* The ratings as a MATRIX: batch_ratings / condition_ratings (MF/model_api.py:62,113) as DatasetApi_Model.testing fetches them for the NeuRec
 * evaluators' predict() protocol (MF/train_new_api.py:642-696; the reference imports that protocol and never calls it).  out f32 [n_users_blk, n_items]:
 * out[r][j] = head(U[users[r]] . I[items[j]]) with pop[j] (f32 [n_items], required for PDA_HEAD_POP) -- items i32 [n_items] row ids of I, or NULL for
 * rows 0 .. n_items - 1.  No mask, no top-K.  The same k-ordered fmaf chain as every top-K entry point: a value here equals the value a top-K call
 * returns for the pair, bit for bit.  d in {32, 64, 128, 256}.  Not a hot path -- the top-K calls exist so that this matrix is never formed. */
int pda_score_dense_f32(const float* U, const float* I, const float* pop, const int32_t* users, int n_users_blk, const int32_t* items, int n_items, int d,
                        int head, float* out, void* stream);

/* Merge R partial lists per user (R item splits of one GPU, or R ranks after the RCCL all-gather).
 *   in_keys  u64 [R, n_users_blk, K]  each list best-first, empty slots = 0
 *   out_keys u64 [n_users_blk, K] or NULL;  out_idx i32 / out_val f32 [n_users_blk, K] or NULL.
 * When out_idx is given, slots still empty after the merge (a user with fewer than K unmasked items)
 * are filled like tf.nn.top_k would: score -inf, lowest masked item ids first, taken from the
 * user's (sorted) history row when hist_indptr != NULL (else idx = -1).
 * No reference counterpart (single device there); see SURVEY 8(e). */
int pda_topk_merge(const uint64_t* in_keys, int R, int n_users_blk, int K, uint64_t* out_keys, int32_t* out_idx,
                   float* out_val, const int32_t* users, const int64_t* hist_indptr, const int32_t* hist_indices,
                   int hist_row_mode, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused BPR triplet step (A1-A5; replaces sess.run([opt*, loss*, mf_loss*, reg_loss*]),
 * MF/train_new_api.py:1080-1090 over the graph of MF/model_api.py:51-53,102-121 / :449-451,695-706).
 *
 *   U, I         f32 tables, updated in place for PDA_UPD_SGD_FUSED
 *   users,pos,neg i32 [B];  pos_pop,neg_pop f32 [B] or both NULL (NULL = plain BPRMF, no ELU/pop)
 *   regs         --regs;  reg_div = the --batch_size flag constant (MF/model_api.py:118)
 *   lr           SGD learning rate (PDA_UPD_SGD_FUSED only)
 *   g_user/g_pos/g_neg  f32 [B,d] per-occurrence gradient out (PDA_UPD_NONE), may be NULL otherwise
 *   gU, gI       dense f32 gradient accumulators (PDA_UPD_DENSE_GRAD), same shape as the tables
 *   loss_acc     f32 [3]: (loss, mf_loss, reg_loss) are ADDED to it (zero it to fetch one step)
 * ------------------------------------------------------------------------------------------------ */
int pda_bpr_step_f32(float* U, float* I, const int32_t* users, const int32_t* pos, const int32_t* neg,
                     const float* pos_pop, const float* neg_pop, int B, int d, float regs, float reg_div, float lr,
                     int update_mode, float* g_user, float* g_pos, float* g_neg, float* gU, float* gI,
                     float* loss_acc, void* stream);

/* Apply phase of the EXACT mini-batch SGD step: after pda_bpr_step_f32(PDA_UPD_NONE, g_user, g_pos, g_neg) -- forward pass
 * and per-occurrence gradients of the WHOLE batch against the unchanged tables -- this scatters
 *   U[users[t]] -= lr g_user[t],  I[pos[t]] -= lr g_pos[t],  I[neg[t]] -= lr g_neg[t]      (fp32 atomics; rows may repeat).
 * Together: one step of plain SGD on the batch loss (the reference's graph with tf.train.GradientDescentOptimizer in
 * place of Adam, MF/model_api.py:83), deterministic up to the order of the fp32 additions. */
int pda_sgd_apply_f32(float* U, float* I, const int32_t* users, const int32_t* pos, const int32_t* neg, const float* g_user,
                      const float* g_pos, const float* g_neg, int B, int d, float lr, void* stream);

/* bf16 tables (config 5): the forward pass gathers bf16 rows (U_bf16 / I_bf16, uint16 bit patterns) and computes in fp32
 * exactly as pda_bpr_step_f32 does on the widened rows; gradients are fp32.  bf16 rows cannot take atomic adds, so
 *   PDA_UPD_NONE        per-occurrence gradients out (masters may be NULL)
 *   PDA_UPD_SGD_FUSED   the fp32 MASTER tables (U_master / I_master, same shape) take the SGD update; afterwards
 *                       pda_refresh_rows_bf16 re-rounds (RNE) the touched rows of the bf16 tables from the masters
 *   PDA_UPD_DENSE_GRAD  gradients summed into gU / gI (fp32) for pda_adam_dense_sweep_f32 on the masters
 * pda_refresh_rows_bf16(master, shadow, rows, n_rows, d): shadow[rows[i]] = bf16(master[rows[i]]); rows == NULL means
 * rows 0 .. n_rows-1 (a dense cast of a whole table).  d a multiple of 8. */
int pda_bpr_step_bf16(const uint16_t* U_bf16, const uint16_t* I_bf16, float* U_master, float* I_master,
                      const int32_t* users, const int32_t* pos, const int32_t* neg, const float* pos_pop,
                      const float* neg_pop, int B, int d, float regs, float reg_div, float lr, int update_mode,
                      float* g_user, float* g_pos, float* g_neg, float* gU, float* gI, float* loss_acc, void* stream);
int pda_refresh_rows_bf16(const float* master, uint16_t* shadow, const int32_t* rows, int n_rows, int d, void* stream);

/* Item-parallel training (north_star "each rank owns an item-embedding slice, BPR negatives sampled
 * locally"): rank r holds the item rows [item_offset, item_offset + n_local) and a replica of U.  Its sub-batch has
 * positives AND negatives inside that slice (global ids).  pda_bpr_step_shard_f32 applies the SGD update to the local
 * item rows (like PDA_UPD_SGD_FUSED), leaves U untouched and writes the per-triplet user gradient to g_user [B_local, d]
 * (row stride g_stride floats, >= d, multiple of 4);
 * after ONE all-gather of (users, g_user) every rank calls pda_apply_user_grads_f32 on all R*B_local rows, which keeps
 * the replicas of U identical.  mean_div = the GLOBAL batch size (the 1/B of the BPR mean, MF/model_api.py:114),
 * reg_div as in pda_bpr_step_f32; loss_acc receives this rank's share of (loss, mf, reg): their sum over ranks is the
 * loss of the global batch.  R shard steps + the apply equal one pda_bpr_step_f32(PDA_UPD_SGD_FUSED) on the concatenated
 * batch (tests/test_gpu_bpr_step.py).  g rows may be strided (g_stride floats, >= d) so that the packed exchange buffer
 * can be applied in place.  gI_shard != NULL (f32, shape of I_shard) switches to the reference's optimiser: the item
 * gradients are summed into gI_shard instead of applied, the gathered user gradients are summed into a dense gU with
 * pda_apply_user_grads_f32(gU, .., lr = -1), and pda_adam_dense_sweep_f32 runs on U (identically on every rank) and on the
 * rank's I_shard.  No reference counterpart (single device there). */
int pda_bpr_step_shard_f32(const float* U, float* I_shard, int item_offset, const int32_t* users, const int32_t* pos,
                           const int32_t* neg, const float* pos_pop, const float* neg_pop, int B_local, int d, float regs,
                           float reg_div, float mean_div, float lr, float* g_user, int g_stride, float* gI_shard,
                           float* loss_acc, void* stream);
int pda_apply_user_grads_f32(float* U, const int32_t* users, const float* g, int n, int d, int g_stride, float lr,
                             void* stream);

/* TF-1.14 AdamOptimizer `_apply_sparse_shared`: decay m,v on EVERY row, add the (pre-summed) sparse
 * gradient, update EVERY row (MF/model_api.py:83,:470-471 [TF-ext]).  `g` is the dense accumulator
 * filled by PDA_UPD_DENSE_GRAD; it is reset to zero by this sweep.  n = rows*d.  lr_t is the
 * bias-corrected rate lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller. */
int pda_adam_dense_sweep_f32(float* var, float* m, float* v, float* g, size_t n, float lr_t, float beta1,
                             float beta2, float eps, void* stream);
/* The same sweep over TWO tables (U and I of the model) in one launch; element for element the arithmetic of two
 * pda_adam_dense_sweep_f32 calls. */
int pda_adam_dense_sweep2_f32(float* var_a, float* m_a, float* v_a, float* g_a, size_t n_a, float* var_b, float* m_b,
                              float* v_b, float* g_b, size_t n_b, float lr_t, float beta1, float beta2, float eps,
                              void* stream);

/* Round 6 -- one reference train step in TWO launches (MF/model_api.py:83,470-471: minimize() = the batch's gradients + TF-1.14 dense-decay Adam on
 * every row of both tables; MF/train_new_api.py:1078-1090 runs it once per batch).
 *   pda_adam_step_f32: the step kernel of pda_bpr_step_f32(PDA_UPD_DENSE_GRAD) sums the batch's gradients into gU / gI AND writes `step_tag` into
 *     tagU[user] / tagI[pos], tagI[neg] (i32 [rows], zero before the first step; step_tag >= 1 and different from the previous call's -- the step
 *     number); then pda_adam_dense_sweep4_f32's kernel sweeps both tables, reads gU / gI only where tag == step_tag and zeroes them there.
 *     No bitmaps, no mark launch, no memsets (pda_adam_mark_rows + pda_adam_dense_sweep3_f32: five launches per step).
 *     flags: PDA_UPD_ANY_ORDER | PDA_UPD_USERS_DISTINCT or 0.  loss_acc f32 [3] accumulates (loss, mf, reg) or NULL.
 *     PRECONDITION (both functions): gU / gI are ZERO on every row whose tag differs from step_tag (true when they are only ever written by
 *     this function: the sweep zeroes what the step wrote).  A caller that accumulates gradients of other rows must tag them, too.
 *   cache_policy: PDA_ADAM_CACHE_AUTO = plain loads / stores while x, m, v of both tables (3 (rows_a + rows_b) d 4 bytes) fit
 *     PDA_ADAM_RESIDENT_BYTES -- they then stay in the 256 MiB Infinity Cache from step to step (C1 / C2: 54 MB) --, non-temporal streams above
 *     (C3: 1.8 GB); _RESIDENT / _STREAM force one.  Bit-identical tables either way, and to pda_adam_dense_sweep2_f32 / 3. */
#define PDA_ADAM_CACHE_AUTO 0
#define PDA_ADAM_CACHE_RESIDENT 1
#define PDA_ADAM_CACHE_STREAM 2
#define PDA_ADAM_RESIDENT_BYTES (160u << 20)
int pda_adam_dense_sweep4_f32(float* var_a, float* m_a, float* v_a, float* g_a, size_t rows_a, const int32_t* tag_a, float* var_b, float* m_b, float* v_b,
                              float* g_b, size_t rows_b, const int32_t* tag_b, int d, int tag, float lr_t, float beta1, float beta2, float eps,
                              int cache_policy, void* stream);
int pda_adam_step_f32(float* U, float* mU, float* vU, float* gU, int32_t* tagU, size_t n_users, float* I, float* mI, float* vI, float* gI, int32_t* tagI,
                      size_t n_items, const int32_t* users, const int32_t* pos, const int32_t* neg, const float* pos_pop, const float* neg_pop, int B,
                      int d, float regs, float reg_div, int step_tag, float lr_t, float beta1, float beta2, float eps, int flags, int cache_policy,
                      float* loss_acc, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Ranking metrics (A8; replaces get_performance + the Pool(5) reduction, MF/used_metric.py:4-80,
 * MF/train_new_api.py:741-778).   topk i32 [n_rows, k_cols]; targets as CSR by block row;
 * Ks i32 [n_ks] (device);  sums f64 [4, n_ks] = precision, recall, ndcg, hit -- ADDED to (not divided).
 * ------------------------------------------------------------------------------------------------ */
int pda_metrics(const int32_t* topk, int n_rows, int k_cols, const int64_t* tgt_indptr, const int32_t* tgt_indices,
                const int32_t* Ks, int n_ks, double* sums, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Device-side triplet sampler (N1; replaces generator_n_batch / generator_n_batch_with_pop,
 * MF/train_new_api.py:260-288, 366-412).  One batch: users[r] given (unique, as rd.sample),
 * pos uniform over the user's train items (0 when empty, :387-389), neg uniform over the catalogue
 * minus the user's train items by rejection (:397-401), pops = pop_matrix[item, slot_of_pos] (:402-403).
 *   users      i32 [B]: read when gen_users == 0; WRITTEN when gen_users != 0 (B distinct users drawn
 *              from user_pool[0..n_pool) -- or from 0..n_pool-1 when user_pool is NULL -- by a keyed
 *              Feistel permutation; with replacement when B > n_pool, :383)
 *   train CSR indexed by user id, indices sorted ascending; train_slots i32 parallel to indices or NULL
 *   negatives are drawn from [neg_lo, neg_hi) (the whole catalogue, or a rank's item shard)
 *   pop_matrix f32 [n_items, n_slots] or NULL.   Counter-based RNG: (seed, step, row) -> draws.
 * ------------------------------------------------------------------------------------------------ */
int pda_sample_triplets(int32_t* users, int gen_users, const int32_t* user_pool, int n_pool, int B,
                        const int64_t* train_indptr, const int32_t* train_indices, const int32_t* train_slots,
                        int neg_lo, int neg_hi, const float* pop_matrix, int n_slots, uint64_t seed, uint64_t step,
                        int32_t* pos, int32_t* neg, float* pos_pop, float* neg_pop, void* stream);

/* The same sampler with the step taken from DEVICE memory (*step_dev), so that sampler + step can be captured in a HIP
 * graph and replayed with a new batch every time.  The stream of batches advances either with
 * pda_counter_add(step_dev, 1, stream) inside the graph, or -- one launch less -- through step_next: the kernel stores
 * *step_dev + 1 there (it must be a DIFFERENT location; alternate two slots from launch to launch, an even number of
 * launches per graph).  pda_sample_triplets_dev(.., seed, step_dev, ..) draws exactly what
 * pda_sample_triplets(.., seed, *step_dev, ..) draws. */
int pda_sample_triplets_dev(int32_t* users, int gen_users, const int32_t* user_pool, int n_pool, int B,
                            const int64_t* train_indptr, const int32_t* train_indices, const int32_t* train_slots,
                            int neg_lo, int neg_hi, const float* pop_matrix, int n_slots, uint64_t seed,
                            const uint64_t* step_dev, uint64_t* step_next, int32_t* pos, int32_t* neg, float* pos_pop,
                            float* neg_pop, void* stream);
int pda_counter_add(uint64_t* counter, uint64_t inc, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PDA_HIP_H */
