/* CPU oracle (plain C) for the PDA full-catalogue score + mask + top-K path.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg through ctypes.  The product (pda_amd/) never links or calls it.
 *
 * It restates, for sizes numpy is too slow for:
 *   R = U[users] @ I^T                              MF/model_api.py:62 (:459 for BPRMF)
 *   (elu(R)+1) * pop[None,:]                        MF/model_api.py:113, MF/train_new_api.py:601-602
 *   R + SparseTensor(-inf at train items)           MF/train_new_api.py:597,603,608
 *   tf.nn.top_k(., K): desc, ties -> lower index    MF/train_new_api.py:598,604,609  [TF-ext]
 *
 * Parity status: the TF graph cannot be run (TensorFlow 1.14 absent) => this restatement is
 * "parity unpinned" against TF itself; it is cross-checked against oracle/pda_oracle.py
 * (float64 numpy) and, on tie-free rows, against the reference's own C++ arg_top_k_2d
 * (util/cython/include/arg_topk.h:15-45) built into oracle/_ref by oracle/Makefile.
 *
 * Two accumulation orders are offered for the fp32 dot:
 *   order 0: float64 accumulate, rounded once           (the "truth" used with a tolerance)
 *   order 1: the HIP kernel's exact fp32 arithmetic       (bit-exact comparison of scores):
 *            two fmaf chains, chain c&1 over the k-chunks c = 0..d/8-1:
 *              for s in 0..3: acc=fmaf(u[8c+s],i[8c+s],acc); acc=fmaf(u[8c+4+s],i[8c+4+s],acc)
 *            (what v_mfma_f32_32x32x2_f32 computes when lane-half h supplies k = 8c+4h+s),
 *            then score = chain0 + chain1 (see pda_amd/csrc/pda_score_topk.hip).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float dot_chain(const float* u, const float* v, int d) {
    float acc[2] = {0.0f, 0.0f};                 /* even / odd k-chunks: two independent chains */
    for (int c = 0; c < d / 8; ++c)
        for (int s = 0; s < 4; ++s) {
            acc[c & 1] = fmaf(u[8 * c + s], v[8 * c + s], acc[c & 1]);
            acc[c & 1] = fmaf(u[8 * c + 4 + s], v[8 * c + 4 + s], acc[c & 1]);
        }
    return acc[0] + acc[1];
}

static inline float dot_f64(const float* u, const float* v, int d) {
    double acc = 0.0;
    for (int k = 0; k < d; ++k) acc += (double)u[k] * (double)v[k];
    return (float)acc;
}

/* mode: 0 = main_branch (raw dot), 1 = main_with_pop / condition ((elu+1)*pop) */
static inline float head(float s, int mode, const float* pop, int item) {
    if (mode == 0) return s;
    float t = s > 0.0f ? s + 1.0f : expf(s);
    return t * pop[item];
}

/* better(a,b): a ranks before b  <=>  (val desc, idx asc) */
static inline int better(float va, int ia, float vb, int ib) {
    return va > vb || (va == vb && ia < ib);
}

/* One row: keep a K-sized binary min-heap on (val, idx) with `better` as the order. */
static void topk_row(const float* sc, int n, int K, int* out_idx, float* out_val) {
    float* hv = out_val;
    int* hi = out_idx;
    int sz = 0;
    for (int j = 0; j < n; ++j) {
        float v = sc[j];
        if (sz < K) {
            int p = sz++;
            hv[p] = v; hi[p] = j;
            while (p > 0) {                      /* sift up: root = worst */
                int q = (p - 1) / 2;
                if (better(hv[q], hi[q], hv[p], hi[p])) {
                    float tv = hv[q]; hv[q] = hv[p]; hv[p] = tv;
                    int ti = hi[q]; hi[q] = hi[p]; hi[p] = ti;
                    p = q;
                } else break;
            }
        } else if (better(v, j, hv[0], hi[0])) {
            hv[0] = v; hi[0] = j;
            int p = 0;
            for (;;) {
                int l = 2 * p + 1, r = l + 1, w = p;
                if (l < sz && better(hv[w], hi[w], hv[l], hi[l])) w = l;
                if (r < sz && better(hv[w], hi[w], hv[r], hi[r])) w = r;
                if (w == p) break;
                float tv = hv[w]; hv[w] = hv[p]; hv[p] = tv;
                int ti = hi[w]; hi[w] = hi[p]; hi[p] = ti;
                p = w;
            }
        }
    }
    /* heap -> sorted best-first (simple insertion sort, K <= 64) */
    for (int a = 1; a < sz; ++a) {
        float v = hv[a]; int i = hi[a]; int b = a - 1;
        while (b >= 0 && better(v, i, hv[b], hi[b])) { hv[b + 1] = hv[b]; hi[b + 1] = hi[b]; --b; }
        hv[b + 1] = v; hi[b + 1] = i;
    }
}

/* Full path for a block of users.  hist CSR rows are block-relative (row r = users[r]).
 * item_offset/n_items_local let the caller score an item shard [off, off+n_local) of the table
 * (indices in hist and in out_idx are GLOBAL item ids).  Returns 0, or -1 on bad args. */
int oracle_score_topk(const float* U, const float* I, const float* pop, const int32_t* users, int n_users_blk,
                      int n_items_local, int item_offset, int d, const int64_t* hist_indptr,
                      const int32_t* hist_indices, int K, int mode, int order, int32_t* out_idx, float* out_val,
                      float* scores_out /* optional [n_users_blk, n_items_local] or NULL */) {
    if (K > n_items_local || K < 1 || d % 8 != 0) return -1;
#pragma omp parallel
    {
        float* sc = (float*)malloc(sizeof(float) * (size_t)n_items_local);
#pragma omp for schedule(dynamic, 8)
        for (int r = 0; r < n_users_blk; ++r) {
            const float* u = U + (size_t)users[r] * d;
            for (int j = 0; j < n_items_local; ++j) {
                const float* v = I + (size_t)(item_offset + j) * d;
                float s = order == 1 ? dot_chain(u, v, d) : dot_f64(u, v, d);
                sc[j] = head(s, mode, pop, item_offset + j);
            }
            if (hist_indptr)
                for (int64_t p = hist_indptr[r]; p < hist_indptr[r + 1]; ++p) {
                    int it = hist_indices[p] - item_offset;
                    if (it >= 0 && it < n_items_local) sc[it] = -INFINITY;
                }
            if (scores_out) memcpy(scores_out + (size_t)r * n_items_local, sc, sizeof(float) * (size_t)n_items_local);
            topk_row(sc, n_items_local, K, out_idx + (size_t)r * K, out_val + (size_t)r * K);
            for (int k = 0; k < K; ++k) out_idx[(size_t)r * K + k] += item_offset;
        }
        free(sc);
    }
    return 0;
}

/* CPU-native top-K baseline ("CPU-native-topk"): per-row selection of the best K of a precomputed
 * [rows, n_items] rating matrix (a K-sized heap per row: the algorithm class of the reference's per-row partial_sort of an
 * index vector, util/cython/include/arg_topk.h:15-45), rows over OpenMP threads (the reference: a thread pool, :29-45).
 * out_idx int32 [rows, K], best first, ties to the lower index. */
void oracle_arg_topk_2d(const float* ratings, int n_items, int rows, int K, int32_t* out_idx) {
#pragma omp parallel
    {
        float* hv = (float*)malloc(sizeof(float) * (size_t)K);
#pragma omp for schedule(dynamic, 16)
        for (int r = 0; r < rows; ++r) topk_row(ratings + (size_t)r * n_items, n_items, K, out_idx + (size_t)r * K, hv);
        free(hv);
    }
}

/* Scores only, chain order, for bit-exactness tests of the MFMA accumulation. */
void oracle_scores_chain(const float* U, const float* I, const int32_t* users, int n_users_blk, int n_items, int d,
                         float* out) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n_users_blk; ++r)
        for (int j = 0; j < n_items; ++j)
            out[(size_t)r * n_items + j] = dot_chain(U + (size_t)users[r] * d, I + (size_t)j * d, d);
}

/* Ranking metrics from a top-K matrix and a target CSR: MF/used_metric.py:4-80 and the
 * reduction of MF/train_new_api.py:741-778.  sums[4][nK] = precision, recall, ndcg, hit (NOT yet /tot_user). */
void oracle_metrics(const int32_t* topk, int n_rows, int Kcols, const int64_t* tgt_indptr, const int32_t* tgt_indices,
                    const int32_t* Ks, int nK, double* sums) {
    for (int i = 0; i < 4 * nK; ++i) sums[i] = 0.0;
    for (int r = 0; r < n_rows; ++r) {
        int64_t b = tgt_indptr[r], e = tgt_indptr[r + 1];
        int npos = (int)(e - b);
        for (int q = 0; q < nK; ++q) {
            int K = Ks[q];
            double hits = 0, dcg = 0, idcg = 0;
            for (int k = 0; k < K && k < Kcols; ++k) {
                int it = topk[(size_t)r * Kcols + k], h = 0;
                for (int64_t p = b; p < e; ++p) if (tgt_indices[p] == it) { h = 1; break; }
                hits += h;
                dcg += h / log2((double)k + 2.0);
            }
            for (int k = 0; k < K && k < npos; ++k) idcg += 1.0 / log2((double)k + 2.0);
            sums[0 * nK + q] += hits / K;
            sums[1 * nK + q] += hits / npos;
            sums[2 * nK + q] += idcg == 0.0 ? 0.0 : dcg / idcg;
            sums[3 * nK + q] += hits > 1.0 ? 1.0 : hits;
        }
    }
}
