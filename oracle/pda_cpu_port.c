/* CPU baseline, native: the evaluation hot path (MF/model_api.py:62,113; MF/train_new_api.py:594-612,780-794) restated for host
 * cores the way one would write it in C -- fused, never materialising the [2048, n_items] rating block the reference re-reads
 * eight times: per user the dot products stream over the catalogue (compiler-vectorised, AVX2 + FMA), the head is bounded by
 * (max(s, 0) + 1) pop before the exponential is paid for, train items are looked up only for the few pairs that would enter the
 * list, and a K-entry heap keeps the best.  Users over all OpenMP threads.
 *
 * TEST / BENCH INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg, kind = "port"): not linked into the product, not a parity
 * checker (built with -ffast-math: the summation order of a dot product is the vectoriser's; ties resolve to the lower item id
 * like tf.nn.top_k).  oracle/pda_oracle.c stays the bit-exact restatement. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float v; int32_t i; } ent_t;

/* min-heap on (v asc, i desc): the root is the WORST kept entry (lowest value; among equals the HIGHEST item id) */
static inline int worse(ent_t a, ent_t b) { return a.v < b.v || (a.v == b.v && a.i > b.i); }
static inline void sift_down(ent_t* h, int n, int p) {
    for (;;) {
        int l = 2 * p + 1, r = l + 1, m = p;
        if (l < n && worse(h[l], h[m])) m = l;
        if (r < n && worse(h[r], h[m])) m = r;
        if (m == p) return;
        ent_t t = h[p]; h[p] = h[m]; h[m] = t;
        p = m;
    }
}
static inline void sift_up(ent_t* h, int p) {
    while (p > 0) {
        int q = (p - 1) / 2;
        if (!worse(h[p], h[q])) return;
        ent_t t = h[p]; h[p] = h[q]; h[q] = t;
        p = q;
    }
}
static int cmp_best_first(const void* a, const void* b) {
    const ent_t *x = (const ent_t*)a, *y = (const ent_t*)b;
    if (x->v != y->v) return x->v > y->v ? -1 : 1;
    return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
}

#include <immintrin.h>

/* dots of ONE user row with FOUR consecutive item rows: 4 accumulators over d / 8 chunks, then a 4-way horizontal sum */
static inline __m128 dot4(const float* restrict u, const float* restrict v, int d) {
    __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
    for (int k = 0; k < d; k += 8) {
        const __m256 x = _mm256_loadu_ps(u + k);
        a0 = _mm256_fmadd_ps(x, _mm256_loadu_ps(v + k), a0);
        a1 = _mm256_fmadd_ps(x, _mm256_loadu_ps(v + d + k), a1);
        a2 = _mm256_fmadd_ps(x, _mm256_loadu_ps(v + 2 * d + k), a2);
        a3 = _mm256_fmadd_ps(x, _mm256_loadu_ps(v + 3 * d + k), a3);
    }
    const __m256 h01 = _mm256_hadd_ps(a0, a1), h23 = _mm256_hadd_ps(a2, a3);
    const __m256 h = _mm256_hadd_ps(h01, h23);                         /* (s0, s1, s2, s3) in each 128-bit half */
    return _mm_add_ps(_mm256_castps256_ps128(h), _mm256_extractf128_ps(h, 1));
}

typedef struct {
    ent_t heap[64];
    int n;
    float tau;
} list_t;

static inline void offer(list_t* L, int K, float s, int j, int mode, const float* pop, const int32_t* hist, int64_t hb, int64_t he) {
    float ub = s;
    if (mode) ub = ((s > 0.f ? s : 0.f) + 1.0f) * pop[j];
    if (L->n == K && !(ub > L->tau)) return;              /* (equal: the earlier, lower id stays) */
    float hv = s;
    if (mode) hv = (s > 0.f ? s + 1.0f : expf(s)) * pop[j];
    if ((L->n == K && !(hv > L->tau)) || hv != hv) return;
    for (int64_t p = hb; p < he; ++p)
        if (hist[p] == j) return;
    if (L->n < K) {
        L->heap[L->n].v = hv; L->heap[L->n].i = j;
        sift_up(L->heap, L->n);
        ++L->n;
    } else {
        L->heap[0].v = hv; L->heap[0].i = j;
        sift_down(L->heap, L->n, 0);
    }
    if (L->n == K) L->tau = L->heap[0].v;
}

/* mode 0: raw head (rec_type 'main_branch'); 1: (elu(s) + 1) pop ('condition' / 'main_with_pop').  hist CSR rows are block rows.
 * out_idx int32 [n_users_blk, K], out_val float [n_users_blk, K], best first; rows with fewer than K unmasked items end with -1.
 * Cache blocking: a thread takes 32 user rows (16 KiB at d = 128) and walks the catalogue in groups of 4 item rows (2 KiB): every
 * item row comes from memory once per 32 users.  d % 8 == 0, K <= 64. */
int cpu_port_score_topk(const float* U, const float* I, const float* pop, const int32_t* users, int n_users_blk, int n_items, int d,
                        const int64_t* hist_indptr, const int32_t* hist_indices, int K, int mode, int32_t* out_idx, float* out_val) {
    if (K < 1 || K > 64 || n_items < 1 || d < 8 || (d & 7) || (mode && !pop)) return -1;
    enum { UB = 32 };      /* (8 users per item pass: 10 k users/s on 128 threads at 200 000 items -- every pass streams the 102 MB table from memory) */
    const int n_groups = (n_users_blk + UB - 1) / UB;
#pragma omp parallel for schedule(dynamic, 1)
    for (int g = 0; g < n_groups; ++g) {
        const int r0 = g * UB, nr = n_users_blk - r0 < UB ? n_users_blk - r0 : UB;
        list_t L[UB];
        const float* urow[UB];
        int64_t hb[UB], he[UB];
        for (int q = 0; q < nr; ++q) {
            L[q].n = 0;
            L[q].tau = -INFINITY;
            urow[q] = U + (size_t)users[r0 + q] * d;
            hb[q] = hist_indptr ? hist_indptr[r0 + q] : 0;
            he[q] = hist_indptr ? hist_indptr[r0 + q + 1] : 0;
        }
        int j = 0;
        for (; j + 4 <= n_items; j += 4) {
            const float* v = I + (size_t)j * d;
            for (int q = 0; q < nr; ++q) {
                float s4[4];
                _mm_storeu_ps(s4, dot4(urow[q], v, d));
                /* one vector compare against the row's threshold would save the scalar tests; they are rare branches either way */
                for (int t = 0; t < 4; ++t) offer(&L[q], K, s4[t], j + t, mode, pop, hist_indices, hb[q], he[q]);
            }
        }
        for (; j < n_items; ++j)
            for (int q = 0; q < nr; ++q) {
                const float* v = I + (size_t)j * d;
                float s = 0.f;
                for (int k = 0; k < d; ++k) s += urow[q][k] * v[k];
                offer(&L[q], K, s, j, mode, pop, hist_indices, hb[q], he[q]);
            }
        for (int q = 0; q < nr; ++q) {
            qsort(L[q].heap, (size_t)L[q].n, sizeof(ent_t), cmp_best_first);
            for (int k = 0; k < K; ++k) {
                out_idx[(size_t)(r0 + q) * K + k] = k < L[q].n ? L[q].heap[k].i : -1;
                out_val[(size_t)(r0 + q) * K + k] = k < L[q].n ? L[q].heap[k].v : -INFINITY;
            }
        }
    }
    return 0;
}
