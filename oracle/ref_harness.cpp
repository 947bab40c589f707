// Harness that compiles the REFERENCE's own native top-K / metric headers where they lie
// under /root/reference (never copied) and exposes them with a C ABI so tests can check the
// oracle restatement against them.  TEST INFRASTRUCTURE ONLY; output goes to oracle/_ref/.
//
//   arg_top_k_2d          util/cython/include/arg_topk.h:29-45
//   cpp_evaluate_matrix   evaluator/backend/cpp/include/evaluate.h:53-72 (+ metric.h:17-109)
//
// Caveat: std::partial_sort_copy leaves the order of equal scores unspecified, so
// these are oracles for tie-free rows / sets only; TF's rule is "lower index first".
#include <unordered_set>
#include <vector>
#include "arg_topk.h"
#include "evaluate.h"

extern "C" {

void ref_arg_top_k_2d(float* ratings, int rating_len, int rows_num, int top_k, int thread_num, int* results) {
    arg_top_k_2d(ratings, rating_len, rows_num, top_k, thread_num, results);
}

// test_items given as CSR; metric ids as in metric.h:111-116 (1 precision, 2 recall, 3 ap, 4 ndcg, 5 mrr).
void ref_cpp_evaluate_matrix(float* rating_matrix, int rating_len, int n_users, const long long* indptr,
                             const int* indices, const int* metric, int n_metric, int top_k, int thread_num,
                             float* results) {
    std::vector<std::unordered_set<int>> test_items(n_users);
    for (int u = 0; u < n_users; ++u)
        for (long long p = indptr[u]; p < indptr[u + 1]; ++p) test_items[u].insert(indices[p]);
    std::vector<int> m(metric, metric + n_metric);
    cpp_evaluate_matrix(rating_matrix, rating_len, test_items, m, top_k, thread_num, results);
}
}
