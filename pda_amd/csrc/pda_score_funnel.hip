// The funnel (round 5): score + mask + top-K for sweeps that meet hundreds of list insertions per user (raw head, natural order) on the huge
// geometry's machine mapping -- pda_v7_funnel.h.  A translation unit of its own (pda_score_topk_v4.hip takes minutes to build).
#include "pda_v4_shared.h"
#include <cstdlib>

namespace {
#include "pda_v7_funnel.h"
}  // namespace

#include <cmath>
#include <vector>

namespace {

// ---- tunables (pda_debug_funnel_tune: measurements only) ------------------------------------------------------------------------
double g_fail_p = 1e-6;        // a launch's threshold lies above the row's final K-th value with at most this probability (then: the exact fallback)
int g_growth = 4;              // every launch sees this many times the items seen before it (2 from a sixth of the catalogue on)
int g_cap_e = 64;              // entries per (user, quarter) list and launch
int g_first_tiles = 4;         // without a maxima launch: the first launch, 256 items against -inf
int g_maxima = 1;              // the first launch keeps maxima only (rank-8 thresholds for free: no lists, no selection)
int g_first_mult = 2;          // the first emitting launch behind a maxima launch over m tiles: tiles [0, g_first_mult x m)
int g_late_den = 6;            // from 1 / g_late_den of the catalogue on the parts grow by g_late_growth_x10 / 10 instead of g_growth
int g_late_growth_x10 = 20;
bool g_tuned = false;          // pda_debug_funnel_tune(2) was called: the globals above rule whatever the item splits
constexpr int kFallbackSplits = 8;
// (a tune call that restores every default hands the schedule back to the library: the parts that grow by 8 / 4 under item splits)
static void funnel_tuned_or_default() {
    g_tuned = !(g_fail_p == 1e-6 && g_growth == 4 && g_cap_e == 64 && g_first_tiles == 4 && g_first_mult == 2 && g_late_den == 6 && g_late_growth_x10 == 20);
}

struct Stage7 {
    int lo, hi;                // global 64-item tiles [lo, hi)
    int rank_next;             // the threshold of the next launch: this rank among the lower bounds seen (0: none follows)
    int maxima;                // 1: the first launch in maxima mode (sweep7_kernel<.., MAXM> + maxthr7_kernel: rank 8, nothing written per half-tile)
};

// P(Gamma(r, 1) <= x) = 1 - exp(-x) sum_{i < r} x^i / i!
double gamma_cdf7(int r, double x) {
    double term = 1.0, sum = 1.0;
    for (int i = 1; i < r; ++i) {
        term *= x / i;
        sum += term;
    }
    return 1.0 - std::exp(-x) * sum;
}
// The items are visited in a random order, so the first m are a uniform sample of the n: the number of items ABOVE the r-th largest of the sample
// is ~ Gamma(r) n / m.  The threshold is safe when at least K items of the catalogue reach it: the smallest r with P(Gamma(r) < K m / n) <= p.
int rank_for7(int K, double m, double n, double p) {
    if (m >= n) return K;
    const double x = (double)K * m / n;
    for (int r = 2; r < K; ++r)
        if (gamma_cdf7(r, x) <= p) return r;
    return K;
}
std::vector<Stage7> schedule7(int n_tiles, int n_items, int K, int n_splits = 1) {
    std::vector<Stage7> st;
    // What a launch writes per list is ~ rank x (growth - 1) / (4 quarters x S item splits): a block that fills the chip with item splits (few users:
    // the reference's 2 048-user blocks run 32 splits, configs 1 / 2 five) has room for parts that grow by 8, then by 4 -- one or two launches and
    // selections less where every launch is fixed cost (round 6, tools/funnel_small_tune.sh: 2 048 users x 200 000 items 0.91 -> 0.75 ms, config 1
    // 1.64 -> 1.48, config 2 1.56 -> 1.49).  One split (262 144 users) keeps growth 4 / 2: there the longer parts overflow the lists.
    const bool wide = n_splits >= 4 && !g_tuned;
    const int growth = wide ? 8 : g_growth, first_mult = wide ? 4 : g_first_mult, late_growth_x10 = wide ? 40 : g_late_growth_x10;
    int lo = 0, hi = std::min(n_tiles, std::max(1, g_first_tiles));
    if (g_maxima) {
        // the maxima launch: as many tiles as rank 8 carries (P(Gamma(8) < K m / n) <= p), at least two; the first emitting launch starts over at tile 0
        int m2 = 2;
        while (m2 * 2 <= n_tiles / 8 && gamma_cdf7(8, (double)K * (m2 * 2) * 64.0 / n_items) <= g_fail_p) m2 *= 2;
        if (m2 * 3 / 2 <= n_tiles / 8 && gamma_cdf7(8, (double)K * (m2 * 3 / 2) * 64.0 / n_items) <= g_fail_p) m2 = m2 * 3 / 2;
        if (gamma_cdf7(8, (double)K * m2 * 64.0 / n_items) <= g_fail_p && K >= 8) {
            st.push_back(Stage7{0, m2, 8, 1});
            hi = std::min(n_tiles, m2 * std::max(1, first_mult));            // (rank 8 is a coarse estimate: the first emitting launch stays short)
        }
    }
    for (;;) {
        // (a last part of less than half a step joins the one before it)
        if (n_tiles - hi < (hi - lo) / 2) hi = n_tiles;
        Stage7 s7{lo, hi, 0, 0};
        if (hi < n_tiles) s7.rank_next = rank_for7(K, std::min((double)hi * 64.0, (double)n_items), (double)n_items, g_fail_p);
        st.push_back(s7);
        if (hi >= n_tiles) break;
        lo = hi;
        // the parts grow by g_growth, and by 2 from a sixth of the catalogue on: what a launch writes per list is ~ rank x (growth - 1) + the
        // pairs inside the bound's band, and the lists of the last, longest launches are the ones that fill
        const int gr10 = (long long)hi * g_late_den >= n_tiles ? std::max(11, late_growth_x10) : 10 * std::max(2, growth);
        hi = (int)std::min<long long>((long long)n_tiles, ((long long)hi * gr10 + 9) / 10);
    }
    return st;
}

// item splits of the emitting launches: the huge geometry's rule (ops.huge_splits; rounds of 256 workgroups x tiles per split)
int funnel_splits7(int n_users, int n_items_local, int d) {
    const int ut = d == 256 ? 512 : 1024;
    const int utiles = (n_users + ut - 1) / ut;
    const int tiles = (n_items_local + 63) / 64;
    const int smax = std::max(1, std::min(64, tiles / 32));
    int best = 1;
    double best_cost = -1.0;
    for (int s = 1; s <= smax; ++s) {
        const double cost = (double)((utiles * s + 255) / 256) * (0.02 + 1.0 / s) + 0.004 * s * ((double)n_users / 262144.0);
        if (best_cost < 0.0 || cost < best_cost - 1e-9) {
            best = s;
            best_cost = cost;
        }
    }
    return best;
}

struct Ws7 {
    size_t ufrag, unorm, uerr, eu, ecnt, mrun, elist, thr, tk, tmax, ncand, flags, cand, qpool, qcnt, bloom, fail_list, fail_count, users2, seed2, fb_keys, fb_ws, total;
    int n_splits, cap_e, cap_q;
};
Ws7 ws7_layout(int n, int n_items_local, int d) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    Ws7 w{};
    w.n_splits = funnel_splits7(n, n_items_local, d);
    // (a power of two: threshold7_kernel then reads a row's 4 S lists interleaved -- 12 sparse register groups at S = 5 cost it 2.6 ns per row, 0.9 at S = 1)
    while (w.n_splits & (w.n_splits - 1)) w.n_splits &= w.n_splits - 1;
    w.cap_e = g_cap_e;
    // slots of a (row, quarter, split) list of the pool: what a launch adds per quarter shrinks with the splits; the first launch (everything above
    // -inf: 256 items over 4 quarters and S splits) must fit
    w.cap_q = w.n_splits == 1 ? 64 : w.n_splits == 2 ? 40 : w.n_splits <= 4 ? 28 : 24;        // (one split: 48 slots overflowed on 33 of 262 144 rows at config 3)
    const int ut = d == 256 ? 512 : 1024, nu = ut / 64;
    const size_t utiles = ((size_t)n + ut - 1) / ut, n_pad = utiles * ut, wgs = utiles * (size_t)w.n_splits;
    size_t b = 256;
    w.ufrag = b;
    b = al(b + n_pad * 2 * (size_t)d);
    w.unorm = b;
    b = al(b + n_pad * 4);
    w.uerr = b;
    b = al(b + n_pad * 4);
    w.eu = b;
    b = al(b + utiles * 4 * 8);
    w.ecnt = b;
    b = al(b + wgs * 4 * nu * 64 * 4);
    w.mrun = b;
    b = al(b + wgs * 4 * (size_t)(2 * nu + 1) * 64 * 4);
    w.elist = b;
    b = al(b + wgs * 4 * (size_t)w.cap_e * (64 * nu * 48));
    w.thr = b;
    b = al(b + (size_t)n * 4);
    w.tk = b;
    b = al(b + (size_t)n * 4);
    w.tmax = b;
    b = al(b + (size_t)n * 4);
    w.ncand = b;
    b = al(b + (size_t)n * 4);
    w.flags = b;
    b = al(b + (size_t)n * 4);
    w.cand = b;
    b = al(b + (size_t)n * kCand7 * 16);
    w.qpool = b;
    b = al(b + (size_t)n * 4 * w.n_splits * w.cap_q * 16);
    w.qcnt = b;
    b = al(b + (size_t)n * 4 * w.n_splits * 4);
    w.bloom = b;
    b = al(b + (size_t)n * 128);
    w.fail_list = b;
    b = al(b + (size_t)n * 4);
    w.fail_count = b;
    b = al(b + 256);
    w.users2 = b;
    b = al(b + (size_t)n * 4);
    w.seed2 = b;
    b = al(b + (size_t)n * 4);
    w.fb_keys = b;
    b = al(b + (size_t)kFallbackSplits * n * PDA_MAX_K * 8);
    w.fb_ws = b;
    b = al(b + pda_score_topk4_workspace_bytes(n, n_items_local, d, kFallbackSplits));
    w.total = b;
    return w;
}

__global__ void __launch_bounds__(256) init7_kernel(Rows7 r, int n, int* fail_count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *fail_count = 0;
    if (i >= n) return;
    r.thr[i] = -INFINITY;
    r.tk[i] = -INFINITY;
    r.tmax[i] = -INFINITY;
    r.ncand[i] = 0;
    r.flags[i] = 0u;
}

template <int D, bool BF>
int run_funnel_t(const void* U, const void* I_shard, const void* prep, const int32_t* users, int n, int item_offset, int n_items_local, const int64_t* hist_indptr,
                 const int32_t* hist_indices, int hist_row_mode, int K, uint64_t* out_keys, void* workspace, hipStream_t s) {
    constexpr int UPW = D == 256 ? 128 : 256, UT = 4 * UPW;
    const Ws7 W = ws7_layout(n, n_items_local, D);
    const Prep4Layout L = prep4_layout(n_items_local, D);
    const unsigned char* pb = reinterpret_cast<const unsigned char*>(prep);
    unsigned char* wsb = reinterpret_cast<unsigned char*>(workspace);
    if (hipMemsetAsync(workspace, 0, 256, s) != hipSuccess) return PDA_ERR_LAUNCH;
    Rows7 R{reinterpret_cast<float*>(wsb + W.thr), reinterpret_cast<float*>(wsb + W.tk), reinterpret_cast<float*>(wsb + W.tmax), reinterpret_cast<int*>(wsb + W.ncand),
            reinterpret_cast<uint64_t*>(wsb + W.qpool), reinterpret_cast<unsigned*>(wsb + W.qcnt), W.cap_q, reinterpret_cast<uint64_t*>(wsb + W.cand),
            reinterpret_cast<unsigned*>(wsb + W.flags)};
    int* fail_count = reinterpret_cast<int*>(wsb + W.fail_count);
    hipLaunchKernelGGL(init7_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, R, n, fail_count);
    PDA_CHECK_LAUNCH();
    const int n_pad = (n + UT - 1) / UT * UT;
    hipLaunchKernelGGL((uprep5_kernel<D, BF, true, UPW, true>), dim3((unsigned)(((size_t)n_pad * (D / 8) + 255) / 256)), dim3(256), 0, s, U, users, n, n_pad, wsb + W.ufrag,
                       reinterpret_cast<float*>(wsb + W.unorm), reinterpret_cast<float*>(wsb + W.uerr));
    PDA_CHECK_LAUNCH();
    uint32_t* bloom = nullptr;
    if (hist_indptr != nullptr) {
        bloom = reinterpret_cast<uint32_t*>(wsb + W.bloom);
        hipLaunchKernelGGL(hist_bloom7_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, users, hist_indptr, hist_indices, hist_row_mode, n,
                           reinterpret_cast<const int*>(pb + L.pos_of), item_offset, n_items_local, bloom);
        PDA_CHECK_LAUNCH();
    }
    Args7 e{pb + L.rows5, reinterpret_cast<const float*>(pb + L.meta5), wsb + W.ufrag, reinterpret_cast<const float*>(wsb + W.unorm), reinterpret_cast<const float*>(wsb + W.uerr), R.thr,
            wsb + W.elist,
            reinterpret_cast<unsigned*>(wsb + W.ecnt), reinterpret_cast<float*>(wsb + W.eu), reinterpret_cast<unsigned*>(workspace), n, W.n_splits, L.n_tiles, 0, 0, W.cap_e,
            reinterpret_cast<float*>(wsb + W.mrun), nullptr, reinterpret_cast<const int*>(pb + L.hdr)};
    Sel7 q{e, R, reinterpret_cast<const u32x4*>(pb + L.pinfo), users, hist_indptr, hist_indices, bloom, hist_row_mode, item_offset, n_items_local, K, 0, 0, U, I_shard, out_keys,
           reinterpret_cast<int*>(wsb + W.fail_list), fail_count};
    const std::vector<Stage7> stages = schedule7(L.n_tiles, n_items_local, K, W.n_splits);
    for (const Stage7& st : stages) {
        e.tile_lo = st.lo;
        e.tile_hi = st.hi;
        if (st.maxima) {
            const int rc = launch_sweep7<D, BF, UPW, true>(e, s);
            if (rc != PDA_OK) return rc;
            q.e = e;
            hipLaunchKernelGGL((maxthr7_kernel<D>), dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, q);
            PDA_CHECK_LAUNCH();
            continue;
        }
        const int rc = launch_sweep7<D, BF, UPW>(e, s);
        if (rc != PDA_OK) return rc;
        q.e = e;
        q.rank_next = st.rank_next;
        q.first_launch = (st.lo == 0 && !stages[0].maxima) ? 1 : 0;
        hipLaunchKernelGGL((expand7_kernel<D>), dim3((unsigned)(((size_t)(n + UT - 1) / UT) * W.n_splits * 64)), dim3(256), 0, s, q);
        PDA_CHECK_LAUNCH();
        {
            const int nsl = (4 * W.n_splits * W.cap_q + 63) / 64;          // 64-slot groups of a row's (quarter, split) lists
            const dim3 gr((unsigned)((n + 3) / 4));
            if (nsl <= 4) hipLaunchKernelGGL((threshold7_kernel<D, 4>), gr, dim3(256), 0, s, q);
            else if (nsl <= 7) hipLaunchKernelGGL((threshold7_kernel<D, 7>), gr, dim3(256), 0, s, q);
            else if (nsl <= 12) hipLaunchKernelGGL((threshold7_kernel<D, 12>), gr, dim3(256), 0, s, q);
            else hipLaunchKernelGGL((threshold7_kernel<D, 0>), gr, dim3(256), 0, s, q);
            PDA_CHECK_LAUNCH();
        }
    }
    hipLaunchKernelGGL((resolve7_kernel<D, BF>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, q);
    PDA_CHECK_LAUNCH();
    hipLaunchKernelGGL(stat7_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, s, R, n, (const int*)nullptr, reinterpret_cast<unsigned*>(workspace));
    PDA_CHECK_LAUNCH();
    // ---- the exact fallback: generation 4's many-candidates geometry on the failed rows (a device-side count: nothing runs when nobody failed)
    int32_t* users2 = reinterpret_cast<int32_t*>(wsb + W.users2);
    float* seed2 = reinterpret_cast<float*>(wsb + W.seed2);
    const int by_row = (hist_indptr != nullptr && hist_row_mode != PDA_HIST_BY_USER_ID) ? 1 : 0;
    int* n_dev2 = fail_count + 1;
    hipLaunchKernelGGL(fail_users7_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, users, q.fail_list, fail_count, n, R.tk, R.flags, by_row, users2, seed2, n_dev2);
    PDA_CHECK_LAUNCH();
    uint64_t* fb_keys = reinterpret_cast<uint64_t*>(wsb + W.fb_keys);
    const int rc = pda_v4_run_score4_dev(U, I_shard, BF, prep, nullptr, users2, n, n_dev2, item_offset, n_items_local, D, hist_indptr, hist_indices, hist_row_mode, K,
                                         PDA_HEAD_RAW, PDA_SWEEP_MANY_CANDIDATES, kFallbackSplits, seed2, fb_keys, wsb + W.fb_ws, s);
    if (rc != PDA_OK) return rc;
    hipLaunchKernelGGL(fail_merge7_kernel, dim3((unsigned)std::min((n + 3) / 4, 1024)), dim3(256), 0, s, fb_keys, kFallbackSplits, n, K, q.fail_list, fail_count, by_row, out_keys, reinterpret_cast<unsigned*>(workspace));
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

int run_funnel(const void* U, const void* I_shard, bool bf16, const void* prep, const int32_t* users, int n_users_blk, int item_offset, int n_items_local, int d,
               const int64_t* hist_indptr, const int32_t* hist_indices, int hist_row_mode, int K, int head, uint64_t* out_keys, void* workspace, hipStream_t s) {
    if (!U || !I_shard || !prep || !users || !out_keys || !workspace) return PDA_ERR_ARG;
    if (n_users_blk <= 0 || n_items_local <= 0 || item_offset < 0) return PDA_ERR_ARG;
    if (K < 1 || K > PDA_MAX_K) return PDA_ERR_ARG;
    if (hist_indptr && !hist_indices) return PDA_ERR_ARG;
    if (head != PDA_HEAD_RAW) return PDA_ERR_UNSUPPORTED;                          // (the popularity head in visiting order: generation 4 / the huge geometry)
    if (hist_indptr && hist_row_mode != PDA_HIST_BY_USER_ID && hist_row_mode != PDA_HIST_BY_BLOCK_ROW) return PDA_ERR_ARG;
    if (K > 54) return PDA_ERR_UNSUPPORTED;                                        // (the fallback is generation 4)
    if ((uint64_t)n_items_local > (1ull << 26) || n_items_local < 64 * 64) return PDA_ERR_UNSUPPORTED;
    switch (d) {
#define PDA_F7(DD) case DD: return bf16 ? run_funnel_t<DD, true>(U, I_shard, prep, users, n_users_blk, item_offset, n_items_local, hist_indptr, hist_indices, hist_row_mode, K, out_keys, workspace, s) \
                                         : run_funnel_t<DD, false>(U, I_shard, prep, users, n_users_blk, item_offset, n_items_local, hist_indptr, hist_indices, hist_row_mode, K, out_keys, workspace, s);
        PDA_F7(64) PDA_F7(128) PDA_F7(256)
#undef PDA_F7
        default: return PDA_ERR_UNSUPPORTED;
    }
}

}  // namespace

extern "C" size_t pda_score_topk7_workspace_bytes(int n_users_blk, int n_items_local, int d) {
    if (n_users_blk <= 0 || n_items_local <= 0 || (d != 64 && d != 128 && d != 256)) return 0;
    return ws7_layout(n_users_blk, n_items_local, d).total;
}
extern "C" int pda_score_topk7_f32(const float* U, const float* I_shard, const void* prep, const int32_t* users, int n_users_blk, int item_offset, int n_items_local, int d,
                                   const int64_t* hist_indptr, const int32_t* hist_indices, int hist_row_mode, int K, int head, uint64_t* out_keys, void* workspace,
                                   void* stream) {
    return run_funnel(U, I_shard, false, prep, users, n_users_blk, item_offset, n_items_local, d, hist_indptr, hist_indices, hist_row_mode, K, head, out_keys, workspace,
                      reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pda_score_topk7_bf16(const uint16_t* U, const uint16_t* I_shard, const void* prep, const int32_t* users, int n_users_blk, int item_offset, int n_items_local,
                                    int d, const int64_t* hist_indptr, const int32_t* hist_indices, int hist_row_mode, int K, int head, uint64_t* out_keys,
                                    void* workspace, void* stream) {
    return run_funnel(U, I_shard, true, prep, users, n_users_blk, item_offset, n_items_local, d, hist_indptr, hist_indices, hist_row_mode, K, head, out_keys, workspace,
                      reinterpret_cast<hipStream_t>(stream));
}
// measurements only: failure probability target, growth factor, list capacity, first tiles (0 / negative: keep)
extern "C" int pda_debug_funnel_tune(double fail_p, int growth, int cap_e, int first_tiles) {
    g_tuned = true;
    if (fail_p > 0.0) g_fail_p = fail_p;
    if (growth >= 2) g_growth = growth;
    if (cap_e > 0) g_cap_e = cap_e;
    if (first_tiles > 0) g_first_tiles = first_tiles;
    funnel_tuned_or_default();
    return PDA_OK;
}
extern "C" int pda_debug_funnel_tune2(int first_mult, int late_den, int late_growth_x10) {
    g_tuned = true;
    if (first_mult > 0) g_first_mult = first_mult;
    if (late_den > 0) g_late_den = late_den;
    if (late_growth_x10 > 10) g_late_growth_x10 = late_growth_x10;
    funnel_tuned_or_default();
    return PDA_OK;
}
// measurements / tests only: the first launch in maxima mode (1, the default) or as an emitting launch against -inf (0)
extern "C" int pda_debug_funnel_maxima(int on) {
    g_maxima = on ? 1 : 0;
    return PDA_OK;
}
// the workspace of a funnel: offs[0..9] = thr, tk, tmax, ncand, flags, cand, fail_list, fail_count, ecnt, elist; returns (n_splits << 16) | cap_e
extern "C" int pda_debug_funnel_layout(int n_users_blk, int n_items_local, int d, size_t* offs) {
    const Ws7 w = ws7_layout(n_users_blk, n_items_local, d);
    const size_t o[10] = {w.thr, w.tk, w.tmax, w.ncand, w.flags, w.cand, w.fail_list, w.fail_count, w.ecnt, w.elist};
    for (int i = 0; i < 10; ++i) offs[i] = o[i];
    return (w.n_splits << 16) | w.cap_e;
}
// the launches of a funnel: out[3 i .. 3 i + 2] = (first tile, end tile, rank of the next threshold); returns their number
extern "C" int pda_debug_funnel_schedule_for(int n_users_blk, int n_items_local, int d, int K, int* out, int max_stages);
extern "C" int pda_debug_funnel_schedule(int n_items_local, int K, int* out, int max_stages) { return pda_debug_funnel_schedule_for(0, n_items_local, 0, K, out, max_stages); }
// ... of a block of n_users_blk users at embed dim d (the item splits decide how fast the parts grow); n_users_blk = 0: one item split
extern "C" int pda_debug_funnel_schedule_for(int n_users_blk, int n_items_local, int d, int K, int* out, int max_stages) {
    const int S = n_users_blk > 0 ? ws7_layout(n_users_blk, n_items_local, d).n_splits : 1;
    const std::vector<Stage7> st = schedule7((n_items_local + 63) / 64, n_items_local, K, S);
    for (size_t i = 0; i < st.size() && (int)i < max_stages; ++i) {
        out[3 * i] = st[i].lo;
        out[3 * i + 1] = st[i].hi;
        out[3 * i + 2] = st[i].maxima ? -8 : st[i].rank_next;
    }
    return (int)st.size();
}


// ---- which kernel serves a call: the policy behind the C ABI (round 5; rounds 2 - 4 kept it in pda_amd/ops.py, where a C caller could not reach it).
// Results never depend on it (every path returns the same packed keys); the choices are by measurement.
extern "C" int pda_score_topk_huge_splits(int n_users_blk, int n_items_local, int d) {
    // item splits with which the huge geometry runs a block (one workgroup = 1 024 users x one split; d = 256: 512), or 0 when it should not: rounds of
    // 256 workgroups x (a workgroup's fixed cost + its share of the catalogue) + a little per split; the smallest S at the minimum
    if (n_users_blk < 4096 || n_items_local <= 0) return 0;
    const int ut = d == 256 ? 512 : 1024;
    const int utiles = (n_users_blk + ut - 1) / ut;
    const int s = funnel_splits7(n_users_blk, n_items_local, d);
    return utiles * s >= 128 ? s : 0;
}
extern "C" int pda_score_topk_plan(int n_users_blk, int n_items_local, int d, int K, int head, int sweep_mode, int table_bf16, int hist_row_mode,
                                   pda_score_plan* plan) {
    if (!plan || n_users_blk <= 0 || n_items_local <= 0 || K < 1 || K > PDA_MAX_K) return PDA_ERR_ARG;
    if (head != PDA_HEAD_RAW && head != PDA_HEAD_POP) return PDA_ERR_ARG;
    if (sweep_mode < PDA_SWEEP_MODE_DEFAULT || sweep_mode > PDA_SWEEP_MODE_VISITING_ORDER) return PDA_ERR_ARG;
    pda_score_plan p{};
    const bool dv = d == 64 || d == 128 || d == 256;
    // the sweep mode the library would choose itself: early-terminating for the popularity head; the raw head: dense in visiting order (by norm)
    // at d <= 128, natural order at d = 256
    if (sweep_mode == PDA_SWEEP_MODE_DEFAULT)
        sweep_mode = head == PDA_HEAD_POP ? PDA_SWEEP_MODE_EARLY_STOP : ((d == 64 || d == 128) ? PDA_SWEEP_MODE_VISITING_ORDER : PDA_SWEEP_MODE_NATURAL);
    p.sweep_mode = sweep_mode;
    const bool early = sweep_mode == PDA_SWEEP_MODE_EARLY_STOP, ordered = sweep_mode != PDA_SWEEP_MODE_NATURAL;
    // the raw head on catalogues of 4 096 items and more: the funnel, whatever the number of users (round 6: a block of 64 .. 1 000 users fills ONE 1 024-user tile partly and still
    // runs 0.25 - 0.38 ms against generation 4's 2.0 - 3.2 ms at 200 000 items, 0.20 - 0.33 against 0.48 - 1.0 at 20 000: CROSS_FEW=1 tools/funnel_crossover.py) -- unless its workspace (~27 KB per user: lists 12 KB, pools 7 KB, the
    // fallback's keys 4 KB ...; 7 GB at 262 144 users, stepping up where more item splits are taken) would pass PDA_FUNNEL_WORKSPACE_BUDGET: such
    // a block stays with generation 4 (same keys); callers with larger blocks cut them (pda_amd.ops.score_topk_keys: <= 262 144 users per call)
    const bool funnel = head == PDA_HEAD_RAW && dv && K <= 54 && n_items_local >= 4096 && (uint64_t)n_items_local <= (1ull << 26) &&
                        n_users_blk >= 1 && (n_items_local >= 20000 || n_users_blk <= 16384) && !early &&
                        (hist_row_mode < 0 || hist_row_mode == PDA_HIST_BY_USER_ID || (hist_row_mode == PDA_HIST_BY_BLOCK_ROW && n_users_blk <= 16384)) &&
                        pda_score_topk7_workspace_bytes(n_users_blk, n_items_local, d) <= PDA_FUNNEL_WORKSPACE_BUDGET;
    if (!dv || K > PDA_TOPK_CAP - 4) {
        if (table_bf16) return PDA_ERR_UNSUPPORTED;
        p.path = PDA_PATH_EXACT_F32;                             // pda_score_topk_f32: the exact fp32-MFMA kernel
        p.n_splits = pda_score_topk_auto_splits(n_users_blk, n_items_local);
        p.order = PDA_ORDER_NATURAL;
    } else if (funnel) {
        // (tools/funnel_crossover.py: from ONE 1 024-user tile on the funnel beats generation 4's many-candidates geometry on catalogues of 65 536 items and
        // more -- 2 048 users x 200 000 items 0.88 vs 1.85 ms --, and on smaller ones too (config 2, 50 000 users x 20 000 items: 1.73 vs 2.13 ms; config 1: 1.58 vs
        // 2.30) except the very smallest with many users (16 384 items x 65 536 users: 2.1 vs 1.9).  Round 6 (the selection kernels no longer latency-bound on few
        // users; CROSS_SMALL=1 tools/funnel_crossover.py): from 4 096 items on -- the funnel's own minimum -- up to 16 384 users: 8 192 items x 4 096 users 0.32 - 0.34 vs
        // 0.43 - 0.55 ms, 4 096 x 16 384 0.56 - 0.58 vs 0.69 - 0.90; 65 536 users on catalogues below 20 000 items stay with generation 4 (1.4 - 1.7 vs 0.95 - 1.5))
        p.path = PDA_PATH_FUNNEL;                                // pda_score_topk7_*
        p.n_splits = 1;                                          // (ONE list per user comes back: the item splits are merged inside)
        p.order = PDA_ORDER_RANDOM;
        p.workspace_bytes = pda_score_topk7_workspace_bytes(n_users_blk, n_items_local, d);
    } else if (K <= 54 && (uint64_t)n_items_local <= (1ull << 26) && !(d == 256 && (head == PDA_HEAD_RAW || !ordered))) {
        p.path = PDA_PATH_GEN4;                                  // pda_score_topk4_*
        p.order = !ordered ? PDA_ORDER_NATURAL : (head == PDA_HEAD_POP ? PDA_ORDER_BY_POPULARITY : PDA_ORDER_BY_NORM);
        p.prep_with_pop = head == PDA_HEAD_POP ? 1 : 0;
        p.n_splits = pda_score_topk4_auto_splits(n_users_blk, n_items_local, d);
        int hint = 0;
        const int hs = (head == PDA_HEAD_POP && sweep_mode == PDA_SWEEP_MODE_VISITING_ORDER) ? pda_score_topk_huge_splits(n_users_blk, n_items_local, d) : 0;
        if (hs > 0) {
            p.n_splits = hs;
            hint = PDA_SWEEP_HUGE;                               // the huge geometry, the catalogue split so that the block fills the chip
        } else if (head == PDA_HEAD_POP && sweep_mode == PDA_SWEEP_MODE_VISITING_ORDER && n_users_blk >= 196609 && d <= 128) {
            hint = PDA_SWEEP_HUGE;
        } else if (d <= 128 && !early && (head == PDA_HEAD_RAW || !ordered)) {
            hint = PDA_SWEEP_MANY_CANDIDATES;                    // hundreds of list insertions per user
        }
        p.early_stop = (early ? 1 : 0) | hint;
        p.workspace_bytes = pda_score_topk4_workspace_bytes(n_users_blk, n_items_local, d, p.n_splits);
    } else {
        p.path = ordered ? PDA_PATH_GEN3_ORDERED : PDA_PATH_GEN3;    // pda_score_topk_ordered_* | pda_score_topk_prepped_f32 / _bf16
        p.order = !ordered ? PDA_ORDER_NATURAL : (head == PDA_HEAD_POP ? PDA_ORDER_BY_POPULARITY : PDA_ORDER_BY_NORM);
        p.prep_with_pop = head == PDA_HEAD_POP && ordered ? 1 : 0;
        p.n_splits = pda_score_topk_auto_splits(n_users_blk, n_items_local);
        p.early_stop = early ? 1 : 0;
        p.workspace_bytes = pda_score_topk_workspace_bytes(n_users_blk);
    }
    p.keys_rows = (size_t)p.n_splits * (size_t)n_users_blk;
    *plan = p;
    return PDA_OK;
}

// ---- debug / measurement entry of the funnel's emitting sweep (tools/time_emit.py): one launch against the caller's thresholds ----------
// workspace: [256 B counters | user image | norms | eu per wave | cursors | lists]; offs[0..4] = byte offsets of (unorm, eu, cursors, lists, end)
extern "C" size_t pda_debug_emit_layout(int n_users_blk, int d, int n_splits, int cap_e, size_t* offs) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const int ut = d == 256 ? 512 : 1024, nu = ut / 64;
    const size_t utiles = ((size_t)n_users_blk + ut - 1) / ut, n_pad = utiles * ut;
    size_t b = 256;
    b = al(b + n_pad * 2 * (size_t)d);
    offs[0] = b;
    b = al(b + n_pad * 4);
    offs[1] = b;
    b = al(b + utiles * 4 * 8);
    offs[2] = b;
    b = al(b + utiles * (size_t)n_splits * 4 * nu * 64 * 4);
    offs[3] = b;
    b = al(b + utiles * (size_t)n_splits * 4 * (size_t)cap_e * (64 * nu * 48) + (1 << 18));
    offs[4] = b;                               // (the rows' residual norms sit behind everything)
    return al(b + n_pad * 4);
}
extern "C" int pda_debug_emit_sweep(const void* U, int bf16, const int32_t* users, int n_users_blk, const void* prep, int n_items_local, int d, const float* thr,
                                    int tile_lo, int tile_hi, int n_splits, int cap_e, void* workspace, void* stream) {
    if (!U || !users || !prep || !thr || !workspace || n_users_blk <= 0 || n_splits <= 0 || cap_e <= 0) return PDA_ERR_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    size_t offs[5];
    pda_debug_emit_layout(n_users_blk, d, n_splits, cap_e, offs);
    const Prep4Layout L = prep4_layout(n_items_local, d);
    const unsigned char* pb = reinterpret_cast<const unsigned char*>(prep);
    unsigned char* wsb = reinterpret_cast<unsigned char*>(workspace);
    Args7 g{pb + L.rows5, reinterpret_cast<const float*>(pb + L.meta5), wsb + 256, reinterpret_cast<const float*>(wsb + offs[0]),
            reinterpret_cast<const float*>(wsb + offs[4]), thr, wsb + offs[3],
            reinterpret_cast<unsigned*>(wsb + offs[2]), reinterpret_cast<float*>(wsb + offs[1]), reinterpret_cast<unsigned*>(wsb), n_users_blk, n_splits, L.n_tiles,
            tile_lo, tile_hi, cap_e, nullptr, nullptr, reinterpret_cast<const int*>(pb + L.hdr)};
#define PDA_E7(DD, BFV, UPWV)                                                                                                                      \
    {                                                                                                                                              \
        constexpr int UT = 4 * UPWV;                                                                                                               \
        const int n_pad = (n_users_blk + UT - 1) / UT * UT;                                                                                        \
        hipLaunchKernelGGL((uprep5_kernel<DD, BFV, true, UPWV, true>), dim3((unsigned)(((size_t)n_pad * (DD / 8) + 255) / 256)), dim3(256), 0, s, U, users, n_users_blk, n_pad, \
                           wsb + 256, reinterpret_cast<float*>(wsb + offs[0]), reinterpret_cast<float*>(wsb + offs[4]));                           \
        PDA_CHECK_LAUNCH();                                                                                                                        \
        return launch_sweep7<DD, BFV, UPWV>(g, s);                                                                                                 \
    }
    if (d == 64) { if (bf16) PDA_E7(64, true, 256) else PDA_E7(64, false, 256) }
    if (d == 128) { if (bf16) PDA_E7(128, true, 256) else PDA_E7(128, false, 256) }
    if (d == 256) { if (bf16) PDA_E7(256, true, 128) else PDA_E7(256, false, 128) }
#undef PDA_E7
    return PDA_ERR_UNSUPPORTED;
}

