// Item-side preparation of the pre-filtered score kernels of generation 3 (pda_score_topk_v3.hip) and their C entry points:
// bf16 planes, padded row norms, the 16 bf16 per item of the folded threshold test, visiting order, suffix bounds, history
// rows in visiting positions.  (Generation 2 -- three bf16 MFMAs per k-step on a hi/lo split, approximate per-user lists with
// an exact finish -- lived in this file until round 2; generation 4 serves its last use, the early-terminating sweep over
// bf16 tables at d = 256, faster: 3.5 vs 4.5 ms per 65 536 users on a config-5 shard.  The lo plane it needed is still
// written: the layout of the prep buffer is part of the ABI.)
#include "pda_topk_common.h"
#include <cstdlib>

using namespace pda_topk;

namespace {



// one row per D/8 threads: fp32 -> bf16 hi, bf16 lo, padded norm
// BF: the table is bf16 already -- there is nothing to split; `hi` (may be NULL) receives the row when it has to be
// gathered into visiting order, `lo` is unused.
template <int D, bool BF>
__global__ void __launch_bounds__(256) item_prep_kernel(const void* __restrict__ I, int n, uint16_t* __restrict__ hi,
                                                        uint16_t* __restrict__ lo, float* __restrict__ nrm, int* __restrict__ nrm_max_bits,
                                                        const int* __restrict__ order, const float* __restrict__ pop,
                                                        float* __restrict__ pop_p, int* __restrict__ pos_of, int* __restrict__ bad,
                                                        uint16_t* __restrict__ bex) {
    constexpr int TPR = D / 8;
    const int row = blockIdx.x * (256 / TPR) + threadIdx.x / TPR, e = threadIdx.x % TPR;
    float ss = 0.f, popv = 1.0f;
    if (row < n) {
        int src = row;
        if (pop) popv = pop[order ? min(max(order[row], 0), n - 1) : row];
        if (order) {                      // ordered prep: position `row` holds item order[row]
            src = order[row];
            if (src < 0 || src >= n) { if (e == 0) atomicOr(bad, 1); src = 0; }
            else if (e == 0) {
                if (atomicExch(&pos_of[src], row) != -1) atomicOr(bad, 1);   // not a permutation
                if (pop) pop_p[row] = pop[src];
            }
        }
        const f32x4 a = pda_load4<BF>(I, (size_t)src * D + 8 * e);
        const f32x4 b = pda_load4<BF>(I, (size_t)src * D + 8 * e + 4);
        if constexpr (BF) {
            if (hi) *reinterpret_cast<u32x4*>(hi + (size_t)row * D + 8 * e) =
                        *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(I) + (size_t)src * D + 8 * e);
        } else {
            u32x4 h, l;
            split8(a, b, h, l);
            *reinterpret_cast<u32x4*>(hi + (size_t)row * D + 8 * e) = h;
            *reinterpret_cast<u32x4*>(lo + (size_t)row * D + 8 * e) = l;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) ss += a[k] * a[k] + b[k] * b[k];
    }
#pragma unroll
    for (int o = TPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (row < n && e == 0) {
        const float v = sqrtf(ss) * 1.0009765625f * 1.0001f;
        nrm[row] = v;
        atomicMax(nrm_max_bits, __float_as_int(v));   // v >= 0: integer order == float order
        // The extra k-step of v3's folded test (pda_score_topk_v3.hip): with the A side holding the pieces of -threshold, -1
        // and +eps_scale, one more MFMA turns the accumulator into  s~ - thr * (1/pop)' + (1 + eps)  -- "candidate" is then
        // "accumulator > 0".  (1/pop)' is rounded DOWN and capped (conservative for thr > 0; thr <= 0 takes the kernel's
        // pop > thr path), split into three bf16 pieces so that the eight products carry thr/pop to 2^-30.
        uint32_t p1 = 0x3F80u, p2 = 0, p3 = 0, k1 = 0, k2 = 0;   // PDA_HEAD_RAW: 1/pop := 1, constant := +8e-6 slack
        if (pop) {
            float ip = (popv > 0.f) ? fminf((1.0f / popv) * 0.9999995f, 1.0e6f) : 1.0e6f;
            bf16_split3(ip, p1, p2, p3);
            k1 = 0x3F80u;
            k2 = bf16_up(8.0e-6f);
            if (!(popv == popv)) k1 = 0xFF61u;   // NaN popularity: never a candidate (the exact kernels' comparisons are false too)
        } else {
            k1 = bf16_up(8.0e-6f);
        }
        u32x4 lo4, hi4;
        lo4[0] = p1 | (p2 << 16);
        lo4[1] = p1 | (p2 << 16);
        lo4[2] = p3 | (p1 << 16);
        lo4[3] = p3 | (p2 << 16);
        hi4[0] = k1 | (k2 << 16);
        hi4[1] = bf16_up(v);
        hi4[2] = 0;
        hi4[3] = 0;
        *reinterpret_cast<u32x4*>(bex + (size_t)row * 16) = lo4;
        *reinterpret_cast<u32x4*>(bex + (size_t)row * 16 + 8) = hi4;
    }
}

// per-tile maxima of |pop| and |pop|*||i|| in visiting order, then their suffix maxima (single workgroup)
__global__ void __launch_bounds__(256) tile_bound_kernel(const float* __restrict__ pop_p, const float* __restrict__ nrm, int n, int n_tiles,
                                                         float* __restrict__ tA, float* __restrict__ tB) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    float ma = 0.f, mb = 0.f;
    for (int q = 0; q < 32; ++q) {
        const int i = t * 32 + q;
        if (i < n) {
            const float pa = pop_p ? fabsf(pop_p[i]) : 0.f, pb = pop_p ? pa * nrm[i] * 1.000001f : nrm[i];
            ma = fmaxf(ma, pa);
            mb = fmaxf(mb, pb);
        }
    }
    tA[t] = ma;
    tB[t] = mb;
}
__global__ void __launch_bounds__(1024) suffix_max_kernel(float* __restrict__ tA, float* __restrict__ tB, int n_tiles) {
    __shared__ float sa[1024], sb[1024];
    const int per = (n_tiles + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, n_tiles);
    float ma = 0.f, mb = 0.f;
    for (int t = lo; t < hi; ++t) { ma = fmaxf(ma, tA[t]); mb = fmaxf(mb, tB[t]); }
    sa[threadIdx.x] = ma;
    sb[threadIdx.x] = mb;
    __syncthreads();
    float ra = 0.f, rb = 0.f;                      // max over the chunks behind mine
    for (int q = threadIdx.x + 1; q < 1024; ++q) { ra = fmaxf(ra, sa[q]); rb = fmaxf(rb, sb[q]); }
    for (int t = hi - 1; t >= lo; --t) {
        ra = fmaxf(ra, tA[t]);
        rb = fmaxf(rb, tB[t]);
        tA[t] = ra;
        tB[t] = rb;
    }
}

// History rows rewritten in visiting positions (in-shard item ids -> item_offset + pos_of[id - item_offset]) and sorted
// again.  One wave per row: bitonic sort in LDS up to 2048 entries, rank sort in global memory beyond.
__global__ void __launch_bounds__(64) hist_reorder_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                          const int* __restrict__ pos_of, int item_offset, int n_items_local,
                                                          int32_t* __restrict__ out) {
    __shared__ int buf[2048];
    const int lane = threadIdx.x;
    const int64_t b = indptr[blockIdx.x], e = indptr[blockIdx.x + 1];
    const int64_t L = e - b;
    if (L <= 0) return;
    auto mapped = [&](int64_t i) __attribute__((always_inline)) {
        const int v = indices[b + i];
        const int loc = v - item_offset;
        return (loc >= 0 && loc < n_items_local) ? item_offset + pos_of[loc] : v;
    };
    if (L <= 2048) {
        int P = 64;
        while (P < L) P <<= 1;
        for (int i = lane; i < P; i += 64) buf[i] = i < L ? mapped(i) : 0x7fffffff;
        __syncthreads();
        for (int k = 2; k <= P; k <<= 1)
            for (int jst = k >> 1; jst > 0; jst >>= 1) {
                for (int q = lane; q < P / 2; q += 64) {
                    const int i = ((q & ~(jst - 1)) << 1) | (q & (jst - 1)), p = i | jst;
                    const int x = buf[i], y = buf[p];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { buf[i] = y; buf[p] = x; }
                }
                __syncthreads();
            }
        for (int i = lane; i < L; i += 64) out[b + i] = buf[i];
    } else {
        for (int64_t i = lane; i < L; i += 64) {
            const int v = mapped(i);
            int64_t rank = 0;
            for (int64_t q = 0; q < L; ++q) {
                const int w = mapped(q);
                rank += (w < v || (w == v && q < i)) ? 1 : 0;
            }
            out[b + rank] = v;
        }
    }
}

__global__ void __launch_bounds__(256) pop_max_kernel(const float* __restrict__ pop, int n, int* __restrict__ out_bits) {
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = fmaxf(m, fabsf(pop[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_int(m));
}

}  // namespace

static inline size_t prep_plane_bytes(int n, int d) { return (((size_t)n * d * 2) + 255) & ~(size_t)255; }
static inline size_t prep_norm_bytes(int n) { return (((size_t)n * 4) + 255) & ~(size_t)255; }

namespace {
// prep blob: [norm f32 n][max-norm bits, bad-order flag (256 B)] then, ordered only, [pop_p f32 n][order i32 n]
// [pos_of i32 n][sufA f32 n_tiles][sufB f32 n_tiles]; the bf16 planes come LAST (2 for fp32 tables, 0 / 1 for bf16
// tables unordered / ordered) so that the offsets of the small arrays do not depend on the table type.
struct PrepLayout {
    size_t plane, hi, lo, norm, nmax, pop_p, order, pos_of, sufA, sufB, bex, total;
};
PrepLayout prep_layout(int n, int d, bool ordered, int n_planes) {
    PrepLayout L{};
    L.plane = prep_plane_bytes(n, d);
    L.norm = 0;
    L.nmax = L.norm + prep_norm_bytes(n);      // int bits of the max norm at +0, "order is not a permutation" flag at +4
    size_t off = L.nmax + 256;
    if (ordered) {
        const size_t tiles = (((size_t)(n + 31) / 32) * 4 + 255) & ~(size_t)255;
        L.pop_p = off;
        L.order = L.pop_p + prep_norm_bytes(n);
        L.pos_of = L.order + prep_norm_bytes(n);
        L.sufA = L.pos_of + prep_norm_bytes(n);
        L.sufB = L.sufA + tiles;
        off = L.sufB + tiles;
    }
    L.bex = off;
    off += ((size_t)n * 32 + 255) & ~(size_t)255;
    L.hi = off;
    L.lo = off + L.plane;
    L.total = off + (size_t)n_planes * L.plane;
    return L;
}
inline int planes_of(bool bf16, bool ordered) { return bf16 ? (ordered ? 1 : 0) : 2; }

int run_item_prep(const void* I_shard, bool bf16, const float* pop, const int* order, int n, int d, void* prep, hipStream_t s) {
    const bool ordered = order != nullptr;
    const PrepLayout L = prep_layout(n, d, ordered, planes_of(bf16, ordered));
    char* pb = reinterpret_cast<char*>(prep);
    uint16_t* hi = planes_of(bf16, ordered) > 0 ? reinterpret_cast<uint16_t*>(pb + L.hi) : nullptr;
    uint16_t* lo = planes_of(bf16, ordered) > 1 ? reinterpret_cast<uint16_t*>(pb + L.lo) : nullptr;
    float* nrm = reinterpret_cast<float*>(pb + L.norm);
    int* nmax = reinterpret_cast<int*>(pb + L.nmax);
    float* pop_p = ordered && pop ? reinterpret_cast<float*>(pb + L.pop_p) : nullptr;
    int* pos_of = ordered ? reinterpret_cast<int*>(pb + L.pos_of) : nullptr;
    uint16_t* bex = reinterpret_cast<uint16_t*>(pb + L.bex);
    if (hipMemsetAsync(nmax, 0, 256, s) != hipSuccess) return PDA_ERR_LAUNCH;
    if (ordered) {
        if (hipMemsetAsync(pos_of, 0xFF, (size_t)n * 4, s) != hipSuccess) return PDA_ERR_LAUNCH;
        if (hipMemcpyAsync(pb + L.order, order, (size_t)n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return PDA_ERR_LAUNCH;
    }
#define PDA_PREP(DD)                                                                                         \
    case DD: {                                                                                               \
        constexpr int RPB = 256 / (DD / 8);                                                                  \
        const dim3 grid((unsigned)((n + RPB - 1) / RPB));                                                    \
        if (bf16) hipLaunchKernelGGL((item_prep_kernel<DD, true>), grid, dim3(256), 0, s, I_shard, n, hi, lo, nrm, nmax, order, pop, pop_p, pos_of, nmax + 1, bex); \
        else hipLaunchKernelGGL((item_prep_kernel<DD, false>), grid, dim3(256), 0, s, I_shard, n, hi, lo, nrm, nmax, order, pop, pop_p, pos_of, nmax + 1, bex); \
        break;                                                                                               \
    }
    switch (d) {
        PDA_PREP(64) PDA_PREP(128) PDA_PREP(256)
        default: return PDA_ERR_UNSUPPORTED;
    }
#undef PDA_PREP
    PDA_CHECK_LAUNCH();
    if (ordered) {
        const int n_tiles = (n + 31) / 32;
        float* tA = reinterpret_cast<float*>(pb + L.sufA);
        float* tB = reinterpret_cast<float*>(pb + L.sufB);
        hipLaunchKernelGGL(tile_bound_kernel, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, s, pop_p, nrm, n, n_tiles, tA, tB);
        PDA_CHECK_LAUNCH();
        hipLaunchKernelGGL(suffix_max_kernel, dim3(1), dim3(1024), 0, s, tA, tB, n_tiles);
        PDA_CHECK_LAUNCH();
    }
    return PDA_OK;
}

int run_score_prepped(const void* U, const void* I_shard, bool bf16, const void* prep, bool ordered, const float* pop_shard,
                      const int32_t* users, int n_users_blk, int item_offset, int n_items_local, int d,
                      const int64_t* hist_indptr, const int32_t* hist_indices, const int32_t* hist_indices_ord,
                      int hist_row_mode, int K, int head, int early_stop, int n_splits, uint64_t* out_keys, void* workspace,
                      hipStream_t s) {
    if (!U || !I_shard || !prep || !users || !out_keys || !workspace) return PDA_ERR_ARG;
    if (n_users_blk <= 0 || n_items_local <= 0 || item_offset < 0) return PDA_ERR_ARG;
    if (K < 1 || K > PDA_TOPK_CAP - 4) return PDA_ERR_ARG;
    if (head != PDA_HEAD_RAW && head != PDA_HEAD_POP) return PDA_ERR_ARG;
    if (head == PDA_HEAD_POP && !pop_shard) return PDA_ERR_ARG;
    if (hist_indptr && (!hist_indices || (ordered && !hist_indices_ord))) return PDA_ERR_ARG;
    if (n_splits <= 0) n_splits = pda_score_topk_auto_splits(n_users_blk, n_items_local);
    const PrepLayout L = prep_layout(n_items_local, d, ordered, planes_of(bf16, ordered));
    const char* pb = reinterpret_cast<const char*>(prep);
    int* ws = reinterpret_cast<int*>(workspace);
    if (hipMemsetAsync(workspace, 0, pda_score_topk_workspace_bytes(n_users_blk), s) != hipSuccess) return PDA_ERR_LAUNCH;
    if (head == PDA_HEAD_POP) {
        hipLaunchKernelGGL(pop_max_kernel, dim3(64), dim3(256), 0, s, pop_shard, n_items_local, ws);
        PDA_CHECK_LAUNCH();
    }
    // bf16 tables, natural order: the B tiles are read from the table itself
    const uint16_t* plane_hi = (bf16 && !ordered) ? reinterpret_cast<const uint16_t*>(I_shard) : reinterpret_cast<const uint16_t*>(pb + L.hi);
    ScoreArgs2 aa{{reinterpret_cast<const float*>(U), reinterpret_cast<const float*>(I_shard), pop_shard, users, hist_indptr,
                   ordered ? hist_indices_ord : hist_indices, out_keys, n_users_blk, item_offset, n_items_local, hist_row_mode, K,
                   n_splits, nullptr},
                  plane_hi, reinterpret_cast<const uint16_t*>(pb + L.lo),
                  reinterpret_cast<const float*>(pb + L.norm), reinterpret_cast<const float*>(pb + L.nmax),
                  reinterpret_cast<const float*>(ws), ws + 4,
                  ordered ? reinterpret_cast<const int*>(pb + L.order) : nullptr,
                  ordered ? reinterpret_cast<const float*>(pb + L.pop_p) : nullptr,
                  ordered && early_stop ? reinterpret_cast<const float*>(pb + L.sufA) : nullptr,   // NULL: visit everything
                  ordered && early_stop ? reinterpret_cast<const float*>(pb + L.sufB) : nullptr,
                  reinterpret_cast<unsigned long long*>(ws + 2),
                  reinterpret_cast<const uint16_t*>(pb + L.bex), hist_indices};
    // generation 3: 1 MFMA per k-step, candidate ring, exact lists, exact fp32-MFMA warm-up of the first tiles (d <= 128).
    // It packs (row, item id) into 32-bit ring words and uses 32-bit plane offsets: larger shards take the exact kernel.
    const bool v3_fits = (uint64_t)item_offset + (uint64_t)n_items_local <= (1ull << 27) && (uint64_t)n_items_local * (uint64_t)d < (1ull << 32);
    if (v3_fits) return pda_topk::launch_score_v3(aa, d, head, ordered, bf16, s);
    if (bf16) return PDA_ERR_UNSUPPORTED;
    {
        ScoreArgs v1 = aa.a;
        v1.hist_indices = hist_indices;
        return pda_topk::launch_score_v1(v1, d, head, s, false);
    }
}
}  // namespace

extern "C" size_t pda_item_prep_bytes(int n_items_local, int d) { return prep_layout(n_items_local, d, false, 2).total; }
extern "C" size_t pda_item_prep_ordered_bytes(int n_items_local, int d) { return prep_layout(n_items_local, d, true, 2).total; }
extern "C" size_t pda_item_prep_bf16_bytes(int n_items_local, int d) { return prep_layout(n_items_local, d, false, 0).total; }
extern "C" size_t pda_item_prep_ordered_bf16_bytes(int n_items_local, int d) { return prep_layout(n_items_local, d, true, 1).total; }

extern "C" size_t pda_score_topk_workspace_bytes(int n_users_blk) {
    // [max |pop| f32][pad][u64 item tiles scored, summed over workgroups][tile_flags i32 per 128-user tile]
    const size_t tiles = (size_t)(n_users_blk + kUserTile - 1) / kUserTile;
    return (16 + tiles * 4 + 255) & ~(size_t)255;
}

extern "C" int pda_item_prep_f32(const float* I_shard, int n_items_local, int d, void* prep, void* stream) {
    if (!I_shard || !prep || n_items_local <= 0) return PDA_ERR_ARG;
    return run_item_prep(I_shard, false, nullptr, nullptr, n_items_local, d, prep, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pda_item_prep_bf16(const uint16_t* I_shard, int n_items_local, int d, void* prep, void* stream) {
    if (!I_shard || !prep || n_items_local <= 0) return PDA_ERR_ARG;
    return run_item_prep(I_shard, true, nullptr, nullptr, n_items_local, d, prep, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pda_item_prep_ordered_f32(const float* I_shard, const float* pop_shard, const int32_t* order, int n_items_local,
                                         int d, void* prep, void* stream) {
    if (!I_shard || !order || !prep || n_items_local <= 0) return PDA_ERR_ARG;
    return run_item_prep(I_shard, false, pop_shard, order, n_items_local, d, prep, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pda_item_prep_ordered_bf16(const uint16_t* I_shard, const float* pop_shard, const int32_t* order, int n_items_local,
                                          int d, void* prep, void* stream) {
    if (!I_shard || !order || !prep || n_items_local <= 0) return PDA_ERR_ARG;
    return run_item_prep(I_shard, true, pop_shard, order, n_items_local, d, prep, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pda_item_prep_ordered_check(const void* prep, int n_items_local, int d, void* stream) {
    if (!prep || n_items_local <= 0) return PDA_ERR_ARG;
    const PrepLayout L = prep_layout(n_items_local, d, true, 0);
    int bad = 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(&bad, reinterpret_cast<const char*>(prep) + L.nmax + 4, 4, hipMemcpyDeviceToHost, s) != hipSuccess) return PDA_ERR_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return PDA_ERR_LAUNCH;
    return bad ? PDA_ERR_ARG : PDA_OK;
}

extern "C" int pda_hist_reorder(const void* prep, int n_items_local, int d, int item_offset, const int64_t* hist_indptr,
                                const int32_t* hist_indices, int n_rows, int32_t* out_indices, void* stream) {
    if (!prep || !hist_indptr || !hist_indices || !out_indices || n_items_local <= 0 || n_rows < 0 || item_offset < 0) return PDA_ERR_ARG;
    if (n_rows == 0) return PDA_OK;
    const PrepLayout L = prep_layout(n_items_local, d, true, 0);      // the small arrays sit in front of the planes
    hipLaunchKernelGGL(hist_reorder_kernel, dim3((unsigned)n_rows), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), hist_indptr,
                       hist_indices, reinterpret_cast<const int*>(reinterpret_cast<const char*>(prep) + L.pos_of), item_offset,
                       n_items_local, out_indices);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}

#define PDA_SCORE_ARGS_DECL                                                                                                   \
    const float* pop_shard, const int32_t* users, int n_users_blk, int item_offset, int n_items_local, int d,                \
        const int64_t* hist_indptr, const int32_t* hist_indices
#define PDA_SCORE_TAIL_DECL int hist_row_mode, int K, int head, int n_splits, uint64_t* out_keys, void* workspace, void* stream
#define PDA_SCORE_TAIL_ORD_DECL \
    int hist_row_mode, int K, int head, int early_stop, int n_splits, uint64_t* out_keys, void* workspace, void* stream

extern "C" int pda_score_topk_prepped_f32(const float* U, const float* I_shard, const void* prep, PDA_SCORE_ARGS_DECL,
                                          PDA_SCORE_TAIL_DECL) {
    return run_score_prepped(U, I_shard, false, prep, false, pop_shard, users, n_users_blk, item_offset, n_items_local, d, hist_indptr,
                             hist_indices, nullptr, hist_row_mode, K, head, 0, n_splits, out_keys, workspace,
                             reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pda_score_topk_ordered_f32(const float* U, const float* I_shard, const void* prep, PDA_SCORE_ARGS_DECL,
                                          const int32_t* hist_indices_ord, PDA_SCORE_TAIL_ORD_DECL) {
    return run_score_prepped(U, I_shard, false, prep, true, pop_shard, users, n_users_blk, item_offset, n_items_local, d, hist_indptr,
                             hist_indices, hist_indices_ord, hist_row_mode, K, head, early_stop, n_splits, out_keys, workspace,
                             reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pda_score_topk_bf16(const uint16_t* U, const uint16_t* I_shard, const void* prep, PDA_SCORE_ARGS_DECL,
                                   PDA_SCORE_TAIL_DECL) {
    return run_score_prepped(U, I_shard, true, prep, false, pop_shard, users, n_users_blk, item_offset, n_items_local, d, hist_indptr,
                             hist_indices, nullptr, hist_row_mode, K, head, 0, n_splits, out_keys, workspace,
                             reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pda_score_topk_ordered_bf16(const uint16_t* U, const uint16_t* I_shard, const void* prep, PDA_SCORE_ARGS_DECL,
                                           const int32_t* hist_indices_ord, PDA_SCORE_TAIL_ORD_DECL) {
    return run_score_prepped(U, I_shard, true, prep, true, pop_shard, users, n_users_blk, item_offset, n_items_local, d, hist_indptr,
                             hist_indices, hist_indices_ord, hist_row_mode, K, head, early_stop, n_splits, out_keys, workspace,
                             reinterpret_cast<hipStream_t>(stream));
}
