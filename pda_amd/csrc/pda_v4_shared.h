// Pieces shared by the translation units of the pre-filtered score + top-K kernels: pda_score_topk_v4.hip (generation 4, the huge geometry) and
// pda_score_funnel.hip (the funnel, round 5).  The item prep layout, the tile sequences of item splits, the user image of the huge geometry.
#pragma once
#include "pda_topk_common.h"

using namespace pda_topk;

// defined in pda_score_topk_v4.hip: generation 4 on a block whose size is a device-side count (the funnel's exact fallback)
int pda_v4_run_score4_dev(const void* U, const void* I_shard, bool bf16, const void* prep, const float* pop_shard, const int32_t* users, int n_users_blk,
                          const int* n_users_dev, int item_offset, int n_items_local, int d, const int64_t* hist_indptr, const int32_t* hist_indices,
                          int hist_row_mode, int K, int head, int early_stop, int n_splits, const float* seed, uint64_t* out_keys, void* workspace, hipStream_t s);

namespace {

__host__ __device__ constexpr int row_bytes(int d) { return 2 * d + 48; }
__host__ __device__ constexpr int tile_bytes(int d) { return 64 * row_bytes(d); }

struct Prep4Layout {
    size_t hdr, pos_of, sufA, sufB, sufR, rows, rows5, meta5, pinfo, total;
    int n_tiles;
};
Prep4Layout prep4_layout(int n, int d) {
    Prep4Layout L{};
    L.n_tiles = (n + 63) / 64;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    L.hdr = 0;                                  // +0 int: "order is not a permutation"; +4 int: prep was built with a popularity; +8 int: with an order
    L.pos_of = 256;
    L.sufA = L.pos_of + al((size_t)n * 4);
    L.sufB = L.sufA + al((size_t)L.n_tiles * 4);
    L.sufR = L.sufB + al((size_t)L.n_tiles * 4);
    L.rows = L.sufR + al((size_t)L.n_tiles * 4);
    L.total = L.rows + (size_t)L.n_tiles * tile_bytes(d);
    // the image of the huge geometry -- rows scaled by the popularity, bf16, 16-byte chunks XOR-swizzled, 32-item half-tiles of
    // 64 d bytes -- and (pmax, nmax) per half-tile
    L.rows5 = al(L.total);
    L.meta5 = L.rows5 + al((size_t)L.n_tiles * 2 * 64 * (size_t)d);
    L.pinfo = L.meta5 + al((size_t)L.n_tiles * 2 * 16);              // (pmax, nmax, rmax, 0) per half-tile: one 16-byte LDS-DMA lane
    // per visiting position, 16 bytes: (||i|| padded, ||i' - bf16(i')|| padded, local id, popularity) -- what the funnel's selection gathers per pair: a
    // dense 3 MB array (L2 / MALL resident) instead of the tails of the 2 d + 48-byte rows
    L.total = L.pinfo + al((size_t)L.n_tiles * 64 * 16);
    return L;
}

// the tiles of split s: s, s + S, s + 2 S, ...
__device__ __forceinline__ int split_tiles(int n_tiles, int split, int n_splits) {
    return split < n_tiles ? (n_tiles - split + n_splits - 1) / n_splits : 0;
}

constexpr int kUT5 = 1024;            // users per workgroup
constexpr int kNSlot5 = 8;            // half-tile slots in the LDS (pda_v5_loop_asm.h: NSLOT), Loop5<D>::kSlotBytes each: the rows, then the meta entry

__host__ __device__ constexpr int half_bytes5(int d) { return 64 * d; }       // 32 rows of 2 d bytes, 16-byte chunks XOR-swizzled
// an LDS slot: the half-tile's rows, then its meta entry; slots start at multiples of 256 (512 at d = 256: the loop forms fragment addresses by
// XOR with k << 6, which reaches bit 8 from k = 4 on)
__host__ __device__ constexpr int slot_bytes5(int d) { return 64 * d + (d == 256 ? 512 : 256); }
template <int D>
__device__ __forceinline__ int swz5(int row) { return D >= 128 ? (row & 15) : ((row >> 1) & 7); }

// ---- the user image: the block's rows as bf16 MFMA operands, in the order the waves load them into their AGPRs ------------------
// fragment (workgroup wg, wave w, user block u, k-step k): 64 lanes x 16 bytes = 16 users x 32 elements; lane l holds user 16 u + (l & 15),
// elements 32 k + 8 (l >> 4) .. + 7  (the B operand of v_mfma_f32_16x16x32_bf16; S16 = true keeps the kernels' round-4 names)
// eight floats as fp16 (RNE, clamped to the largest finite half: the funnel's bound is formed from the ACTUAL residuals, so any rounding -- an underflow, a
// clamp -- is accounted for), and the exact residuals' squared norm
__device__ __forceinline__ u32x4 pack_half8(const f32x4& a, const f32x4& b, float& rs) {
    u32x4 h;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float x0 = k < 2 ? a[2 * k] : b[2 * (k - 2)], x1 = k < 2 ? a[2 * k + 1] : b[2 * (k - 2) + 1];
        const _Float16 h0 = (_Float16)fminf(fmaxf(x0, -65504.0f), 65504.0f), h1 = (_Float16)fminf(fmaxf(x1, -65504.0f), 65504.0f);
        const float r0 = x0 - (float)h0, r1 = x1 - (float)h1;
        rs += r0 * r0 + r1 * r1;
        h[k] = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
    }
    return h;
}

// F16: the image in fp16 for v_mfma_f32_16x16x32_f16 (the funnel: eleven significant bits instead of eight -- the rounding residuals, hence the bound's band, 8 x smaller)
template <int D, bool BF, bool S16, int UPW, bool F16 = false>
__global__ void __launch_bounds__(256) uprep5_kernel(const void* __restrict__ U, const int32_t* __restrict__ users, int n_users_blk, int n_pad,
                                                     unsigned char* __restrict__ ufrag, float* __restrict__ unorm, float* __restrict__ uerr = nullptr) {
    constexpr int TPR = D / 8;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int rb = gid / TPR, c = gid % TPR;
    if (rb >= n_pad) return;
    f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = x;
    if (rb < n_users_blk) {
        const int uid = users[rb];
        x = pda_load4<BF>(U, (size_t)uid * D + 8 * c);
        y = pda_load4<BF>(U, (size_t)uid * D + 8 * c + 4);
    }
    u32x4 hq, lq;
    float ss = 0.f, rs = 0.f;
    if constexpr (F16) {
        hq = pack_half8(x, y, rs);
#pragma unroll
        for (int k = 0; k < 4; ++k) ss += x[k] * x[k] + y[k] * y[k];
    } else {
        split8(x, y, hq, lq);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ss += x[k] * x[k] + y[k] * y[k];
            const float r0 = x[k] - __uint_as_float((k & 1) ? (hq[k >> 1] & 0xFFFF0000u) : (hq[k >> 1] << 16));       // exact rounding residuals
            const float r1 = y[k] - __uint_as_float((k & 1) ? (hq[2 + (k >> 1)] & 0xFFFF0000u) : (hq[2 + (k >> 1)] << 16));
            rs += r0 * r0 + r1 * r1;
        }
    }
#pragma unroll
    for (int o = TPR / 2; o > 0; o >>= 1) { ss += __shfl_xor(ss, o, 64); rs += __shfl_xor(rs, o, 64); }
    static_assert(S16, "the 32 x 32 x 16 image left with its loop (round 5)");
    constexpr int NK = D / 32, NU = UPW / 16;                        // (wgw: the wave's index among all waves of the launch)
    const int wgw = rb / UPW, u = (rb >> 4) & (NU - 1), j = rb & 15, k = c >> 2, g4 = c & 3;
    *reinterpret_cast<u32x4*>(ufrag + ((((size_t)wgw * NU + u) * NK + k) * 64 + (j + 16 * g4)) * 16) = hq;
    if (c == 0) unorm[rb] = sqrtf(ss) * 1.0009765625f * 1.0001f;          // padded ||u|| (as generation 4's nu_row)
    if (c == 0 && uerr != nullptr) uerr[rb] = sqrtf(rs) * 1.0009765625f * 1.0001f;       // padded ||u - bf16(u)|| (the funnel's bound)
}

// Train items at the candidate stage: a candidate that passes the exact threshold must not be a train item of its user -- a
// binary search in the user's id-sorted history, i.e. six to seven DEPENDENT loads from a 200 MB array: 5 of the 6.5 us of a
// rescoring pass of 16 candidates (cycle counters, natural-order sweep).  A 1024-bit Bloom filter per block row (two hashes;
// 50 train items: 0.9 % false positives) is read with the candidate's rows instead; only a hit pays for the search.
__device__ __forceinline__ unsigned bloom_h1(int item) { return ((unsigned)item * 0x9E3779B1u) >> 22; }
__device__ __forceinline__ unsigned bloom_h2(int item) { return ((unsigned)item * 0x85EBCA6Bu + 0x27D4EB2Fu) >> 22; }
// 32 rows per workgroup, eight lanes per row: each lane hashes every eighth train item into the row's 32 words in LDS.
// skip_if_ordered: sweeps of the popularity head in visiting order meet next to no candidates -- the filters would cost more
// than they save (prep header word 2 = "an order was given"; the sweep reads the same word).
__global__ void __launch_bounds__(256) hist_bloom4_kernel(const int32_t* __restrict__ users, const int64_t* __restrict__ indptr,
                                                          const int32_t* __restrict__ indices, int hist_row_mode, int n_users_blk,
                                                          uint32_t* __restrict__ bloom, const int* __restrict__ prep_hdr, int skip_if_ordered) {
    if (skip_if_ordered && prep_hdr[2] != 0) return;
    __shared__ uint32_t w[32 * 32];
    const int tid = threadIdx.x, r = tid >> 3, sub = tid & 7;
    for (int q = tid; q < 32 * 32; q += 256) w[q] = 0u;
    __syncthreads();
    const int u = (int)blockIdx.x * 32 + r;
    if (u < n_users_blk) {
        const int64_t hr = hist_row_mode == PDA_HIST_BY_USER_ID ? (int64_t)users[u] : (int64_t)u;
        const int64_t b = indptr[hr], e = indptr[hr + 1];
        for (int64_t i0 = b + sub; i0 < e; i0 += 8 * 4) {          // (four loads of a lane in flight: a 150-item row was 19 dependent round trips)
            int item[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) item[q] = i0 + 8 * q < e ? indices[i0 + 8 * q] : -1;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (i0 + 8 * q < e) {
                    const unsigned h1 = bloom_h1(item[q]), h2 = bloom_h2(item[q]);
                    atomicOr(&w[r * 32 + (h1 >> 5)], 1u << (h1 & 31u));
                    atomicOr(&w[r * 32 + (h2 >> 5)], 1u << (h2 & 31u));
                }
        }
    }
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * 32 * 32;
    for (int q = tid; q < 32 * 32; q += 256)
        if ((int)blockIdx.x * 32 + (q >> 5) < n_users_blk) bloom[base + q] = w[q];
}

}  // namespace
