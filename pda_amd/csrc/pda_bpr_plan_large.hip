// The plan of a LARGE batch (B > 4096 triplets) for the exact SGD step without atomics (pda_bpr_plan.hip, whose launches A and B read
// the same layout): pda_triplet_plan sorts the 2B item references of a batch inside one workgroup's LDS; here they go through a
// device-wide stable radix sort (rocPRIM, a library sort: the plan is the sampler's work, batches ahead of the step).
//
//   fill      keys[i] = pos ++ neg, vals[i] = i
//   sort      stable by item id  -> entries (= sorted vals: ascending inside a segment, the order launch B sums in) and sorted keys
//   heads     head[i] = "first reference of its item"; inclusive scan -> the segment of every sorted position
//   scatter   seg_item / seg_start, the header, the "referenced once" bits of every triplet; the list of very long segments
//             (pda_plan_common.h: launch B sums those with several workgroups)
//   users     sorted copy of the user ids; equal neighbours = a user occurs twice (the reference's sampler never does that,
//             MF/train_new_api.py:380-381): hdr[1] := 1, the step rejects the batch
//
// Reference semantics of the step itself: MF/model_api.py:83,102-121 ([TF-ext] IndexedSlices are summed per row, then applied).
#include <cstring>
#include <cstdlib>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "pda_common.h"
#include "pda_plan_common.h"

namespace {

// the layout of pda_bpr_plan.hip (plan_view)
struct PlanViewL {
    int* hdr;              // [4]: segments, "a user occurs twice", 2B, B
    int* seg_item;         // [2B]
    int* seg_start;        // [2B + 2]
    int* entries;          // [2B]
    unsigned char* flags;  // [B]
};
inline size_t plan_bytes_l(int B) {
    const size_t raw = 16 + (size_t)4 * (2 * B) + (size_t)4 * (2 * B + 2) + (size_t)4 * (2 * B) + (size_t)B;
    return (raw + 15) & ~(size_t)15;
}
inline PlanViewL plan_view_l(void* p, int B) {
    unsigned char* b = reinterpret_cast<unsigned char*>(p);
    PlanViewL v;
    v.hdr = reinterpret_cast<int*>(b);
    v.seg_item = v.hdr + 4;
    v.seg_start = v.seg_item + 2 * B;
    v.entries = v.seg_start + 2 * B + 2;
    v.flags = reinterpret_cast<unsigned char*>(v.entries + 2 * B);
    return v;
}

struct WsL {
    size_t keys_in, keys_out, vals_in, ukeys, head, seg_of, temp, temp_bytes, total;
};
WsL ws_layout(int B) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t n = 2 * (size_t)B;
    WsL w{};
    size_t t1 = 0, t2 = 0, t3 = 0;
    (void)rocprim::radix_sort_pairs(nullptr, t1, (const int*)nullptr, (int*)nullptr, (const int*)nullptr, (int*)nullptr, n, 0, 32, (hipStream_t)0);
    (void)rocprim::radix_sort_keys(nullptr, t2, (const int*)nullptr, (int*)nullptr, (size_t)B, 0, 32, (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, t3, (const int*)nullptr, (int*)nullptr, n, rocprim::plus<int>(), (hipStream_t)0);
    w.temp_bytes = t1 > t2 ? (t1 > t3 ? t1 : t3) : (t2 > t3 ? t2 : t3);
    w.keys_in = 0;
    w.keys_out = al(w.keys_in + n * 4);
    w.vals_in = al(w.keys_out + n * 4);
    w.ukeys = al(w.vals_in + n * 4);
    w.head = al(w.ukeys + (size_t)B * 4);
    w.seg_of = al(w.head + n * 4);
    w.temp = al(w.seg_of + n * 4);
    w.total = al(w.temp + w.temp_bytes);
    return w;
}

__global__ void __launch_bounds__(256) fill_kernel(const int32_t* __restrict__ pos, const int32_t* __restrict__ neg, int B, int* __restrict__ keys,
                                                   int* __restrict__ vals) {
    const int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i < 2 * B) {
        keys[i] = i < B ? pos[i] : neg[i - B];
        vals[i] = i;
    }
}

__global__ void __launch_bounds__(256) head_kernel(const int* __restrict__ ks, int n, int* __restrict__ head) {
    const int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i < n) head[i] = (i == 0 || ks[i] != ks[i - 1]) ? 1 : 0;
}

__global__ void __launch_bounds__(256) scatter_kernel(const int* __restrict__ ks, const int* __restrict__ head, const int* __restrict__ seg_of, int B,
                                                      PlanViewL v) {
    const int n = 2 * B;
    const int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (head[i]) {
        const int s = seg_of[i] - 1;
        v.seg_item[s] = ks[i];
        v.seg_start[s] = i;
        if (i == n - 1 || head[i + 1]) {                  // a segment of one: its triplet's row is referenced once in the batch
            const int r = v.entries[i];
            const int t = r < B ? r : r - B;
            atomicOr(reinterpret_cast<unsigned*>(v.flags) + (t >> 2), (r < B ? 1u : 2u) << (8 * (t & 3)));
        }
    }
    if (i == n - 1) {
        const int n_seg = seg_of[i];
        v.seg_start[n_seg] = n;
        v.hdr[0] = n_seg;
        v.hdr[2] = n;
        v.hdr[3] = B;
    }
}

// the very long segments (>= kXlMin references), listed from the end of seg_item backwards; their number in seg_start[2B + 1]
__global__ void __launch_bounds__(256) xl_kernel(int B, PlanViewL v) {
    const int s = (int)blockIdx.x * 256 + threadIdx.x;
    if (s < v.hdr[0] && v.seg_start[s + 1] - v.seg_start[s] >= kXlMin) v.seg_item[2 * B - 1 - atomicAdd(&v.seg_start[2 * B + 1], 1)] = s;
}

__global__ void __launch_bounds__(256) user_dup_kernel(const int* __restrict__ us, int B, int* __restrict__ hdr) {
    const int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i > 0 && i < B && us[i] == us[i - 1]) atomicOr(hdr + 1, 1);
}

}  // namespace

extern "C" size_t pda_triplet_plan_large_workspace_bytes(int B) { return B > 0 ? ws_layout(B).total : 0; }

extern "C" int pda_triplet_plan_large(const int32_t* users, const int32_t* pos, const int32_t* neg, int B, void* plan, void* workspace, void* stream) {
    if (!users || !pos || !neg || !plan || !workspace || B <= 0 || B > (1 << 24)) return PDA_ERR_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const WsL w = ws_layout(B);
    unsigned char* wb = reinterpret_cast<unsigned char*>(workspace);
    int* keys_in = reinterpret_cast<int*>(wb + w.keys_in);
    int* keys_out = reinterpret_cast<int*>(wb + w.keys_out);
    int* vals_in = reinterpret_cast<int*>(wb + w.vals_in);
    int* ukeys = reinterpret_cast<int*>(wb + w.ukeys);
    int* head = reinterpret_cast<int*>(wb + w.head);
    int* seg_of = reinterpret_cast<int*>(wb + w.seg_of);
    void* temp = wb + w.temp;
    size_t tb = w.temp_bytes;
    const PlanViewL v = plan_view_l(plan, B);
    const size_t n = 2 * (size_t)B;
    const unsigned grid = (unsigned)((n + 255) / 256);
    // header and "referenced once" bits start from zero (the rest of the plan is written in full where it is read)
    if (hipMemsetAsync(v.hdr, 0, 16, s) != hipSuccess) return PDA_ERR_LAUNCH;
    if (hipMemsetAsync(v.flags, 0, plan_bytes_l(B) - (size_t)(v.flags - reinterpret_cast<unsigned char*>(plan)), s) != hipSuccess) return PDA_ERR_LAUNCH;
    hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(256), 0, s, pos, neg, B, keys_in, vals_in);
    PDA_CHECK_LAUNCH();
    if (rocprim::radix_sort_pairs(temp, tb, (const int*)keys_in, keys_out, (const int*)vals_in, v.entries, n, 0, 32, s) != hipSuccess) return PDA_ERR_LAUNCH;
    hipLaunchKernelGGL(head_kernel, dim3(grid), dim3(256), 0, s, keys_out, (int)n, head);
    PDA_CHECK_LAUNCH();
    tb = w.temp_bytes;
    if (rocprim::inclusive_scan(temp, tb, (const int*)head, seg_of, n, rocprim::plus<int>(), s) != hipSuccess) return PDA_ERR_LAUNCH;
    hipLaunchKernelGGL(scatter_kernel, dim3(grid), dim3(256), 0, s, keys_out, head, seg_of, B, v);
    PDA_CHECK_LAUNCH();
    if (hipMemsetAsync(v.seg_start + 2 * B + 1, 0, 4, s) != hipSuccess) return PDA_ERR_LAUNCH;
    hipLaunchKernelGGL(xl_kernel, dim3(grid), dim3(256), 0, s, B, v);
    PDA_CHECK_LAUNCH();
    tb = w.temp_bytes;
    if (rocprim::radix_sort_keys(temp, tb, users, ukeys, (size_t)B, 0, 32, s) != hipSuccess) return PDA_ERR_LAUNCH;
    hipLaunchKernelGGL(user_dup_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, ukeys, B, v.hdr);
    PDA_CHECK_LAUNCH();
    return PDA_OK;
}
