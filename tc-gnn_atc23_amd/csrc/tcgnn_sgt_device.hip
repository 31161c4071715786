// tcgnn_sgt_device.hip - sparse-graph translation on the GPU.
//
// Finishes what the reference only sketches: preprocess_gpu (TCGNN_conv/TCGNN.cpp:229-256) calls
// fill_edgeToRow (TCGNN_kernel.cu:21-40, works) and fill_window (:42-80, an empty stub), so the
// reference never produces edgeToColumn / blockPartition on the device.  Here:
//   edgeToRow      one wavefront per CSR row (64 lanes, not 32)
//   edgeToColumn   rocPRIM segmented radix sort of the column ids per 16-row window (carrying the
//                  CSR position), head flags on the sorted keys, one global inclusive scan; the
//                  rank of a key inside its window is scan[p] - scan[window start]; scattered back
//   blockPartition ceil(unique / blockSize_w) per window, 1 for an edgeless window (the value the
//                  reference's host path yields, TCGNN.cpp:160), phantom window included when the
//                  caller's tensor has the slot (TCGNN.cpp:200), never written past bp_len.
// Output is bit-identical to tcgnn_preprocess (tests/test_gpu_parity.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring> // rocprim/iterator/texture_cache_iterator.hpp calls memset unqualified

#include <rocprim/rocprim.hpp>

#include "tcgnn.h"
#include "tcgnn_internal.h"

using namespace tcgnn;

namespace {

// (row pointers beyond the edge arrays - a malformed graph, reported after the one read-back at the end - are clamped: nothing is written out of bounds)
__global__ __launch_bounds__(256) void fill_edge_to_row_kernel(const int32_t* __restrict__ rowptr, int32_t N, int64_t cap, int32_t* e2r) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    int64_t lo = rowptr[row], hi = rowptr[row + 1];
    if (lo < 0) lo = 0;
    if (hi > cap) hi = cap;
    for (int64_t e = lo + (threadIdx.x & 63); e < hi; e += 64) e2r[e] = (int32_t)row;
}

// seg[w] = first CSR position of window w, for w = 0 .. nwin (seg[nwin] = E)
__global__ void window_offsets_kernel(const int32_t* __restrict__ rowptr, int32_t N, int32_t bh, int32_t nwin, int64_t cap, int32_t* seg, int32_t* ends) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > nwin) return;
    const int64_t r = w * bh;
    int64_t v = rowptr[r < N ? r : N];
    if (w == 0) ends[0] = (int32_t)v;          // nodePointer[0] and nodePointer[N] as they are: validated on the host behind the one read-back
    if (w == nwin) ends[1] = (int32_t)v;
    if (v < 0) v = 0;
    if (v > cap) v = cap;
    if (w > 0) {                                // (a non-monotone pointer array must not produce a negative segment either)
        int64_t p = rowptr[(r - bh) < N ? (r - bh) : N];
        if (p < 0) p = 0;
        if (p > cap) p = cap;
        if (v < p) v = p;
    }
    seg[w] = (int32_t)v;
}

__global__ void head_flags_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ pos, const int32_t* __restrict__ e2r,
                                  int32_t bh, int64_t cap, const int32_t* __restrict__ seg, int32_t nwin, int32_t* flags) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= cap) return;
    if (p >= seg[nwin]) { flags[p] = 0; return; }   // (beyond the CSR the row pointers describe: the scan runs over the whole array)
    int f = 1;
    if (p > 0) {
        const int wp = e2r[pos[p]] / bh, wq = e2r[pos[p - 1]] / bh;
        f = (wp != wq) || (keys[p] != keys[p - 1]);
    }
    flags[p] = f;
}

__global__ void scatter_rank_kernel(const int32_t* __restrict__ scan, const int32_t* __restrict__ pos, const int32_t* __restrict__ e2r,
                                    const int32_t* __restrict__ seg, int32_t bh, int64_t cap, int32_t nwin, int32_t* e2c) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= cap || p >= seg[nwin]) return;
    const int e = pos[p];
    const int w = e2r[e] / bh;
    e2c[e] = scan[p] - scan[seg[w]];
}

// bp for the windows the reference's loop visits (0 .. N/bh inclusive); counts[w] feeds the total
__global__ void block_partition_kernel(const int32_t* __restrict__ scan, const int32_t* __restrict__ seg, int32_t nwin, int64_t visited,
                                       int32_t bw, int32_t* bp, int64_t bp_len, int32_t* counts) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= visited) return;
    int uniq = 0;
    if (w < nwin) {
        const int s = seg[w], t = seg[w + 1];
        if (t > s) uniq = scan[t - 1] - scan[s] + 1;
    }
    const int eff = uniq ? uniq : 1;
    const int v = (eff + bw - 1) / bw;
    if (w < bp_len) bp[w] = v;
    counts[w] = v;
}

// largest column id of the CSR the row pointers describe (ids beyond num_nodes are legal, as in the host path): decides whether the
// key bits sorted - chosen from num_nodes without a read-back - were enough
__global__ __launch_bounds__(256) void max_id_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ seg, int32_t nwin, int64_t cap, uint32_t* out) {
    int64_t E = seg[nwin];
    if (E > cap) E = cap;
    uint32_t m = 0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < E; p += (int64_t)gridDim.x * blockDim.x) m = keys[p] > m ? keys[p] : m;
    for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)m, off, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

struct SgtResult { int32_t ends[2]; uint32_t max_id; uint32_t pad; int64_t total; };

// workspace layout (256-byte aligned parts): result words, seg[nwin + 1], counts[visited], keys / pos / flags / scan [E each], rocPRIM scratch
struct SgtLayout {
    size_t off_res, off_seg, off_counts, off_keys, off_pos, off_flags, off_scan, off_tmp, tmp_bytes, total;
};
inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }

int sgt_layout(int32_t num_nodes, int64_t num_edges, int32_t bh, SgtLayout* L) {
    const int32_t nwin = (int32_t)(((int64_t)num_nodes + bh - 1) / bh);
    const int64_t visited = (int64_t)num_nodes / bh + 1;
    const size_t E = (size_t)num_edges;
    size_t tb_sort = 0, tb_scan = 0, tb_red = 0;
    // (size queries only: nothing is launched with a null scratch pointer)
    rocprim::counting_iterator<int32_t> vin(0);
    if (E > 0) {
        if (rocprim::segmented_radix_sort_pairs(nullptr, tb_sort, (const uint32_t*)nullptr, (uint32_t*)nullptr, vin, (int32_t*)nullptr, (unsigned)E, (unsigned)nwin,
                                                (const int32_t*)nullptr, (const int32_t*)nullptr, 0u, 32u, (hipStream_t)0) != hipSuccess) return 1;
        if (rocprim::inclusive_scan(nullptr, tb_scan, (int32_t*)nullptr, (int32_t*)nullptr, E, rocprim::plus<int32_t>(), (hipStream_t)0) != hipSuccess) return 1;
    }
    if (rocprim::reduce(nullptr, tb_red, (const int32_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)visited, rocprim::plus<int64_t>(), (hipStream_t)0) != hipSuccess) return 1;
    size_t o = 0;
    L->off_res = o; o += up256(sizeof(SgtResult));
    L->off_seg = o; o += up256(((size_t)nwin + 1) * 4);
    L->off_counts = o; o += up256((size_t)visited * 4);
    L->off_keys = o; o += up256(E * 4);
    L->off_pos = o; o += up256(E * 4);
    L->off_flags = o; o += up256(E * 4);
    L->off_scan = o; o += up256(E * 4 + 4);
    L->tmp_bytes = up256(std::max(std::max(tb_sort, tb_scan), tb_red) + 256);
    L->off_tmp = o; o += L->tmp_bytes;
    L->total = o;
    return 0;
}

} // namespace

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? TCGNN_ERR_OOM : TCGNN_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" int tcgnn_preprocess_gpu_workspace_bytes(int32_t num_nodes, int64_t num_edges, int32_t blockSize_h, size_t* bytes) {
    if (!bytes || num_nodes < 0 || num_edges < 0 || blockSize_h <= 0) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_preprocess_gpu_workspace_bytes: null pointer or bad size");
    if (num_edges > 0x7fffffffLL) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_preprocess_gpu: int32 CSR positions only (E = %lld)", (long long)num_edges);
    SgtLayout L;
    if (sgt_layout(num_nodes, num_edges, blockSize_h, &L)) return fail(TCGNN_ERR_HIP, "tcgnn_preprocess_gpu_workspace_bytes: rocPRIM size query failed");
    *bytes = L.total;
    return TCGNN_OK;
}

// The translation on caller scratch: no allocation, no free, ONE stream synchronisation (the read-back of 24 bytes: the two row-pointer
// ends, the largest column id, TC_Blocks).  A graph whose ids need more key bits than num_nodes has (legal: the host path takes them too)
// is sorted a second time with all 32 - the only way to a second synchronisation.
extern "C" int tcgnn_preprocess_gpu_ws(const int32_t* d_edgeList, const int32_t* d_nodePointer, int32_t num_nodes, int64_t num_edges,
                                       int32_t blockSize_h, int32_t blockSize_w, int32_t* d_blockPartition, int64_t bp_len,
                                       int32_t* d_edgeToColumn, int32_t* d_edgeToRow, void* d_workspace, size_t workspace_bytes,
                                       int64_t* tc_blocks, void* stream_v) {
    if (!d_nodePointer || num_nodes < 0 || num_edges < 0 || blockSize_h <= 0 || blockSize_w <= 0 || bp_len < 0 ||
        (num_edges > 0 && (!d_edgeList || !d_edgeToColumn || !d_edgeToRow)) || (bp_len > 0 && !d_blockPartition))
        return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_preprocess_gpu: null array or bad size");
    if (num_edges > 0x7fffffffLL) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_preprocess_gpu: int32 CSR positions only (E = %lld)", (long long)num_edges);
    SgtLayout L;
    if (sgt_layout(num_nodes, num_edges, blockSize_h, &L)) return fail(TCGNN_ERR_HIP, "tcgnn_preprocess_gpu: rocPRIM size query failed");
    if (!d_workspace || workspace_bytes < L.total || (reinterpret_cast<uintptr_t>(d_workspace) & 255))
        return fail(TCGNN_ERR_WORKSPACE, "tcgnn_preprocess_gpu_ws: workspace needs %zu bytes 256-aligned (tcgnn_preprocess_gpu_workspace_bytes), got %zu at %p", L.total, workspace_bytes, d_workspace);
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    const int32_t nwin = (int32_t)(((int64_t)num_nodes + blockSize_h - 1) / blockSize_h);
    const int64_t visited = (int64_t)num_nodes / blockSize_h + 1; // TCGNN.cpp:200 loop bound
    char* const ws = static_cast<char*>(d_workspace);
    SgtResult* const d_res = reinterpret_cast<SgtResult*>(ws + L.off_res);
    int32_t* const seg = reinterpret_cast<int32_t*>(ws + L.off_seg);
    int32_t* const counts = reinterpret_cast<int32_t*>(ws + L.off_counts);
    uint32_t* const keys = reinterpret_cast<uint32_t*>(ws + L.off_keys);
    int32_t* const pos = reinterpret_cast<int32_t*>(ws + L.off_pos);
    int32_t* const flags = reinterpret_cast<int32_t*>(ws + L.off_flags);
    int32_t* const scan = reinterpret_cast<int32_t*>(ws + L.off_scan);
    void* const tmp = ws + L.off_tmp;
    const int64_t cap = num_edges;   // the edge arrays' length: the CSR the row pointers describe (<= cap) is what gets translated, TCGNN.cpp:196-197
    const uint32_t* kin = reinterpret_cast<const uint32_t*>(d_edgeList);

    unsigned end_bit = 1;            // key bits of ids below num_nodes; a larger id (seen in the read-back) takes the second pass
    while (end_bit < 32 && (1ull << end_bit) < (unsigned long long)std::max(num_nodes, 1)) ++end_bit;
    SgtResult res;
    for (int pass = 0; pass < 2; ++pass) {
        HIP_TRY(hipMemsetAsync(d_res, 0, sizeof(SgtResult), stream));
        hipLaunchKernelGGL(window_offsets_kernel, dim3((unsigned)(nwin / 256 + 1)), dim3(256), 0, stream, d_nodePointer, num_nodes, blockSize_h, nwin, cap, seg, d_res->ends);
        if (num_nodes > 0 && cap > 0)
            hipLaunchKernelGGL(fill_edge_to_row_kernel, dim3((unsigned)((num_nodes + 3) / 4)), dim3(256), 0, stream, d_nodePointer, num_nodes, cap, d_edgeToRow);
        HIP_TRY(hipGetLastError());
        if (cap > 0) {
            hipLaunchKernelGGL(max_id_kernel, dim3(1024), dim3(256), 0, stream, kin, seg, nwin, cap, &d_res->max_id);
            rocprim::counting_iterator<int32_t> vin(0);
            size_t tb = L.tmp_bytes;
            HIP_TRY(rocprim::segmented_radix_sort_pairs(tmp, tb, kin, keys, vin, pos, (unsigned)cap, (unsigned)nwin, seg, seg + 1, 0u, end_bit, stream));
            const unsigned eg = (unsigned)((cap + 255) / 256);
            hipLaunchKernelGGL(head_flags_kernel, dim3(eg), dim3(256), 0, stream, keys, pos, d_edgeToRow, blockSize_h, cap, seg, nwin, flags);
            tb = L.tmp_bytes;
            HIP_TRY(rocprim::inclusive_scan(tmp, tb, flags, scan, (size_t)cap, rocprim::plus<int32_t>(), stream));
            hipLaunchKernelGGL(scatter_rank_kernel, dim3(eg), dim3(256), 0, stream, scan, pos, d_edgeToRow, seg, blockSize_h, cap, nwin, d_edgeToColumn);
            HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(block_partition_kernel, dim3((unsigned)(visited / 256 + 1)), dim3(256), 0, stream, scan, seg, nwin, visited, blockSize_w, d_blockPartition, bp_len, counts);
        HIP_TRY(hipGetLastError());
        {
            size_t tb = L.tmp_bytes;
            HIP_TRY(rocprim::reduce(tmp, tb, counts, &d_res->total, (int64_t)0, (size_t)visited, rocprim::plus<int64_t>(), stream));
        }
        HIP_TRY(hipMemcpyAsync(&res, d_res, sizeof res, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (res.ends[0] != 0) return fail(TCGNN_ERR_BAD_GRAPH, "tcgnn_preprocess_gpu: nodePointer[0] = %d, expected 0", res.ends[0]);
        if (res.ends[1] < 0 || (int64_t)res.ends[1] > num_edges)
            return fail(TCGNN_ERR_BAD_GRAPH, "tcgnn_preprocess_gpu: nodePointer[num_nodes] = %d but edgeList holds %lld entries", res.ends[1], (long long)num_edges);
        if (end_bit >= 32 || res.max_id < (1ull << end_bit)) break;
        end_bit = 32;   // ids beyond num_nodes: every key bit
    }
    if (tc_blocks) *tc_blocks = res.total;
    return TCGNN_OK;
}

// The reference-shaped entry point (TCGNN.cpp:229-256 takes no scratch): allocates the workspace for the call.  Callers that translate more
// than once - or care about the ~2 GB hipMalloc / hipFree pair a Reddit-sized graph costs here - use tcgnn_preprocess_gpu_ws.
extern "C" int tcgnn_preprocess_gpu(const int32_t* d_edgeList, const int32_t* d_nodePointer, int32_t num_nodes, int64_t num_edges,
                                    int32_t blockSize_h, int32_t blockSize_w, int32_t* d_blockPartition, int64_t bp_len,
                                    int32_t* d_edgeToColumn, int32_t* d_edgeToRow, int64_t* tc_blocks, void* stream_v) {
    size_t bytes = 0;
    if (const int rc = tcgnn_preprocess_gpu_workspace_bytes(num_nodes, num_edges, blockSize_h > 0 ? blockSize_h : 1, &bytes)) return rc;
    void* ws = nullptr;
    {
        const hipError_t e = hipMalloc(&ws, bytes ? bytes : 1);
        if (e != hipSuccess) {
            (void)hipGetLastError();   // (the runtime's sticky error word would otherwise surface in the caller's next, unrelated launch)
            return fail(e == hipErrorOutOfMemory ? TCGNN_ERR_OOM : TCGNN_ERR_HIP, "tcgnn_preprocess_gpu: %zu bytes of scratch: %s (tcgnn_preprocess_gpu_ws takes caller scratch)", bytes, hipGetErrorString(e));
        }
    }
    const int rc = tcgnn_preprocess_gpu_ws(d_edgeList, d_nodePointer, num_nodes, num_edges, blockSize_h, blockSize_w, d_blockPartition, bp_len, d_edgeToColumn, d_edgeToRow,
                                           ws, bytes, tc_blocks, stream_v);
    (void)hipFree(ws);   // (the call above synchronised the stream)
    return rc;
}
