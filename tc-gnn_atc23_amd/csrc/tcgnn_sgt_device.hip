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

#include <cstdint>
#include <cstring> // rocprim/iterator/texture_cache_iterator.hpp calls memset unqualified

#include <rocprim/rocprim.hpp>

#include "tcgnn.h"
#include "tcgnn_internal.h"

using namespace tcgnn;

namespace {

__global__ __launch_bounds__(256) void fill_edge_to_row_kernel(const int32_t* __restrict__ rowptr, int32_t N, int32_t* e2r) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    for (int64_t e = rowptr[row] + (threadIdx.x & 63); e < rowptr[row + 1]; e += 64) e2r[e] = (int32_t)row;
}

// seg[w] = first CSR position of window w, for w = 0 .. nwin (seg[nwin] = E)
__global__ void window_offsets_kernel(const int32_t* __restrict__ rowptr, int32_t N, int32_t bh, int32_t nwin, int32_t* seg) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > nwin) return;
    const int64_t r = w * bh;
    seg[w] = rowptr[r < N ? r : N];
}

__global__ void head_flags_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ pos, const int32_t* __restrict__ e2r,
                                  int32_t bh, int64_t E, int32_t* flags) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E) return;
    int f = 1;
    if (p > 0) {
        const int wp = e2r[pos[p]] / bh, wq = e2r[pos[p - 1]] / bh;
        f = (wp != wq) || (keys[p] != keys[p - 1]);
    }
    flags[p] = f;
}

__global__ void scatter_rank_kernel(const int32_t* __restrict__ scan, const int32_t* __restrict__ pos, const int32_t* __restrict__ e2r,
                                    const int32_t* __restrict__ seg, int32_t bh, int64_t E, int32_t* e2c) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E) return;
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

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

} // namespace

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? TCGNN_ERR_OOM : TCGNN_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" int tcgnn_preprocess_gpu(const int32_t* d_edgeList, const int32_t* d_nodePointer, int32_t num_nodes, int64_t num_edges,
                                    int32_t blockSize_h, int32_t blockSize_w, int32_t* d_blockPartition, int64_t bp_len,
                                    int32_t* d_edgeToColumn, int32_t* d_edgeToRow, int64_t* tc_blocks, void* stream_v) {
    if (!d_nodePointer || num_nodes < 0 || num_edges < 0 || blockSize_h <= 0 || blockSize_w <= 0 || bp_len < 0 ||
        (num_edges > 0 && (!d_edgeList || !d_edgeToColumn || !d_edgeToRow)) || (bp_len > 0 && !d_blockPartition))
        return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_preprocess_gpu: null array or bad size");
    if (num_edges > 0x7fffffffLL) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_preprocess_gpu: int32 CSR positions only (E = %lld)", (long long)num_edges);
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    const int32_t nwin = (int32_t)(((int64_t)num_nodes + blockSize_h - 1) / blockSize_h);
    const int64_t visited = (int64_t)num_nodes / blockSize_h + 1; // TCGNN.cpp:200 loop bound

    DevBuf seg, keys, pos, flags, scan, counts, total, tmp, maxid;
    HIP_TRY(seg.alloc(((size_t)nwin + 1) * 4));
    HIP_TRY(counts.alloc((size_t)visited * 4));
    HIP_TRY(total.alloc(8));
    HIP_TRY(maxid.alloc(4));
    hipLaunchKernelGGL(window_offsets_kernel, dim3((unsigned)(nwin / 256 + 1)), dim3(256), 0, stream, d_nodePointer, num_nodes, blockSize_h, nwin, seg.as<int32_t>());
    HIP_TRY(hipGetLastError());
    // The CSR the row pointers describe, not the caller's array length, is what gets translated (the host path does the
    // same, TCGNN.cpp:196-197): an edgeList longer than nodePointer[num_nodes] (main_tcgnn.py:45-46 sizes the edge arrays
    // by the RAW edge count) leaves the sort outputs beyond it unwritten, and every later kernel would index with them.
    int32_t ends[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(&ends[0], seg.as<int32_t>(), 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(&ends[1], seg.as<int32_t>() + nwin, 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (ends[0] != 0) return fail(TCGNN_ERR_BAD_GRAPH, "tcgnn_preprocess_gpu: nodePointer[0] = %d, expected 0", ends[0]);
    if (ends[1] < 0 || (int64_t)ends[1] > num_edges)
        return fail(TCGNN_ERR_BAD_GRAPH, "tcgnn_preprocess_gpu: nodePointer[num_nodes] = %d but edgeList holds %lld entries", ends[1], (long long)num_edges);
    const int64_t E = ends[1];
    HIP_TRY(keys.alloc((size_t)E * 4));
    HIP_TRY(pos.alloc((size_t)E * 4));
    HIP_TRY(flags.alloc((size_t)E * 4));
    HIP_TRY(scan.alloc((size_t)E * 4));

    if (num_nodes > 0 && E > 0)
        hipLaunchKernelGGL(fill_edge_to_row_kernel, dim3((unsigned)((num_nodes + 3) / 4)), dim3(256), 0, stream, d_nodePointer, num_nodes, d_edgeToRow);
    HIP_TRY(hipGetLastError());

    if (E > 0) {
        const uint32_t* kin = reinterpret_cast<const uint32_t*>(d_edgeList);
        // sort as many key bits as the largest id has (ids >= num_nodes are legal here, as in the host path)
        uint32_t host_max = 0;
        {
            size_t tb = 0;
            DevBuf rtmp;
            HIP_TRY(rocprim::reduce(nullptr, tb, kin, maxid.as<uint32_t>(), 0u, (size_t)E, rocprim::maximum<uint32_t>(), stream));
            HIP_TRY(rtmp.alloc(tb));
            HIP_TRY(rocprim::reduce(rtmp.p, tb, kin, maxid.as<uint32_t>(), 0u, (size_t)E, rocprim::maximum<uint32_t>(), stream));
            HIP_TRY(hipMemcpyAsync(&host_max, maxid.p, 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
        unsigned end_bit = 1;
        while (end_bit < 32 && (1ull << end_bit) <= (unsigned long long)host_max) ++end_bit;
        rocprim::counting_iterator<int32_t> vin(0);
        size_t tb_sort = 0, tb_scan = 0;
        HIP_TRY(rocprim::segmented_radix_sort_pairs(nullptr, tb_sort, kin, keys.as<uint32_t>(), vin, pos.as<int32_t>(), (unsigned)E, (unsigned)nwin,
                                                    seg.as<int32_t>(), seg.as<int32_t>() + 1, 0u, end_bit, stream));
        HIP_TRY(rocprim::inclusive_scan(nullptr, tb_scan, flags.as<int32_t>(), scan.as<int32_t>(), (size_t)E, rocprim::plus<int32_t>(), stream));
        HIP_TRY(tmp.alloc(tb_sort > tb_scan ? tb_sort : tb_scan));
        HIP_TRY(rocprim::segmented_radix_sort_pairs(tmp.p, tb_sort, kin, keys.as<uint32_t>(), vin, pos.as<int32_t>(), (unsigned)E, (unsigned)nwin,
                                                    seg.as<int32_t>(), seg.as<int32_t>() + 1, 0u, end_bit, stream));
        const unsigned eg = (unsigned)((E + 255) / 256);
        hipLaunchKernelGGL(head_flags_kernel, dim3(eg), dim3(256), 0, stream, keys.as<uint32_t>(), pos.as<int32_t>(), d_edgeToRow, blockSize_h, E, flags.as<int32_t>());
        HIP_TRY(rocprim::inclusive_scan(tmp.p, tb_scan, flags.as<int32_t>(), scan.as<int32_t>(), (size_t)E, rocprim::plus<int32_t>(), stream));
        hipLaunchKernelGGL(scatter_rank_kernel, dim3(eg), dim3(256), 0, stream, scan.as<int32_t>(), pos.as<int32_t>(), d_edgeToRow, seg.as<int32_t>(), blockSize_h, E, d_edgeToColumn);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(block_partition_kernel, dim3((unsigned)(visited / 256 + 1)), dim3(256), 0, stream, scan.as<int32_t>(), seg.as<int32_t>(), nwin, visited,
                       blockSize_w, d_blockPartition, bp_len, counts.as<int32_t>());
    HIP_TRY(hipGetLastError());
    {
        size_t tb = 0;
        DevBuf rtmp;
        HIP_TRY(rocprim::reduce(nullptr, tb, counts.as<int32_t>(), total.as<int64_t>(), (int64_t)0, (size_t)visited, rocprim::plus<int64_t>(), stream));
        HIP_TRY(rtmp.alloc(tb));
        HIP_TRY(rocprim::reduce(rtmp.p, tb, counts.as<int32_t>(), total.as<int64_t>(), (int64_t)0, (size_t)visited, rocprim::plus<int64_t>(), stream));
        int64_t host_total = 0;
        HIP_TRY(hipMemcpyAsync(&host_total, total.p, 8, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream)); // also keeps the temporaries alive until the work is done
        if (tc_blocks) *tc_blocks = host_total;
    }
    return TCGNN_OK;
}
