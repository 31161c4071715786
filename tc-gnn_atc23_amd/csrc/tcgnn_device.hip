// tcgnn_device.hip - gfx950 (MI355X / CDNA4) kernels and C-ABI launchers of libtcgnn_hip.so.
//
// What replaces what (reference paths relative to /root/reference):
//   pack_kernel        legacy metadata -> packed tile stream.  The reference has no such step: its
//                      kernels rebuild every 16x8 tile by rescanning ALL edges of the window once
//                      per tile (TCGNN_kernel.cu:400-408, :656-663).  Packing once per graph
//                      removes that O(tiles x edges) work from every call.
//   absmax / convert   fp32 X -> fp16 staging copy, scaled by a per-call power of two so the
//                      10-bit-mantissa rounding equals the reference's TF32 rounding
//                      (wmma::__float_to_tf32, TCGNN_kernel.cu:438-444) without fp16's range limit.
//   spmm_kernel        TCGNN_kernel.cu:336-454 (binary A) and :459-578 (edge-valued A).
//   sddmm_kernel       TCGNN_kernel.cu:584-727.
//   spmm_blocked_kernel / spmm_lds_kernel (tcgnn_lds_spmm.inc) / spmm_small_kernel
//                      the same SpMM for big feature matrices (column-range-blocked gather), dense graphs (column ranges
//                      resident in LDS) and launch-latency-sized graphs (one launch, fp32 MFMA on fp32 X).
//   agnn_kernel        the SDDMM + edge-weighted SpMM pair of gnn_conv.py:115-158 in one gather, forward and backward.
//   *_csr_kernel       slow-but-correct HIP paths for CSRs whose rows are not strictly increasing
//                      (the packed edge-offset table assumes canonical rows).
//
// Data flow of one SpMM workgroup (one 16-row window, 1 or 4 wavefronts):
//   each wavefront walks its share of the window's 16x32 tiles; per tile it
//     (1) reads 32 source-row ids and a 16x32 adjacency bitmask (coalesced, 192 B),
//     (2) gathers the 32 fp16 feature rows straight into LDS with per-lane-addressed
//         global_load_lds_dwordx4 (no VGPR round trip, no ds_write), in an XOR-swizzled image,
//     (3) synthesises the MFMA A fragment from the bitmask in registers,
//     (4) reads B fragments with ds_read_b64_tr_b16 (hardware transpose: K runs over gathered
//         rows, which are the LDS rows) and issues one v_mfma_f32_16x16x32_f16 per 16 columns;
//   partial accumulators of the wavefronts are summed through LDS in a fixed order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <type_traits>
#include <vector>

#include "tcgnn.h"
#include "tcgnn_internal.h"

using namespace tcgnn;

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 fp16x4_raw;

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return fail(TCGNN_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct tcgnn_plan {
    int32_t N = 0, nw = 0, nw_eff = 0;   // N: rows of A (= rows of Y)
    double near_frac = 0;                // share of the condensed columns within num_cols / 16 rows of their window (locality_kernel)
    int32_t Nc = 0;                      // columns of A = rows of X (== N unless row-sharded)
    int32_t row_off = 0;                 // X row holding A's row 0 (row-sharded SDDMM)
    int64_t E = 0, tc_blocks = 0, total_wb = 0, max_wb = 0;   // max_wb: wide blocks of the longest window
    int canonical = 0, waves = 1;
    int32_t max_degree = 0;              // longest row of A (max_degree_kernel at creation): the range guard's thresholds follow it
    const int32_t *rowptr = nullptr, *col = nullptr, *bp = nullptr, *e2c = nullptr, *e2r = nullptr; // borrowed
    int64_t* d_wb_ptr = nullptr;  // [nw_eff + 1] first wide block of each window
    int32_t* d_order = nullptr;   // [nw_eff] window ids, heaviest first
    int32_t* d_cols = nullptr;    // [total_wb][32] source row of each condensed column (N = none)
    uint32_t* d_mask = nullptr;   // [total_wb][16] bit c of word r: A[r][c] != 0
    int32_t* d_ebase = nullptr;   // [total_wb][16] CSR position of the first edge of row r in the tile
    // column buckets for the range-blocked SpMM (0 = graph too sparse per window / too small)
    int32_t nbuckets = 0, bucket_rows = 0;
    uint32_t* d_bptr = nullptr;   // [nw_eff][nbuckets + 1] tile offset of the first tile whose first column is in bucket >= k
    int32_t num_cus = 256;
    size_t bytes = 0;
    // cell streams of the LDS-resident column-range SpMM (tcgnn_lds_spmm.inc), one per range length in use (lds_stream_of: 4 windows per
    // wavefront x 4 / 2 / 1 / 3 planes, 8 windows x 2 / 1 planes); nranges == 0: not built
    struct CellStream {   // published by build_lds_cells under its mutex; nranges is written last (release) and read first (acquire)
        std::atomic<int32_t> nranges{0};
        int32_t nwg = 0;
        int64_t tiles = 0;
        int32_t* d_order = nullptr;        // [nwg * 16 * maxw] window id of slot (workgroup, wavefront, window slot), -1 = none (lds_place_windows)
        // hot (workgroup, range) pairs: the ranges a workgroup streams into LDS.  Pair k of the compact cell table belongs to
        // workgroup wg for rbase[wg] <= k < rbase[wg + 1] and covers column range rlist[k].
        int32_t npairs = 0;
        int32_t* d_rbase = nullptr;        // [nwg + 1]
        int32_t* d_rlist = nullptr;        // [npairs + 4]
        // cold remainder: columns of the (workgroup, range) pairs too thin for a range fill, re-condensed per window in the gather
        // walks' packed format; run by spmm_kernel, ADDING into what the LDS-resident kernel stored
        int64_t cold_tiles = 0, hot_cols = 0, cold_cols = 0, cold_max = 0;   // cold_max: cold tiles of the longest window
        uint32_t* d_parts = nullptr;       // [slots] split windows: part | parts << 8 | LDS scratch index << 16 (nullptr: no window is split)
        int32_t nsplit = 0;                // windows shared by several wavefronts of their workgroup
        int64_t* d_cold_ptr = nullptr;     // [nw_eff + 1]
        int32_t* d_cold_cols = nullptr;    // [cold_tiles][32]
        uint32_t* d_cold_mask = nullptr;   // [cold_tiles][16]
        uint32_t* d_cell_ptr = nullptr;    // [nwg * nranges * 16 * maxw + 1] tile offset of cell (workgroup, range, wavefront, window slot)
        uint32_t* d_cell_tiles = nullptr;  // [tiles][32] 32 u16 row ids local to the range + 16 mask words
        // FLAT stream (tcgnn_lds_flat.inc): every cell of a hot pair has exactly flat_tpc tiles at a computed position - no cell
        // table, no ordinary tiles; what a cell holds beyond 32 flat_tpc columns sits in the cold remainder, which
        // spmm_cold_planar_kernel adds from the planar image.  0: an ordinary stream.
        int32_t flat_tpc = 0;
        uint32_t* d_flat = nullptr;        // [npairs][16 wavefronts][32 * maxw * flat_tpc words] metadata blocks
        int32_t* d_wcold_ptr = nullptr;    // [nwg * 16 + 1] the cold remainder as per-wavefront record lists, multiplied inside the flat kernel
        uint32_t* d_wcold = nullptr;       // [cold tiles][64 words] 32 column ids, 16 mask words, window slot (layouts with LDS to spare)
        // single-edge streams only (slot 6): CSR position of the edge in every K slot of the flat stream / of the cold tiles, -1: none
        int32_t* d_eidx = nullptr;         // [tiles][32], tile (pair k, wavefront v, entry x) = (k * 16 + v) * maxw * flat_tpc + x  (build time only)
        int32_t* d_cold_eidx = nullptr;    // [cold_tiles][32]                                                                          (build time only)
        uint16_t* d_eidx16 = nullptr;      // the same as offsets from the window's first CSR edge, 0xffff: none (what val_permute_kernel reads)
        uint16_t* d_cold_eidx16 = nullptr;
    };
    CellStream lds[7];   // (kLdsStreams; slot 6: the single-edge stream of the edge-valued LDS-resident SpMM, tcgnn_lds_val.inc)
    // single-edge tile stream (built with slot 6, on the first edge-valued call that would use it): every condensed column repeated
    // once per edge, so a K slot of a tile carries exactly ONE edge and a per-call value array can sit beside the slots
    int64_t* d_xwb_ptr = nullptr;   // [nw_eff + 1]
    int32_t* d_xcols = nullptr;     // [total_xwb][32]
    uint32_t* d_xmask = nullptr;    // [total_xwb][16] one bit per column
    int32_t* d_xeidx = nullptr;     // [total_xwb][32] CSR position of the slot's edge, -1: none
    int64_t total_xwb = 0;
    std::atomic<int8_t> val_choice{-1};   // -1 not tried, 0 the single-edge stream is of no use here (gather walks), 1 built
    mutable std::atomic<int8_t> lds_choice[65];   // automatic mode, per padded width / 16: -1 not decided yet, 0 gather walks, 1 LDS-resident kernel
    tcgnn_plan() { for (auto& c : lds_choice) c.store(-1, std::memory_order_relaxed); }
    // optional kernel timing (tcgnn_plan_set_timing): event pairs around the main kernel launches; slots are claimed atomically
    // (two streams may call into one plan)
    mutable std::vector<hipEvent_t> ev;
    mutable std::atomic<int> ev_used{0};
    mutable std::atomic<const char*> last_kernel{""};   // name of the main kernel the last call launched (tcgnn_plan_last_kernel)
    std::vector<int32_t> h_bp;       // blockPartition on the host: window weights for the placement of the LDS-resident walks
    mutable std::atomic<int> lds_extra[2] = {{-1}, {-1}};   // window slots the split hub windows add (4 / 8 windows per wavefront); -1: not computed
};

// Brackets the dominant kernel (spmm / sddmm proper, not the staging pass) with HIP events on the
// stream it is launched on, when timing is enabled and a pair is free.
struct KernelTimer {
    const tcgnn_plan* p; hipStream_t s; int slot = -1;
    KernelTimer(const tcgnn_plan* plan, hipStream_t stream, const char* kernel_name = nullptr) : p(plan), s(stream) {
        if (p && kernel_name) p->last_kernel.store(kernel_name, std::memory_order_relaxed);
        if (p && !p->ev.empty()) {
            const int k = p->ev_used.fetch_add(1, std::memory_order_relaxed);
            if (2 * k + 1 < (int)p->ev.size()) { slot = k; (void)hipEventRecord(p->ev[2 * slot], s); }
            else p->ev_used.store((int)p->ev.size() / 2, std::memory_order_relaxed);   // full: stay saturated, never wrap
        }
    }
    void stop() { if (slot >= 0) { (void)hipEventRecord(p->ev[2 * slot + 1], s); slot = -1; } }   // (before the range guard's fallback launch: not part of the kernel)
    ~KernelTimer() { stop(); }
};

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------

// Power-of-two exponent k such that absmax * 2^k lies in [2^14, 2^15): fp16-safe (max 65504) with
// 29 binades of normal range below the largest element.  0 for all-zero / non-finite data.
__device__ __forceinline__ int scale_exp_from_bits(uint32_t b) {
    if (b == 0u || b >= 0x7f800000u) return 0;
    int e = (int)(b >> 23) - 127;
    int k = 14 - e;
    return k > 126 ? 126 : (k < -126 ? -126 : k);
}
__device__ __forceinline__ float relu_if(int on, float v) { return on ? fmaxf(v, 0.0f) : v; }
__device__ __forceinline__ float pow2f(int k) { return __uint_as_float((uint32_t)(127 + k) << 23); }

// ---- the within-matrix range guard -------------------------------------------------------
// The fp16 operand image carries ONE power-of-two scale per matrix: an element more than 2^28 below the largest loses mantissa
// bits (fp16 subnormal) and one more than 2^39 below it is flushed to zero, where the reference's TF32 keeps fp32's exponent
// (TCGNN_kernel.cu:438-444).  The absolute error that leaves per element is <= max|X| * 2^-39, which stays far inside the
// contract's |d| <= 1e-3 max(1, |ref|) whenever max|X| < 2^8 - whatever the small elements are.  A matrix that holds BOTH an
// element >= 2^8 and a nonzero one more than 2^28 below the maximum is "wide": for it every fp16-path kernel returns at once and
// a fallback kernel launched behind it (plain fp32, operands rounded exactly like the reference's, CSR order) does the work; for
// every other matrix the fallback returns at once.  The decision is made on the device from the staging pass's header words -
// word k: bits of max |.| (k = 0: X, 1: edge values), word k + 2: 0x7f800000 - bits of the smallest nonzero |.| (0: none seen) -
// so no call synchronises or reads anything back.
// How large is "large": what an element below max 2^-28 loses is at most max 2^-39 in absolute terms, and a result may collect at
// most k such errors - k = min(cap, n_tiny), cap = the longest row of the graph (SpMM) or 2 D (SDDMM, fused AGNN: the terms of one
// dot product), n_tiny = how many elements of the matrix lose bits at all (counted by the conversion pass: ONE stray 1e-5 in a
// matrix of 1e4's - what a training epoch's activations look like, tools/probe_training_ranges.py - costs one error, not cap).
// The sum must stay inside the contract's 1e-3 max(1, |ref|) whatever the result is, i.e. below 2^-10:
//   binary SpMM        k max 2^-39 <= 2^-10                 -> wide iff     log2 max  >= 29 - log2 k
//   SDDMM / fused AGNN k max^2 2^-39 <= 2^-10               -> wide iff 2 * log2 max  >= 29 - log2 k
//   edge-valued SpMM   2 k max|A| max|X| 2^-39 <= 2^-10     -> wide iff log2 max|A| + log2 max|X| >= 28 - log2 k
// Header words (written by the staging pass): 4 = cap for X (0: guard off / an image the caller staged: never wide), 5 = cap for
// the edge values, 6 = n_tiny of X, 7 = the power of max in the bound (1 or 2).
__device__ __forceinline__ bool range_spread(const uint32_t* hdr, int k, int& emax) {
    const uint32_t mx = hdr[k], mi = hdr[k + 2];
    emax = (int)(mx >> 23);
    if (mi == 0u || mx == 0u || mx >= 0x7f800000u) return false;
    return emax - (int)((0x7f800000u - mi) >> 23) > 28;
}
__device__ __forceinline__ int ceil_log2_u32(uint32_t k) { return k <= 1u ? 0 : 32 - __clz((int)(k - 1u)); }
__device__ __forceinline__ bool range_is_wide(const uint32_t* hdr, int) {   // the feature matrix alone (binary SpMM, SDDMM, fused AGNN)
    int emax;
    const uint32_t cap = hdr[4], n = hdr[6];
    if (!range_spread(hdr, 0, emax) || cap == 0u || n == 0u) return false;
    return (int)hdr[7] * (emax - 127) >= 29 - ceil_log2_u32(n < cap ? n : cap);
}
__device__ __forceinline__ bool range_is_wide_val(const uint32_t* hdr) {
    int ex, ea;
    const bool sx = range_spread(hdr, 0, ex), sa = range_spread(hdr, 1, ea);
    const uint32_t cap = hdr[5];
    if (!(sx || sa) || cap == 0u || hdr[0] == 0u || hdr[1] == 0u) return false;
    const uint32_t n = hdr[6], k = (sa || n >= cap) ? cap : (n ? n : 1u);   // (edge values that lose bits are not counted: the longest row bounds them)
    return (ex - 127) + (ea - 127) >= 28 - ceil_log2_u32(k);
}
// ---- SDDMM and the fused AGNN pair at guard level 2 (r04): two ways through a "wide" matrix.  What a training epoch produces is ONE
// or a few lost elements (tools/probe_training_ranges.py), i.e. a handful of DIRTY ROWS of X - recorded by the conversion pass in
// header word 8 (count) and words 16 .. 63 (row numbers).  With at most kSparseRows of them the MFMA kernels run as usual and
// wide_patch_kernel recomputes, in fp32 with the reference's operand rounding, exactly the edges that touch a dirty row (their
// scores, and what those edges contribute to the aggregate and to d_w): a launch that returns at once when nothing is wide and
// costs ~0.1 ms when something is.  More dirty rows than that: the MFMA kernels return and the same launch does all the work in
// plain fp32 (wide_dense_body).  The aggregation operators, whose bound is linear and never reached in training, keep their own
// fallback kernels (spmm_wide_fallback_kernel, the fp32-MFMA walk).
static constexpr uint32_t kSparseRows = 48;
__device__ __forceinline__ bool wide2_sparse(const uint32_t* hdr) { return range_is_wide(hdr, 0) && hdr[8] <= kSparseRows; }
__device__ __forceinline__ bool wide2_dense(const uint32_t* hdr) { return range_is_wide(hdr, 0) && hdr[8] > kSparseRows; }
// conversion pass: a thread that met an element losing bits records the row (duplicates are possible: the patch de-duplicates)
__device__ __forceinline__ void note_dirty_row(uint32_t* hdr, uint32_t nt, int64_t row) {
    if (!hdr || !nt) return;
    const uint32_t k = atomicAdd(hdr + 8, 1u);
    if (k < kSparseRows) hdr[16 + k] = (uint32_t)row;
}
// conversion pass: this thread's count of elements that lose bits (nonzero, below fp16's normal range once scaled) -> hdr[6]
__device__ __forceinline__ void count_tiny(uint32_t* cnt, uint32_t mine) {   // mine <= 15; lanes that left the kernel early count as 0
    if (!cnt) return;
    uint32_t total = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) total += (uint32_t)__popcll(__ballot((mine >> b) & 1u)) << b;
    if (total && (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(cnt, total);
}
// (decided on the SOURCE element and its scaled fp32 value, not on the converted half: an element more than 2^39 below the maximum
//  converts to exactly 0 and must still count - ADVICE r03: one 1e12 among O(1) data had n_tiny = 0 and stayed on the MFMA path)
__device__ __forceinline__ uint32_t is_tiny(float x, float s) { return (x != 0.0f && fabsf(x * s) < 6.103515625e-5f) ? 1u : 0u; }

// Round to a 10-bit mantissa, nearest with ties AWAY from zero: bit-for-bit what the reference's
// wmma::__float_to_tf32 (cvt.rna.tf32.f32, TCGNN_kernel.cu:441-444) does.  The result has at most
// 11 significant bits, so the following fp32 -> fp16 conversion is exact for every element within
// 2^-28 of the (scaled) maximum; fp16's default nearest-EVEN would differ on ties (about one
// element in 2^13), which shows up as 2^-10-sized output differences.
__device__ __forceinline__ float round_rna10(float x) {   // the same rounding kept in fp32 (fallback kernels)
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7f800000u) != 0x7f800000u) u = (u + 0x1000u) & 0xffffe000u;
    return __uint_as_float(u);
}
__device__ __forceinline__ _Float16 to_half_rna(float x) {
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7f800000u) != 0x7f800000u) u = (u + 0x1000u) & 0xffffe000u;
    return (_Float16)__uint_as_float(u);
}

// LDS image of one tile: 32 gathered rows x (NT*16) halves, addressed in 16-byte slots.
// global_load_lds lands lane l of an instruction at slot (instr*64 + l), so the image is any
// bijection (row, slot-in-row) <-> slot we like, applied on the SOURCE address.  The bijection is
// chosen so that every 32-lane pass of ds_read_b64_tr_b16 (8 rows x 32 B) touches all 64 banks once.
template <int NT>
struct TileImage {
    static constexpr int C = 2 * NT; // 16-byte slots per row
    __device__ static __forceinline__ int slot(int row, int c) {
        if constexpr (NT == 1) {
            int rp = (row & ~0xC) | ((row & 4) << 1) | ((row & 8) >> 1); // swap row bits 2 and 3
            return rp * 2 + c;
        } else if constexpr (NT == 2) {
            return row * 4 + (c ^ (((row >> 3) & 1) << 1));
        } else if constexpr (NT == 4) {
            return row * 8 + (c ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2)));
        } else if constexpr (NT == 8) {
            return row * 16 + (c ^ (((row & 3) | (((row >> 3) & 1) << 2)) << 1));
        } else {
            return row * C + c;
        }
    }
    __device__ static __forceinline__ void unslot(int q, int& row, int& c) {
        if constexpr (NT == 1) {
            int rp = q >> 1;
            row = (rp & ~0xC) | ((rp & 4) << 1) | ((rp & 8) >> 1);
            c = q & 1;
        } else if constexpr (NT == 2) {
            row = q >> 2;
            c = (q & 3) ^ (((row >> 3) & 1) << 1);
        } else if constexpr (NT == 4) {
            row = q >> 3;
            c = (q & 7) ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2));
        } else if constexpr (NT == 8) {
            row = q >> 4;
            c = (q & 15) ^ (((row & 3) | (((row >> 3) & 1) << 2)) << 1);
        } else {
            row = q / C;
            c = q - row * C;
        }
    }
};

__device__ __forceinline__ half4 lds_read_tr16(const char* p) {
    fp16x4_raw v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4_raw*)(p));
    return __builtin_bit_cast(half4, v);
}

// ------------------------------------------------------------------------------------------
// pack: legacy (nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow) -> tile stream
// ------------------------------------------------------------------------------------------
// Locality of the numbering: how many condensed columns lie within `reach` rows of their own window.  A uniform random graph gives
// 2 reach / num_cols (1/8 at reach = num_cols / 16), a graph whose communities are numbered consecutively nearly all of them.
// Decides between the per-window walk in XCD-contiguous order (co-resident workgroups share their gathered rows in L2) and the
// range-blocked walk (which picks its windows strided over the whole graph).
// longest row of the CSR (grid-stride; one atomic per workgroup)
__global__ __launch_bounds__(256) void max_degree_kernel(const int32_t* __restrict__ rowptr, int32_t N, uint32_t* out) {
    uint32_t m = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (int64_t)gridDim.x * blockDim.x) {
        const int32_t d = rowptr[r + 1] - rowptr[r];
        m = max(m, d > 0 ? (uint32_t)d : 0u);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    __shared__ uint32_t wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) { m = max(max(wm[0], wm[1]), max(wm[2], wm[3])); if (m) atomicMax(out, m); }
}

__global__ __launch_bounds__(256) void locality_kernel(const int64_t* __restrict__ wb_ptr, const int32_t* __restrict__ cols, int32_t nw, int32_t Nc,
                                                       int32_t row_off, int32_t reach, unsigned long long* __restrict__ out) {
    const int w = blockIdx.x;
    if (w >= nw) return;
    const int64_t tb = wb_ptr[w] * kWbCols, n = (wb_ptr[w + 1] - wb_ptr[w]) * kWbCols;
    const int64_t centre = (int64_t)row_off + (int64_t)w * kWinRows + kWinRows / 2;
    unsigned near = 0, all = 0;
    for (int64_t q = threadIdx.x; q < n; q += blockDim.x) {
        const int32_t c = cols[tb + q];
        if (c >= Nc) continue;
        ++all;
        const int64_t d = (int64_t)c - centre;
        near += (d < 0 ? -d : d) <= reach;
    }
    for (int o = 32; o > 0; o >>= 1) { near += __shfl_down(near, o); all += __shfl_down(all, o); }
    if ((threadIdx.x & 63) == 0 && all) { atomicAdd(&out[0], (unsigned long long)near); atomicAdd(&out[1], (unsigned long long)all); }
}

__global__ __launch_bounds__(256) void pack_kernel(const int32_t* __restrict__ rowptr,
                                                   const int32_t* __restrict__ col,
                                                   const int32_t* __restrict__ e2c,
                                                   const int32_t* __restrict__ e2r,
                                                   const int64_t* __restrict__ wb_ptr, int32_t N,
                                                   int32_t Nc, int32_t* cols, uint32_t* mask,
                                                   int32_t* ebase, int32_t* flags) {
    const int w = blockIdx.x;
    const int64_t n0 = (int64_t)w * kWinRows;
    const int64_t n1 = n0 + kWinRows < N ? n0 + kWinRows : N;
    const int64_t base = wb_ptr[w];
    const int64_t nwb = wb_ptr[w + 1] - base;
    for (int64_t k = threadIdx.x; k < nwb * kWbCols; k += blockDim.x) cols[base * kWbCols + k] = Nc; // zero sentinel row
    for (int64_t k = threadIdx.x; k < nwb * kWinRows; k += blockDim.x) {
        mask[base * kWinRows + k] = 0u;
        ebase[base * kWinRows + k] = 0;
    }
    __syncthreads();
    if (n0 >= N) return;
    const int64_t e0 = rowptr[n0], e1 = rowptr[n1];
    for (int64_t e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
        const int c = e2c[e];
        const int r = e2r[e] - (int)n0;
        const int v = col[e];
        if (c < 0 || (int64_t)c >= nwb * kWbCols || r < 0 || r >= kWinRows || v < 0 || v >= Nc) {
            flags[0] = 1;
            continue;
        }
        const int64_t tile = base + (c >> 5);
        cols[tile * kWbCols + (c & 31)] = v; // duplicates of a column write the same id
        atomicOr(&mask[tile * kWinRows + r], 1u << (c & 31));
        bool first_in_tile_row = true;
        if (e > e0 && e2r[e - 1] - (int)n0 == r) {
            const int cp = e2c[e - 1];
            if (cp >= c) flags[1] = 1; // row not strictly increasing: edge-offset table unusable
            first_in_tile_row = (cp >> 5) != (c >> 5);
        }
        if (first_in_tile_row) ebase[tile * kWinRows + r] = (int32_t)e;
    }
}

// ------------------------------------------------------------------------------------------
// staging: absmax + fp32 -> scaled fp16 copy with a zero sentinel row
// ------------------------------------------------------------------------------------------
// (out_lo: where the smallest nonzero magnitude is recorded for the range guard, nullptr: not wanted - images a caller stages
//  itself, tcgnn_stage_absmax, carry no such word and are never "wide")
// Both absmax kernels: kAbsmaxThreads threads per workgroup, four independent 16-byte loads in flight per thread (r03: one load per
// trip of a grid-stride loop left 8 KB in flight per CU - 25 us for the 60 MB of a Reddit-shaped X, 2.4 TB/s), one atomic per
// workgroup and word.
constexpr int kAbsmaxThreads = 1024;
// one workgroup per CU at most, and none without sixteen 16-byte loads per thread to do
static inline int absmax_grid(int64_t n) { return (int)std::min<int64_t>(256, n / ((int64_t)kAbsmaxThreads * 64) + 1); }
template <bool GATED>
__device__ __forceinline__ void absmax_body(const float* __restrict__ p, const float* __restrict__ gate, int64_t n, uint32_t* out, uint32_t* out_lo,
                                            uint32_t guard_cap, uint32_t guard_pow) {
    uint32_t m = 0, lo = 0;   // lo: 0x7f800000 - bits of the smallest nonzero finite magnitude (larger = smaller; 0 = none): range_is_wide
    auto see = [&](float f, float gt) {
        const uint32_t b = (!GATED || gt > 0.0f) ? __float_as_uint(f) & 0x7fffffffu : 0u;
        m = max(m, b);
        lo = max(lo, (b - 1u < 0x7f7fffffu) ? 0x7f800000u - b : 0u);   // (b - 1 wraps for 0: zero, Inf and NaN do not count)
    };
    auto see4 = [&](const float4& v, const float4& gt) { see(v.x, gt.x); see(v.y, gt.y); see(v.z, gt.z); see(v.w, gt.w); };
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    const uintptr_t both = reinterpret_cast<uintptr_t>(p) | (GATED ? reinterpret_cast<uintptr_t>(gate) : 0);
    if ((both & 15) == 0) {   // (scalar loads: 85 us for 2 x 60 MB)
        const int64_t n4 = n >> 2;
        const float4* p4 = reinterpret_cast<const float4*>(p);
        const float4* g4 = reinterpret_cast<const float4*>(gate);
        const float4 one = {1.f, 1.f, 1.f, 1.f};
        int64_t k = gid;
        for (; k + 3 * gsz < n4; k += 4 * gsz) {
            const float4 v0 = p4[k], v1 = p4[k + gsz], v2 = p4[k + 2 * gsz], v3 = p4[k + 3 * gsz];
            if constexpr (GATED) {
                const float4 t0 = g4[k], t1 = g4[k + gsz], t2 = g4[k + 2 * gsz], t3 = g4[k + 3 * gsz];
                see4(v0, t0); see4(v1, t1); see4(v2, t2); see4(v3, t3);
            } else { see4(v0, one); see4(v1, one); see4(v2, one); see4(v3, one); }
        }
        for (; k < n4; k += gsz) see4(p4[k], GATED ? g4[k] : one);
        for (int64_t t = (n4 << 2) + gid; t < n; t += gsz) see(p[t], GATED ? gate[t] : 1.f);
    } else {
        for (int64_t k = gid; k < n; k += gsz) see(p[k], GATED ? gate[k] : 1.f);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { m = max(m, (uint32_t)__shfl_xor((int)m, off)); lo = max(lo, (uint32_t)__shfl_xor((int)lo, off)); }
    // one atomic per workgroup: a single word saturates near 88 atomics/us (MI355X_MICROARCH.md
    // "dequeue"), so per-wave atomics from a 2048-block grid alone cost ~90 us
    __shared__ uint32_t wmax[kAbsmaxThreads / 64], wlo[kAbsmaxThreads / 64];
    if ((threadIdx.x & 63) == 0) { wmax[threadIdx.x >> 6] = m; wlo[threadIdx.x >> 6] = lo; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nwv = (int)(blockDim.x >> 6);
        m = 0; lo = 0;
        for (int w = 0; w < nwv; ++w) { m = max(m, wmax[w]); lo = max(lo, wlo[w]); }
        if (m) atomicMax(out, m);
        if (lo && out_lo) atomicMax(out_lo, lo);
        if (out_lo && blockIdx.x == 0) { out_lo[2] = guard_cap; if (guard_pow) out_lo[5] = guard_pow; }   // (words k + 4 and, for X, 7: range_is_wide)
    }
}
__global__ __launch_bounds__(kAbsmaxThreads) void absmax_kernel(const float* __restrict__ p, int64_t n,
                                                                uint32_t* out, uint32_t* out_lo, uint32_t guard_cap, uint32_t guard_pow) {
    absmax_body<false>(p, nullptr, n, out, out_lo, guard_cap, guard_pow);
}
// absmax over the elements a gate lets through (gate > 0): the ReLU backward mask applied while staging dY
__global__ __launch_bounds__(kAbsmaxThreads) void absmax_gated_kernel(const float* __restrict__ p, const float* __restrict__ gate, int64_t n, uint32_t* out, uint32_t* out_lo, uint32_t guard_cap, uint32_t guard_pow) {
    absmax_body<true>(p, gate, n, out, out_lo, guard_cap, guard_pow);
}

// One thread per 16-byte output chunk (8 halves).  Rows: N real + 1 all-zero sentinel row that
// padding columns of the tile stream point at (the reference zero-fills those, :423-424).
template <bool VEC>
__global__ __launch_bounds__(256) void convert_kernel(const float* __restrict__ X, int32_t N,
                                                      int32_t D, int32_t Dpad, int32_t pitch,
                                                      _Float16* __restrict__ X16,
                                                      const uint32_t* __restrict__ hdr, const float* __restrict__ G = nullptr, int64_t ldx = 0,
                                                      uint32_t* __restrict__ tiny = nullptr, uint32_t* __restrict__ dirty_hdr = nullptr) {
    if (ldx == 0) ldx = D;   // row stride of X (and G) in floats: > D when X is a column block of a wider matrix
    const int cpr = Dpad >> 3;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = ((int64_t)N + 1) * cpr;
    if (q >= total) return;
    const int64_t row = q / cpr;
    const int d0 = (int)(q - row * cpr) * 8;
    const float s = pow2f(scale_exp_from_bits(hdr[0]));
    half8 o;
    uint32_t nt = 0;   // elements that lose bits in the image (range guard): nonzero, below fp16's normal range once scaled
    if (row < N && VEC && d0 + 8 <= D) {
        const float4* src = reinterpret_cast<const float4*>(X + row * ldx + d0);
        const float4 a = src[0], b = src[1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool on = !G || G[row * ldx + d0 + j] > 0.0f;
            o[j] = on ? to_half_rna(v[j] * s) : (_Float16)0.0f;
            nt += on ? is_tiny(v[j], s) : 0u;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = d0 + j;
            const bool on = row < N && d < D && (!G || G[row * ldx + d] > 0.0f);
            const float v = on ? X[row * ldx + d] : 0.0f;
            o[j] = to_half_rna(v * s);
            nt += is_tiny(v, s);
        }
    }
    *reinterpret_cast<half8*>(X16 + row * pitch + d0) = o;
    note_dirty_row(dirty_hdr, nt, row);
    count_tiny(tiny, nt);
}

// ------------------------------------------------------------------------------------------
// SpMM:  Y[window] = A_tile-stream * X16
// ------------------------------------------------------------------------------------------
struct SpmmArgs {
    const int64_t* wb_ptr;
    const int32_t* order;
    const int32_t* cols;
    const uint32_t* mask;
    const int32_t* ebase;
    const _Float16* x16;
    const float* edge_val;
    const uint32_t* hdr;
    float* y;
    int32_t N, D, stride, chunk0;
    int64_t E;
    int32_t xrows;   // rows of X16 including the zero sentinel row
    int32_t relu;    // fused epilogue: Y = max(A X, 0) (binary SpMM only)
    int32_t ldy;     // row stride of Y in floats (== D unless this call is one column block of a wider matrix)
    int32_t big;     // the fp16 image is 4 GB or more: gathers use 64-bit lane addresses (a buffer descriptor's index * stride wraps at 2^32)
    const float* w;  // f3: dense update fused behind the aggregation, Y[N, dout] = (A X) W with W [D, dout] fp32 row-major (nullptr: Y = A X)
    int32_t dout;
    int32_t accumulate;   // Y += A X: the cold remainder of a plan whose dense part the LDS-resident kernel has already stored
};

// ---- f3: the dense update in the aggregation kernel's epilogue (gnn_conv.py:92-97: X' = TCGNN.forward(X); X' = mm(X', W)).
// A window's aggregated rows Acc[16][din] (fp32, true scale) sit in LDS scratch `acc_lds` (row-major, leading dimension ldp, odd:
// conflict-free column reads); one wavefront multiplies them by the W columns of output tile t on the fp32 matrix pipe
// (v_mfma_f32_16x16x4_f32: exact fp32 fma chain, the precision of the torch.mm it replaces) and returns C[row 4g+ii][col 16t+i].
// W (<= 128 x 128 floats) is read through the caches: every window re-reads the same 64 KB at most.
__device__ __forceinline__ floatx4 dense_update_tile(const float* acc_lds, int ldp, int din, const float* __restrict__ w, int k0, int dout,
                                                     int t, int g, int i) {
    floatx4 c = {0.f, 0.f, 0.f, 0.f};
    const int n = 16 * t + i;
    for (int kk = 0; kk < din; kk += 4) {
        const int k = kk + g;
        const float av = k < din ? acc_lds[i * ldp + k] : 0.0f;
        const float bv = (k < din && n < dout) ? w[(int64_t)(k0 + k) * dout + n] : 0.0f;
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c, 0, 0, 0);
    }
    return c;
}

// ---- memory pipeline discipline -----------------------------------------------------------------
// With an LDS-DMA in flight hipcc waits vmcnt(0) at the first use of ANY ordinary load result and
// before any LDS read it can see (cdna_hip_programming.md 5, trap (b)) - that drained the next
// tile's gather in the first version of these kernels.  Hiding loads in inline asm avoids the
// drain, but a VGPR with an asm load in flight is a trap of its own: the register allocator may
// copy it before our wait (found by tools/audit_hidden_loads.py: `v_mov_b32 v2, v3` scheduled
// above the s_waitcnt).  So the rule here is: NO VGPR EVER HAS A LOAD IN FLIGHT OUTSIDE ONE ASM
// STATEMENT.
//   * global -> LDS: LDS-DMA builtins only (per-tile metadata, gathered rows, edge values, SDDMM
//     operands).  No VGPR destination; in flight across loop iterations; retired by wait_vm0().
//   * LDS -> VGPR: the generated blocks of tcgnn_lds_blocks.inc, whose s_waitcnt lgkmcnt(0) is in
//     the same asm statement as the reads (early-clobber outputs).
typedef uint32_t uintx4 __attribute__((ext_vector_type(4)));
#include "tcgnn_lds_blocks.inc"

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// LDS store issued from asm (no destination register, so nothing can be in flight into a VGPR);
// LDS operations of one wavefront execute in order, a later block read sees the data.
// (mask & a) | (~mask & b) in one instruction (hipcc otherwise spends a compare and a select on the loop-invariant b)
__device__ __forceinline__ uint32_t bitfield_select(uint32_t mask, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void lds_write_b32(uint32_t lds_byte_addr, float v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(lds_byte_addr), "v"(v) : "memory");
}

template <int N> __device__ __forceinline__ void lds_ids_block(const uint32_t* ad, uint32_t* v, uint32_t qaddr, uintx4& q) {
    if constexpr (N == 1) lds_ids_block1(ad, v, qaddr, q);
    else if constexpr (N == 2) lds_ids_block2(ad, v, qaddr, q);
    else if constexpr (N == 3) lds_ids_block3(ad, v, qaddr, q);
    else if constexpr (N == 4) lds_ids_block4(ad, v, qaddr, q);
    else if constexpr (N == 5) lds_ids_block5(ad, v, qaddr, q);
    else if constexpr (N == 6) lds_ids_block6(ad, v, qaddr, q);
    else if constexpr (N == 7) lds_ids_block7(ad, v, qaddr, q);
    else if constexpr (N == 8) lds_ids_block8(ad, v, qaddr, q);
    else if constexpr (N == 9) lds_ids_block9(ad, v, qaddr, q);
    else lds_ids_block10(ad, v, qaddr, q);
}
template <int N, int OFF> __device__ __forceinline__ void lds_q_block(const uint32_t* ad, uintx4* q) {
    if constexpr (N == 1) lds_q_block1<OFF>(ad, q);
    else if constexpr (N == 2) lds_q_block2<OFF>(ad, q);
    else if constexpr (N == 3) lds_q_block3<OFF>(ad, q);
    else if constexpr (N == 4) lds_q_block4<OFF>(ad, q);
    else if constexpr (N == 5) lds_q_block5<OFF>(ad, q);
    else if constexpr (N == 6) lds_q_block6<OFF>(ad, q);
    else if constexpr (N == 7) lds_q_block7<OFF>(ad, q);
    else if constexpr (N == 8) lds_q_block8<OFF>(ad, q);
    else if constexpr (N == 9) lds_q_block9<OFF>(ad, q);
    else lds_q_block10<OFF>(ad, q);
}
template <int K, int OFF> __device__ __forceinline__ void lds_tr_block(const uint32_t (*ad)[2], half4* lo, half4* hi) {
    if constexpr (K == 1) lds_tr_block1<OFF>(ad, lo, hi);
    else if constexpr (K == 2) lds_tr_block2<OFF>(ad, lo, hi);
    else if constexpr (K == 3) lds_tr_block3<OFF>(ad, lo, hi);
    else if constexpr (K == 4) lds_tr_block4<OFF>(ad, lo, hi);
    else { lds_tr_block4<OFF>(ad, lo, hi); lds_tr_block<K - 4, OFF>(ad + 4, lo + 4, hi + 4); }
}

// 256-entry table: byte of adjacency bits -> the eight fp16 {0,1} values of a binary A fragment.
// One LDS read replaces ~28 VALU instructions per tile.
__device__ __forceinline__ void fill_afrag_table(char* tab) {
    for (int e = threadIdx.x; e < 256; e += blockDim.x) {
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ((e >> j) & 1) ? (_Float16)1.0f : (_Float16)0.0f;
        *reinterpret_cast<half8*>(tab + e * 16) = v;
    }
}

// Per-tile metadata travels as ONE 256-byte DMA: lane l < 32 fetches cols[t][l], lanes 32..47
// mask[t][l-32], lanes 48..63 ebase[t][l-48]; it lands lane-linear in a per-wavefront pad.
#ifndef TCGNN_NT_STORES
#define TCGNN_NT_STORES 0   // 1: the fused AGNN kernel's score / slice-addend stores are non-temporal (A/B experiments)
#endif
template <typename T> __device__ __forceinline__ void st_stream(T* p, T v) {
#if TCGNN_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
#ifndef TCGNN_META_AUX
#define TCGNN_META_AUX 0   // cache policy of the metadata stream (bit 1 = nt): A/B experiments
#endif
struct MetaSource {
    const char* base;   // this lane's element of tile 0
    int shift;          // log2(bytes per tile) of the array this lane reads
    __device__ __forceinline__ MetaSource(const int32_t* cols, const uint32_t* mask, const int32_t* ebase, int lane) {
        if (lane < 32) { base = reinterpret_cast<const char*>(cols + lane); shift = 7; }
        else if (lane < 48) { base = reinterpret_cast<const char*>(mask + (lane - 32)); shift = 6; }
        else { base = reinterpret_cast<const char*>(ebase + (lane - 48)); shift = 6; }
    }
    __device__ __forceinline__ void dma(int64_t t, uint32_t pad_lds) const {
        const char* src = base + (t << shift);
        __builtin_amdgcn_global_load_lds((GLB_AS const void*)src, (LDS_AS void*)(uintptr_t)pad_lds, 4, 0, TCGNN_META_AUX);
    }
};
static constexpr int kPadBytes = 256;

// eight 4-byte LDS reads at per-lane addresses, retired before anything else is issued (agnn_kernel backward: the saved scores of
// this lane's eight tile columns; TileWalker: its edge values - picked out of the lane's run by address instead of by a cascade of selects)
__device__ __forceinline__ void lds_read8_b32(const uint32_t (&ad)[8], uint32_t (&v)[8]) {
    asm volatile("ds_read_b32 %0, %8\n\t"
                 "ds_read_b32 %1, %9\n\t"
                 "ds_read_b32 %2, %10\n\t"
                 "ds_read_b32 %3, %11\n\t"
                 "ds_read_b32 %4, %12\n\t"
                 "ds_read_b32 %5, %13\n\t"
                 "ds_read_b32 %6, %14\n\t"
                 "ds_read_b32 %7, %15\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7])
                 : "memory");
}

// Per-lane constants and the software-pipelined walk over a run of wide blocks, shared by the
// per-window kernel (run = every WAVES-th tile of one window) and the range-blocked kernel
// (run = the tiles of one window inside one column range).
template <int NT, bool VAL>
struct TileWalker {
    static constexpr int TILE_BYTES = NT * 1024;
    static constexpr int WAVE_LDS = 2 * TILE_BYTES + kPadBytes + (VAL ? 2048 : 0);  // two tile buffers, metadata pad, edge values (2 x 4 per lane)
    static constexpr int NIDS = NT + 1 + (VAL ? 1 : 0);
    using Img = TileImage<NT>;
    const SpmmArgs& a;
    __amdgpu_buffer_rsrc_t xrsrc; // X16 as a structured buffer: record = one fp16 row (pitch bytes);
                                  // the gather address row*pitch + offset is formed by the hardware
    MetaSource meta;
    uint32_t ring, pad, vpad, atab; // LDS byte addresses: tile buffers, metadata pad, edge-value pad, A table
    int lane, g, i;
    uint32_t doff[NT];            // byte offset inside a gathered row: first feature column of the pass + 16-byte piece
    uint32_t idaddr[NIDS];        // LDS addresses in the pad: row id of DMA k for this lane, mask word, edge offset
    uint32_t raddr[NT][2];        // LDS address (buffer 0) this lane hands ds_read_b64_tr_b16 for slice s, K half h
    float sa;                     // edge-value scale (VAL)

    __device__ __forceinline__ TileWalker(const SpmmArgs& args, char* wave_lds, char* atab_ptr, int coloff, float sa_)
        : a(args), meta(args.cols, args.mask, args.ebase, threadIdx.x & 63), sa(sa_) {
        lane = threadIdx.x & 63;
        g = lane >> 4;
        i = lane & 15;
        xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x16, (short)(a.stride * 2), a.xrows, 0x00020000);
        ring = (uint32_t)(uintptr_t)((LDS_AS char*)wave_lds);
        pad = ring + 2 * TILE_BYTES;
        vpad = pad + kPadBytes;
        atab = (uint32_t)(uintptr_t)((LDS_AS char*)atab_ptr);
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            int row, c;
            Img::unslot(k * 64 + lane, row, c);
            idaddr[k] = pad + (uint32_t)row * 4u;
            doff[k] = (uint32_t)(coloff * 2 + c * 16);
        }
        idaddr[NT] = pad + 128u + (uint32_t)i * 4u;
        if constexpr (VAL) idaddr[NT + 1] = pad + 192u + (uint32_t)i * 4u;
        // lane (g, i) receives K = 8g + 4h + {0..3} of column 16s + i when it points the transpose
        // read at row 8g + 4h + (i >> 2), halves 16s + 4(i & 3) .. +3
        const int rrow = 8 * g + (i >> 2);
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const int c = 2 * s + ((i >> 1) & 1);
            raddr[s][0] = ring + (uint32_t)(Img::slot(rrow, c) * 16 + (i & 1) * 8);
            raddr[s][1] = ring + (uint32_t)(Img::slot(rrow + 4, c) * 16 + (i & 1) * 8);
        }
    }

    struct Cur {            // the tile being multiplied: its mask word, edge offset, value-window shift; wide: eight values fetched
        uint32_t m, eb;
        int shift;
        bool wide;
    };
    template <int BUF> __device__ __forceinline__ void dma_gather(const uint32_t* cid) const {
        if (a.big) {   // (wave-uniform: a kernel argument)
            const char* const xb = reinterpret_cast<const char*>(a.x16);
            const uint64_t pitchb = (uint64_t)a.stride * 2u;
#pragma unroll
            for (int k = 0; k < NT; ++k)
                __builtin_amdgcn_global_load_lds((GLB_AS const void*)(xb + (uint64_t)cid[k] * pitchb + doff[k]),
                                                 (LDS_AS void*)(uintptr_t)(ring + BUF * TILE_BYTES + k * 1024), 16, 0, 0);
            return;
        }
#pragma unroll
        for (int k = 0; k < NT; ++k)
            __builtin_amdgcn_struct_ptr_buffer_load_lds(xrsrc, (LDS_AS void*)(uintptr_t)(ring + BUF * TILE_BYTES + k * 1024), 16,
                                                        (int)cid[k], (int)doff[k], 0, 0, 0);
    }
    // edge values of this lane's byte of row i: consecutive floats starting at the first edge of the byte (a row's edges inside
    // eight columns are one run of the CSR), clamped so the read stays inside edge_val; they land in this lane's slot of the value
    // pad.  Four cover the run almost always; when some lane's run is longer (hub rows: every second column is an edge) a second
    // DMA fetches the next four - wave-uniform, decided from the mask.
    __device__ __forceinline__ void dma_vals(Cur& c) const {
        const int64_t e0 = (int64_t)c.eb + __popc(c.m & ((1u << (8 * g)) - 1u));
        int64_t lo = e0 < a.E - 8 ? e0 : a.E - 8;
        if (lo < 0) lo = 0;
        c.shift = (int)(e0 - lo);
        c.wide = a.E >= 8 && __any(__popc((c.m >> (8 * g)) & 0xffu) + c.shift > 4);
        __builtin_amdgcn_global_load_lds((GLB_AS const void*)(a.edge_val + lo), (LDS_AS void*)(uintptr_t)vpad, 16, 0, 0);
        if (c.wide) __builtin_amdgcn_global_load_lds((GLB_AS const void*)(a.edge_val + lo + 4), (LDS_AS void*)(uintptr_t)(vpad + 1024u), 16, 0, 0);
    }

    template <int BUF> __device__ __forceinline__ void multiply(const Cur& cur, const uintx4& q, const uint32_t (&sv)[8], floatx4 (&acc)[NT]) const {
        half8 af;
        if constexpr (VAL) {
            const uint32_t mb = (cur.m >> (8 * g)) & 0xffu;
            const int nb = __popc(mb);
            if (!cur.wide && __builtin_expect(__any(nb + cur.shift > 4), 0)) {
                // (fewer than eight edges in the whole matrix) ordinary loads; the compiler drains the DMA queue for them
                const int64_t e0 = (int64_t)cur.eb + __popc(cur.m & ((1u << (8 * g)) - 1u));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool on = (mb >> j) & 1u;
                    const float v = on ? a.edge_val[e0 + __popc(mb & ((1u << j) - 1u))] * sa : 0.0f;
                    af[j] = to_half_rna(v);
                }
            } else {
                // sv[j]: the value of tile column j of my eight, read from the fetched run BY ADDRESS in stage() (r03: the cascade of
                // selects over the fetched registers cost ~10 VALU instructions per column of a loop that is VALU-bound)
#pragma unroll
                for (int j = 0; j < 8; ++j) af[j] = ((mb >> j) & 1u) ? to_half_rna(__uint_as_float(sv[j]) * sa) : (_Float16)0.0f;
            }
        } else {
            af = __builtin_bit_cast(half8, q);   // table entry of this lane's adjacency byte
        }
        half4 lo[NT], hi[NT];
        lds_tr_block<NT, BUF * TILE_BYTES>(raddr, lo, hi);
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const half8 bf = __builtin_shufflevector(lo[s], hi[s], 0, 1, 2, 3, 4, 5, 6, 7);
            acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[s], 0, 0, 0);
        }
    }

    // One pipeline stage, tile t in tile buffer BUF.  On entry the DMA queue holds gather(t),
    // [edge values(t)] and the metadata of the next tile; after ONE wait everything is read from LDS
    // (ids of the next tile + A fragment or edge values of this one), the next gather, the next edge
    // values and the metadata of the tile after that are issued, and tile t is multiplied underneath.
    // Returns false when t was the last tile of the run.
    template <int BUF>
    __device__ __forceinline__ bool stage(int64_t& t, int64_t& tn, const int64_t te, const int64_t step, const int64_t t_after,
                                          Cur& cur, floatx4 (&acc)[NT]) const {
        wait_vm0();
        uint32_t v[NIDS];
        uintx4 q;
        const uint32_t qaddr = VAL ? vpad + (uint32_t)lane * 16u : atab + (((cur.m >> (8 * g)) & 0xffu) << 4);
        lds_ids_block<NIDS>(idaddr, v, qaddr, q);
        [[maybe_unused]] uint32_t sv[8];
        if constexpr (VAL) {   // (before the next tile's values overwrite the pad)
            const uint32_t mb = (cur.m >> (8 * g)) & 0xffu, vbase = vpad + (uint32_t)lane * 16u;
            uint32_t va = vbase + ((uint32_t)cur.shift << 2), ad[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ad[j] = va;
                va -= (uint32_t)((int32_t)(mb << (31 - j)) >> 31) << 2;                 // + 4 where the edge exists
            }
            if (cur.wide) {   // (wave-uniform: some lane's run crosses into the second block of four)
#pragma unroll
                for (int j = 0; j < 8; ++j) ad[j] += (((ad[j] - vbase) >> 4) & 1u) * 1008u;
            }
            lds_read8_b32(ad, sv);
        }
        const bool more = tn < te;
        Cur nx;
        nx.m = v[NT];
        nx.eb = VAL ? v[NT + (VAL ? 1 : 0)] : 0u;
        nx.shift = 0;
        nx.wide = false;
        const int64_t tnn = tn + step;
        if (more) {
            dma_gather<BUF ^ 1>(v);
            if constexpr (VAL) dma_vals(nx);
            const int64_t tf = tnn < te ? tnn : t_after;   // metadata two tiles ahead; at the end of the run: the caller's next run
            if (tf >= 0) meta.dma(tf, pad);
        }
        multiply<BUF>(cur, q, sv, acc);
        cur = nx;
        t = tn;
        tn = tnn;
        return more;
    }

    // acc += A(tiles t, t+step, ... < te) * X16 rows.  `pad_tile` is the tile whose metadata the pad
    // holds (or has in flight) on entry, and on return; passing the caller's next run's first tile as
    // t_after lets the last stage prefetch it.
    __device__ __forceinline__ void walk(int64_t t, const int64_t te, const int64_t step, floatx4 (&acc)[NT], int64_t& pad_tile,
                                         const int64_t t_after) const {
        if (t >= te) return;
        if (pad_tile != t) meta.dma(t, pad);
        wait_vm0();
        uint32_t v[NIDS];
        uintx4 q;
        lds_ids_block<NIDS>(idaddr, v, atab, q);   // (the 16-byte read is a dummy here)
        Cur cur;
        cur.m = v[NT];
        cur.eb = VAL ? v[NT + (VAL ? 1 : 0)] : 0u;
        cur.shift = 0;
        cur.wide = false;
        dma_gather<0>(v);
        if constexpr (VAL) dma_vals(cur);
        int64_t tn = t + step;
        const int64_t tf = tn < te ? tn : t_after;
        if (tf >= 0) meta.dma(tf, pad);
        for (;;) {   // ping-pong over the two tile buffers: static LDS offsets, no register rotation
            if (!stage<0>(t, tn, te, step, t_after, cur, acc)) break;
            if (!stage<1>(t, tn, te, step, t_after, cur, acc)) break;
        }
        pad_tile = t_after;
    }
};

// (second launch-bound argument = waves per SIMD: keeps the register budget at 128 / 256 so the
// accumulators are allocated as VGPRs - with the default budget hipcc split them into AGPRs and
// spent 24 v_accvgpr_* moves per tile shuffling them)
template <int NT, int WAVES, bool VAL>
__global__ __launch_bounds__(WAVES * 64, (NT <= 4 ? 4 : 2)) void spmm_kernel(const SpmmArgs a) {
    if (VAL ? range_is_wide_val(a.hdr) : range_is_wide(a.hdr, 0)) return;   // (range guard: the fp32 fallback launched behind this kernel does the work)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int w = a.order[blockIdx.x];
    const int coloff = (a.chunk0 + (int)blockIdx.y) * kMaxChunkDims; // first feature column of this pass
    const int64_t tb = a.wb_ptr[w], te = a.wb_ptr[w + 1];
    if (a.accumulate && !a.relu && tb == te) return;   // nothing to add to what the LDS-resident kernel stored
    const int kx = scale_exp_from_bits(a.hdr[0]);
    const int ka = VAL ? scale_exp_from_bits(a.hdr[1]) : 0;

    floatx4 acc[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) acc[s] = floatx4{0.f, 0.f, 0.f, 0.f};
    {
        using TW = TileWalker<NT, VAL>;
        char* atab = smem + WAVES * TW::WAVE_LDS;
        fill_afrag_table(atab);
        __syncthreads();
        const TW tw(a, smem + wave * TW::WAVE_LDS, atab, coloff, pow2f(ka));
        int64_t pad_tile = -1;
        tw.walk(tb + wave, te, WAVES, acc, pad_tile, -1);
    }

    // ---- combine the wavefronts' partial sums in a fixed order and store
    const float inv1 = pow2f(-kx), inv2 = VAL ? pow2f(-ka) : 1.0f; // |kx + ka| may exceed 126: two factors
    const int64_t row0 = (int64_t)w * kWinRows + 4 * g;
    if constexpr (!VAL) {
        if (a.w) {   // f3: Y[window] = (A X)[window] W   (one pass: the launcher only takes this path for D <= 128)
            const int din = NT * 16;                               // (padding columns of X16 are zero)
            const int ldp = din + 1;
            float* accT = reinterpret_cast<float*>(smem) + (WAVES > 1 ? WAVES * NT * 256 : 0);   // behind the reduction buffer
            __syncthreads(); // every wave is done with its ring
            if constexpr (WAVES > 1) {
                floatx4* red = reinterpret_cast<floatx4*>(smem);
#pragma unroll
                for (int s = 0; s < NT; ++s) red[(wave * NT + s) * 64 + lane] = acc[s];
                __syncthreads();
                for (int s = wave; s < NT; s += WAVES) {
                    floatx4 v = red[s * 64 + lane];
#pragma unroll
                    for (int ww = 1; ww < WAVES; ++ww) {
                        const floatx4 o = red[(ww * NT + s) * 64 + lane];
                        v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
                    }
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) accT[(4 * g + ii) * ldp + 16 * s + i] = v[ii] * inv1;
                }
            } else {
#pragma unroll
                for (int s = 0; s < NT; ++s)
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) accT[(4 * g + ii) * ldp + 16 * s + i] = acc[s][ii] * inv1;
            }
            __syncthreads();
            const int tiles = (a.dout + 15) >> 4;
            for (int t = wave; t < tiles; t += WAVES) {
                const floatx4 c = dense_update_tile(accT, ldp, a.D < din ? a.D : din, a.w, 0, a.dout, t, g, i);
                const int colg = 16 * t + i;
                if (colg < a.dout) {
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
                        if (row0 + ii < a.N) a.y[(row0 + ii) * a.ldy + colg] = relu_if(a.relu, c[ii]);
                }
            }
            return;
        }
    }
    if constexpr (WAVES > 1) {
        __syncthreads(); // every wave is done with its ring
        floatx4* red = reinterpret_cast<floatx4*>(smem);
#pragma unroll
        for (int s = 0; s < NT; ++s) red[(wave * NT + s) * 64 + lane] = acc[s];
        __syncthreads();
        for (int s = wave; s < NT; s += WAVES) {
            floatx4 v = red[s * 64 + lane];
#pragma unroll
            for (int ww = 1; ww < WAVES; ++ww) {
                const floatx4 o = red[(ww * NT + s) * 64 + lane];
                v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
            }
            const int colg = coloff + 16 * s + i;
            if (colg < a.D) {
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
                    if (row0 + ii < a.N) {
                        float* dst = a.y + (row0 + ii) * a.ldy + colg;
                        *dst = relu_if(a.relu, v[ii] * inv1 * inv2 + (a.accumulate ? *dst : 0.0f));
                    }
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const int colg = coloff + 16 * s + i;
            if (colg < a.D) {
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
                    if (row0 + ii < a.N) {
                        float* dst = a.y + (row0 + ii) * a.ldy + colg;
                        *dst = relu_if(a.relu, acc[s][ii] * inv1 * inv2 + (a.accumulate ? *dst : 0.0f));
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Range-blocked SpMM for graphs whose fp16 feature image does not fit the 4 MB per-XCD L2.
//
// PMC on the plain kernel (profiles/r01): 34 % L2 hit rate, 10 GB of fabric reads per launch for
// 0.58 GB of algorithmic bytes - every window sweeps all of X16, and X16 (29.8 MB at Reddit D=64)
// only fits the Infinity Cache, whose random-row gather rate (8.3 TB/s, tools/gather_bench) is
// less than half the L2's (18 TB/s).  Here the columns are cut into R ranges of ~2 MB of X16.
// A wavefront owns up to MAXW windows for its whole life and keeps their accumulators in
// registers; it walks range 0 of all its windows, then range 1, ...  All resident wavefronts
// start together and advance at the same average pace, so the rows being gathered at any moment
// belong to one or two ranges and stay L2-resident on every XCD.  No partial sums leave the
// registers, no atomics: the result is as deterministic as the plain kernel's.
// ------------------------------------------------------------------------------------------
struct SpmmBlockedArgs {
    SpmmArgs base;
    const uint32_t* bptr;
    int32_t nbuckets, gsel, nranges, nw, ngroups;
};

__global__ void bucket_ptr_kernel(const int64_t* __restrict__ wb_ptr, const int32_t* __restrict__ cols, int32_t nw,
                                  int32_t nbuckets, int32_t bucket_rows, uint32_t* bptr) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)nw * (nbuckets + 1)) return;
    const int w = (int)(idx / (nbuckets + 1)), k = (int)(idx % (nbuckets + 1));
    const int64_t tb = wb_ptr[w];
    const int64_t n = wb_ptr[w + 1] - tb;
    int64_t lo = 0, hi = n;
    if (k == nbuckets) lo = n;
    else
        while (lo < hi) {  // tiles are ordered by column: first tile whose first column is in bucket >= k
            const int64_t mid = (lo + hi) >> 1;
            if (cols[(tb + mid) * kWbCols] / bucket_rows >= k) hi = mid; else lo = mid + 1;
        }
    bptr[idx] = (uint32_t)lo;
}

template <int NT, int MAXW, bool VAL>
__global__ __launch_bounds__(256, (NT <= 4 ? 4 : 2)) void spmm_blocked_kernel(const SpmmBlockedArgs b) {
    if (VAL ? range_is_wide_val(b.base.hdr) : range_is_wide(b.base.hdr, 0)) return;   // (range guard: the fp32 fallback launched behind this kernel does the work)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SpmmArgs& a = b.base;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int coloff = (a.chunk0 + (int)blockIdx.y) * kMaxChunkDims;
    const int kx = scale_exp_from_bits(a.hdr[0]);
    const int ka = VAL ? scale_exp_from_bits(a.hdr[1]) : 0;
    const float inv1 = pow2f(-kx), inv2 = VAL ? pow2f(-ka) : 1.0f;
    using TW = TileWalker<NT, VAL>;
    char* atab = smem + 4 * TW::WAVE_LDS;
    fill_afrag_table(atab);
    __syncthreads();
    const TW tw(a, smem + wave * TW::WAVE_LDS, atab, coloff, pow2f(ka));

    const int gw = blockIdx.x * 4 + wave, gwn = gridDim.x * 4;
    for (int grp = gw; grp < b.ngroups; grp += gwn) {
        int wj[MAXW];
        int64_t tbj[MAXW];
        uint32_t done[MAXW];
        floatx4 acc[MAXW][NT];
#pragma unroll
        for (int j = 0; j < MAXW; ++j) {
            const int idx = grp + j * b.ngroups;   // strided picks from the heaviest-first order: balanced groups
            wj[j] = idx < b.nw ? __builtin_amdgcn_readfirstlane(a.order[idx]) : -1;
            tbj[j] = wj[j] >= 0 ? a.wb_ptr[wj[j]] : 0;
            done[j] = 0;
#pragma unroll
            for (int s = 0; s < NT; ++s) acc[j][s] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
        int64_t pad_tile = -1;
        // run q = (range r, window j); its bounds are looked up one run ahead so the walk can prefetch
        // the ids of the next run's first tile while it finishes the current one
        uint32_t nend[MAXW];
#pragma unroll
        for (int j = 0; j < MAXW; ++j) nend[j] = wj[j] >= 0 ? b.bptr[(int64_t)wj[j] * (b.nbuckets + 1) + b.gsel] : 0u;
        for (int r = 0; r < b.nranges; ++r) {
            uint32_t end[MAXW], after[MAXW];
#pragma unroll
            for (int j = 0; j < MAXW; ++j) {
                end[j] = nend[j];
                after[j] = (wj[j] >= 0 && r + 1 < b.nranges) ? b.bptr[(int64_t)wj[j] * (b.nbuckets + 1) + (int64_t)(r + 2) * b.gsel] : end[j];
                nend[j] = after[j];
            }
#pragma unroll
            for (int j = 0; j < MAXW; ++j) {
                if (wj[j] < 0) continue;
                // first tile of the next non-empty run: window j+1.. of this range, else window 0.. of the next
                int64_t t_after = -1;
#pragma unroll
                for (int jj = MAXW - 1; jj >= 0; --jj)
                    if (wj[jj] >= 0 && after[jj] > end[jj]) t_after = tbj[jj] + end[jj];
#pragma unroll
                for (int jj = MAXW - 1; jj > j; --jj)
                    if (wj[jj] >= 0 && end[jj] > done[jj]) t_after = tbj[jj] + done[jj];
                tw.walk(tbj[j] + done[j], tbj[j] + end[j], 1, acc[j], pad_tile, t_after);
                done[j] = end[j];
            }
        }
#pragma unroll
        for (int j = 0; j < MAXW; ++j) {
            if (wj[j] < 0) continue;
            const int64_t row0 = (int64_t)wj[j] * kWinRows + 4 * g;
#pragma unroll
            for (int s = 0; s < NT; ++s) {
                const int colg = coloff + 16 * s + i;
                if (colg < a.D) {
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
                        if (row0 + ii < a.N) a.y[(row0 + ii) * a.ldy + colg] = relu_if(a.relu, acc[j][s][ii] * inv1 * inv2);
                }
            }
        }
    }
}

#include "tcgnn_lds_spmm.inc"
#include "tcgnn_lds_flat.inc"
#include "tcgnn_lds_val.inc"

// ------------------------------------------------------------------------------------------
// SDDMM:  ef[e] = <X16[row e], X16[col e]>
// ------------------------------------------------------------------------------------------
struct SddmmArgs {
    const int64_t* wb_ptr;
    const int32_t* order;
    const int32_t* cols;
    const uint32_t* mask;
    const int32_t* ebase;
    const _Float16* x16;
    const uint32_t* hdr;
    float* ef;
    int32_t N, Nc, row_off, Dpad, stride;
    const int32_t* rowptr;
    const uint32_t* bptr;       // range-blocked walk (nranges > 0)
    int32_t nbuckets, gsel, nranges, nw;
    int32_t big;                // fp16 image >= 4 GB: 64-bit lane addresses instead of the buffer descriptor
    int32_t xcd;                // range-major walk with XCD affinity: workgroup b (on XCD b % 8) takes the ranges r = b % 8, b % 8 + 8, ... only
};
static constexpr int kXcdCount = 8;   // workgroups are dealt to the XCDs round-robin in launch order

// LDS of one SDDMM wavefront: two operand buffers, the metadata pad, the output staging area
// (16 rows x kSddmmStageCap floats) and 256 bytes of junk slots for lanes that have nothing to stage.
static constexpr int kSddmmStageCap = 64;
#ifndef TCGNN_SDDMM_NBUF
#define TCGNN_SDDMM_NBUF(ks) ((ks) <= 2 ? 2 : 1)
#endif
// Operand buffers: two (gather of the next tile under the multiply of this one) up to D = 64; one beyond,
// where LDS would otherwise allow a single workgroup per CU (latency is then hidden by wavefront count only).
static constexpr int sddmm_nbuf(int ks) { return TCGNN_SDDMM_NBUF(ks); }
static constexpr int sddmm_wave_lds(int ks) { return sddmm_nbuf(ks) * (2 * ks * 1024) + kPadBytes + 16 * kSddmmStageCap * 4 + 256; }

// KS = number of 32-wide k steps (D <= 32*KS <= 128).  The 16 window rows (MFMA A operand) stay in
// registers.  Neighbour rows are the B operand, which for X * X^T is contiguous per lane: lane
// (i, g) needs halves 32*ks + 8g .. +7 of neighbour row i.  Each lane DMAs exactly those 16 bytes
// (structured-buffer addressing: row id * pitch + offset formed by the hardware) into its own LDS
// slot and reads the slot back - LDS is a per-lane landing pad, trivially conflict-free - so the
// gather of the NEXT tile (both 16-column halves) is in flight while the current one is multiplied
// and scattered, without any VGPR holding a load in flight (see "memory pipeline discipline").
template <int KS, int WAVES, bool BLOCKED>
__global__ __launch_bounds__(WAVES * 64, (KS <= 2 ? 4 : 3)) void sddmm_kernel(const SddmmArgs a) {
    if (wide2_dense(a.hdr)) return;   // (range guard: the fp32 fallback launched behind this kernel does the work)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUF_BYTES = 2 * KS * 1024;               // both halves of one tile
    constexpr int WAVE_LDS = sddmm_wave_lds(KS);
    constexpr int CAP = kSddmmStageCap;                    // staged outputs per row before a flush
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int64_t stride = a.stride;
    const int kx = scale_exp_from_bits(a.hdr[0]);
    // ef = acc * 2^(-2kx); one multiply unless 2kx leaves the fp32 exponent range (then two)
    const bool two_step = kx > 63 || kx < -63;
    const float inv_a = two_step ? pow2f(-kx) : pow2f(-2 * kx), inv_b = two_step ? pow2f(-kx) : 1.0f;
    const half8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t below[2] = {(1u << i) - 1u, (1u << (16 + i)) - 1u};   // condensed columns left of mine, per half

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x16, (short)(a.stride * 2), a.Nc + 1, 0x00020000);
    const MetaSource meta(a.cols, a.mask, a.ebase, lane);
    const uint32_t ring = (uint32_t)(uintptr_t)((LDS_AS char*)(smem + wave * WAVE_LDS));
    const uint32_t pad = ring + sddmm_nbuf(KS) * BUF_BYTES;
    // a lane whose k slice lies past Dpad fetches slice 0 instead (valid memory) and is zeroed at use
    uint32_t boff[KS];
    bool bok[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { bok[ks] = ks * 32 + 8 * g < a.Dpad; boff[ks] = bok[ks] ? (uint32_t)(ks * 32 + 8 * g) * 2u : 0u; }
    // WL (KS = 2, rows of one 128-byte line; r03): whole-line gathers, as in agnn_kernel - instruction q takes the eight rows of tile
    // columns 8q .. 8q+7 (lane L: row slot L >> 3, the chunk that belongs at position L & 7) into block q, row-major; chunk c of row
    // slot rho lies at position c ^ 2 (rho >> 1), which spreads the sixteen lanes of every ds_read_b128 lane group of the operand
    // reads (row i of half sub = row slot i & 7 of block 2 sub + (i >> 3), chunk 4 ks + g) over the sixteen 16-byte bank slots.
    // KS = 4 (256-byte rows): four rows per instruction, eight instructions, tile column tau in row slot tau & 3 of block tau >> 2,
    // chunk c at position c ^ (4 (q & 3) + rho).  (KS = 3: the layout above, see agnn_kernel.)
    constexpr bool WL = KS == 2 || KS == 4;
    constexpr int RB = KS == 2 ? 128 : 256, RPI = 1024 / RB, NI = WL ? 32 / RPI : 2, CPR = RB / 16;
    constexpr int NIDR = NI;                                                                      // row ids this lane reads per tile
    auto wl_sw = [](uint32_t rho, uint32_t q) -> uint32_t {
        if constexpr (KS == 2) return 2u * (rho >> 1);
        else return 4u * (q & 3u) + rho;
    };
    uint32_t idaddr[NIDR];
    if constexpr (WL) {
#pragma unroll
        for (int q = 0; q < NI; ++q) idaddr[q] = pad + (uint32_t)(RPI * q + lane / CPR) * 4u;
    } else { idaddr[0] = pad + (uint32_t)i * 4u; idaddr[1] = pad + 64u + (uint32_t)i * 4u; }      // row ids of both halves
    [[maybe_unused]] uint32_t choff[WL ? NI : 1];                                                 // WL: byte offset of my chunk inside the rows instruction q fetches
    if constexpr (WL) {
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const uint32_t c = (uint32_t)(lane % CPR) ^ wl_sw((uint32_t)(lane / CPR), (uint32_t)q);
            choff[q] = ((int)(c * 8u) < a.Dpad) ? c * 16u : 0u;
        }
    }
    uint32_t qaddr[1 + 2 * KS];                                                                   // edge offsets, then my landing slots
    qaddr[0] = pad + 192u + 16u * (uint32_t)g;
#pragma unroll
    for (int k = 0; k < 2 * KS; ++k) {
        if constexpr (WL) {   // row i of half sub = tile column 16 sub + i
            const uint32_t sub = (uint32_t)(k / KS), ks = (uint32_t)(k % KS), tau = 16u * sub + (uint32_t)i, q = tau / RPI, rho = tau % RPI;
            qaddr[1 + k] = ring + q * 1024u + rho * (uint32_t)RB + (((4u * ks + (uint32_t)g) ^ wl_sw(rho, q)) * 16u);
        } else qaddr[1 + k] = ring + (uint32_t)k * 1024u + (uint32_t)lane * 16u;
    }
    const uint32_t m4addr = pad + 128u + 16u * (uint32_t)g;
    // Output staging.  PMC (profiles/r01): scattering every result with its own 4-byte store costs
    // 7.6e7 L2 write requests per launch on top of the 1.25e8 gather reads, and the kernel runs at the
    // L2 request-rate ceiling.  A wavefront walks CONSECUTIVE tiles, and a row's edges in consecutive
    // tiles are consecutive in ef, so results are staged per row in LDS and each row is flushed as one
    // contiguous store (about one store per tile instead of eight, ~10x fewer write requests).
    const uint32_t stg = pad + kPadBytes;
    const uint32_t junk = stg + 16u * CAP * 4u + (uint32_t)lane * 4u;
    const uint32_t stg_row[4] = {stg + (uint32_t)(4 * g + 0) * CAP * 4u, stg + (uint32_t)(4 * g + 1) * CAP * 4u,
                                 stg + (uint32_t)(4 * g + 2) * CAP * 4u, stg + (uint32_t)(4 * g + 3) * CAP * 4u};
    const uint32_t flush_base = stg + (uint32_t)lane * 4u;        // lane j reads the j-th staged result of every row

  // one run: CONSECUTIVE tiles t .. te-1 of window w (step must be 1: the staging relies on it)
  auto run = [&](const int w, int64_t t, const int64_t te, const int64_t step) {
    if (t >= te) return;
    // A operand: window row i, halves 32*ks + 8g .. +7 (rows past N read the zero sentinel row)
    int64_t arow = (int64_t)w * kWinRows + i;
    arow = arow < a.N ? arow + a.row_off : a.Nc;
    const _Float16* ap = a.x16 + arow * stride + 8 * g;
    half8 af[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) af[ks] = (ks * 32 + 8 * g < a.Dpad) ? *reinterpret_cast<const half8*>(ap + ks * 32) : hz;
    // edges of this window live in ef[e_w0 .. ): 32-bit offsets from a wave-uniform base
    const int64_t wrow = (int64_t)w * kWinRows;
    const int64_t e_w0 = a.rowptr[wrow < a.N ? wrow : a.N];
    char* const ef_w = reinterpret_cast<char*>(a.ef + e_w0);
    bool flush_pending = false;
    uint32_t cnt[4] = {0u, 0u, 0u, 0u};                          // staged results of rows 4g .. 4g+3
    uint32_t rstart[4] = {~0u, ~0u, ~0u, ~0u};                   // window-relative ef position of each row's first staged result
    auto flush = [&]() {
        uint32_t vals[16];
        lds_rows_block8<CAP * 4, 0>(flush_base, vals);
        lds_rows_block8<CAP * 4, 8>(flush_base, vals + 8);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t c_r = (uint32_t)__builtin_amdgcn_readlane((int)cnt[r & 3], 16 * (r >> 2));
            const uint32_t s_r = (uint32_t)__builtin_amdgcn_readlane((int)rstart[r & 3], 16 * (r >> 2));
            if ((uint32_t)lane < c_r) *reinterpret_cast<uint32_t*>(ef_w + ((s_r + (uint32_t)lane) << 2)) = vals[r];
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) { cnt[ii] = 0u; rstart[ii] = ~0u; }
    };

    auto dma_b = [&](const uint32_t* cid, int bufbase) {
        if constexpr (WL) {
            if (a.big) {
                const char* const xb = reinterpret_cast<const char*>(a.x16);
#pragma unroll
                for (int q = 0; q < NI; ++q)
                    __builtin_amdgcn_global_load_lds((GLB_AS const void*)(xb + (uint64_t)cid[q] * (uint64_t)(stride * 2) + choff[q]),
                                                     (LDS_AS void*)(uintptr_t)(ring + bufbase + q * 1024), 16, 0, 0);
                return;
            }
#pragma unroll
            for (int q = 0; q < NI; ++q)
                __builtin_amdgcn_struct_ptr_buffer_load_lds(xrsrc, (LDS_AS void*)(uintptr_t)(ring + bufbase + q * 1024), 16, (int)cid[q], (int)choff[q], 0, 0, 0);
            return;
        }
        if (a.big) {
            const char* const xb = reinterpret_cast<const char*>(a.x16);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    __builtin_amdgcn_global_load_lds((GLB_AS const void*)(xb + (uint64_t)cid[sub] * (uint64_t)(stride * 2) + boff[ks]),
                                                     (LDS_AS void*)(uintptr_t)(ring + bufbase + (sub * KS + ks) * 1024), 16, 0, 0);
            return;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                __builtin_amdgcn_struct_ptr_buffer_load_lds(xrsrc, (LDS_AS void*)(uintptr_t)(ring + bufbase + (sub * KS + ks) * 1024), 16,
                                                            (int)cid[sub], (int)boff[ks], 0, 0, 0);
    };
    struct Cur { uintx4 m4, eb4; };
    auto stage = [&](auto BUFC, Cur& cur, int64_t& tcur, int64_t& tn) -> bool {
        constexpr int BUF = decltype(BUFC)::value;
        wait_vm0();
        uint32_t cid[NIDR];
        uintx4 m4n, q[1 + 2 * KS];
        lds_ids_block<NIDR>(idaddr, cid, m4addr, m4n);             // ids + masks of the next tile
        constexpr int NB = sddmm_nbuf(KS);
        lds_q_block<1 + 2 * KS, (NB == 2 ? BUF : 0) * BUF_BYTES>(qaddr, q);   // its edge offsets, and this tile's operands
        const bool more = tn < te;
        const int64_t tnn = tn + step;
        if (more) {
            dma_b(cid, (NB == 2 ? (BUF ^ 1) : 0) * BUF_BYTES);   // (single buffer: its reads above have completed)
            if (tnn < te) meta.dma(tnn, pad);
        }
        // A flush decided at the end of the previous tile is issued HERE, right behind the gather: its stores then have as
        // long to complete as the gather before the next wait (issued after the multiply they held that wait up).
        if (flush_pending) { flush(); flush_pending = false; }
        // ---- tile tcur
        const uint32_t mm[4] = {cur.m4[0], cur.m4[1], cur.m4[2], cur.m4[3]};
        const uint32_t e0 = (uint32_t)e_w0;   // window-relative edge positions (a window holds far fewer than 2^32 edges)
        const uint32_t rbase[4] = {cur.eb4[0] - e0, cur.eb4[1] - e0, cur.eb4[2] - e0, cur.eb4[3] - e0};
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const uint32_t anyrow = ((mm[0] | mm[1] | mm[2] | mm[3]) >> (16 * sub)) & 0xffffu;
            if (__any(anyrow != 0u)) {              // skip a 16-column half no edge lands in
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    // (a lane whose k slice lies past Dpad holds zeros in af: whatever it fetched for B is multiplied away)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks], __builtin_bit_cast(half8, q[1 + sub * KS + ks]), acc, 0, 0, 0);
                }
                // C[row 4g+ii][col i] -> staged at the row's next free slot if the edge exists
                const int bit = 16 * sub + i;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const bool on = (mm[ii] >> bit) & 1u;
                    const uint32_t pos = cnt[ii] + (uint32_t)__popc(mm[ii] & below[sub]);
                    float v = acc[ii] * inv_a;
                    if (two_step) v *= inv_b;
                    lds_write_b32(on ? stg_row[ii] + (pos << 2) : junk, v);
                }
            }
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            if (rstart[ii] == ~0u && mm[ii] != 0u) rstart[ii] = rbase[ii];
            cnt[ii] += (uint32_t)__popc(mm[ii]);
        }
        // a tile adds at most 32 results to a row: flush while every row still has room for one more tile
        const uint32_t fullest = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3]));
        if (!more) flush();
        else flush_pending = __any(fullest > (uint32_t)(CAP - 32));
        cur.m4 = m4n;
        cur.eb4 = q[0];
        tcur = tn;
        tn = tnn;
        return more;
    };

    // prologue: metadata of the first tile, its operands, metadata of the second
    meta.dma(t, pad);
    wait_vm0();
    Cur cur;
    {
        uint32_t cid[NIDR];
        uintx4 e4[1];
        lds_ids_block<NIDR>(idaddr, cid, m4addr, cur.m4);
        lds_q_block<1, 0>(qaddr, e4);
        cur.eb4 = e4[0];
        dma_b(cid, 0);
    }
    int64_t tn = t + step;
    if (tn < te) meta.dma(tn, pad);
    for (;;) {
        if (!stage(std::integral_constant<int, 0>{}, cur, t, tn)) break;
        if (!stage(std::integral_constant<int, 1>{}, cur, t, tn)) break;
    }
    wait_vm0();   // the run's last stores are retired before the next run reuses pad and slots
  };

    if constexpr (BLOCKED) {
        // persistent wavefronts take (column range, window) items in range-major order: at any moment
        // the whole chip gathers from one or two ranges of X16, which stay L2-resident
        if (a.xcd) {
            // XCD affinity (r03, after the whole-line gathers made an L2 hit worth twice a miss): the workgroups of XCD x gather from
            // the ranges x, x + 8, ... only, one after the other - that XCD's L2 is asked for an eighth of the image, a range or two of
            // it at a time, instead of every range every other XCD is walking as well.  Every edge lies in exactly one range, so
            // nothing is added up afterwards and the scores are bit for bit those of the other walks.
            const int x = (int)(blockIdx.x % (unsigned)kXcdCount);
            const int64_t items_x = (int64_t)(a.nranges / kXcdCount) * a.nw, lstride = (int64_t)(gridDim.x / (unsigned)kXcdCount) * WAVES;
            for (int64_t q = (int64_t)(blockIdx.x / (unsigned)kXcdCount) * WAVES + wave; q < items_x; q += lstride) {
                const int rr = (int)(q / a.nw), r = x + kXcdCount * rr;
                const int w = __builtin_amdgcn_readfirstlane(a.order[q - (int64_t)rr * a.nw]);
                const int64_t tb = a.wb_ptr[w];
                const uint32_t* bp = a.bptr + (int64_t)w * (a.nbuckets + 1);
                run(w, tb + bp[r * a.gsel], tb + bp[(r + 1) * a.gsel], 1);
            }
            return;
        }
        const int64_t items = (int64_t)a.nranges * a.nw;
        for (int64_t q = (int64_t)blockIdx.x * WAVES + wave; q < items; q += (int64_t)gridDim.x * WAVES) {
            const int r = (int)(q / a.nw);
            const int w = __builtin_amdgcn_readfirstlane(a.order[q - (int64_t)r * a.nw]);
            const int64_t tb = a.wb_ptr[w];
            const uint32_t* bp = a.bptr + (int64_t)w * (a.nbuckets + 1);
            run(w, tb + bp[r * a.gsel], tb + bp[(r + 1) * a.gsel], 1);
        }
    } else {
        const int w = a.order[blockIdx.x];
        const int64_t tb = a.wb_ptr[w], te = a.wb_ptr[w + 1];
        const int64_t chunk = (te - tb + WAVES - 1) / WAVES;       // contiguous share of this wavefront
        const int64_t t0 = tb + wave * chunk;
        run(w, t0, t0 + chunk < te ? t0 + chunk : te, 1);
    }
}

// Run-time-K variant for D > 128: window rows are re-read per tile (L1-resident), ordinary loads.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sddmm_wide_kernel(const SddmmArgs a) {
    if (wide2_dense(a.hdr)) return;   // (range guard: the fp32 fallback launched behind this kernel does the work)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int w = a.order[blockIdx.x];
    const int64_t tb = a.wb_ptr[w], te = a.wb_ptr[w + 1];
    const int64_t stride = a.stride;
    const float inv = pow2f(-scale_exp_from_bits(a.hdr[0]));
    const int ksteps = (a.Dpad + 31) >> 5;
    const half8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t arow = (int64_t)w * kWinRows + i;
    arow = arow < a.N ? arow + a.row_off : a.Nc;
    const _Float16* ap = a.x16 + arow * stride + 8 * g;
    for (int64_t t = tb + wave; t < te; t += WAVES) {
        const uint4 m4 = *reinterpret_cast<const uint4*>(a.mask + t * kWinRows + 4 * g);
        const int4 eb4 = *reinterpret_cast<const int4*>(a.ebase + t * kWinRows + 4 * g);
        const uint32_t mm[4] = {m4.x, m4.y, m4.z, m4.w};
        const int ee[4] = {eb4.x, eb4.y, eb4.z, eb4.w};
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const uint32_t anyrow = ((m4.x | m4.y | m4.z | m4.w) >> (16 * sub)) & 0xffffu;
            if (!__any(anyrow != 0u)) continue;
            const int cid = a.cols[t * kWbCols + 16 * sub + i];
            const _Float16* bp = a.x16 + (int64_t)cid * stride + 8 * g;
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < ksteps; ++ks) {
                const bool ok = ks * 32 + 8 * g < a.Dpad;
                const half8 av = ok ? *reinterpret_cast<const half8*>(ap + ks * 32) : hz;
                const half8 bf = ok ? *reinterpret_cast<const half8*>(bp + ks * 32) : hz;
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bf, acc, 0, 0, 0);
            }
            const int bit = 16 * sub + i;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                if ((mm[ii] >> bit) & 1u) {
                    const int64_t e = (int64_t)ee[ii] + __popc(mm[ii] & ((1u << bit) - 1u));
                    a.ef[e] = acc[ii] * inv * inv;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fused AGNN products: one gather of a tile's 32 neighbour rows feeds BOTH the edge scores
// (SDDMM) and the edge-weighted aggregation (SpMM), which the AGNN layer always wants together.
//
// Per tile:   S^T = Xc * Xw^T   (MFMA #1, once per half: 16 tile columns as rows m, the 16 window rows as
//             columns n, K = D).  Row m of half `sub` is tile column 8(m>>2) + 4sub + (m&3), so lane (g, i)
//             ends up holding, for window row i, the scores of tile columns 8g .. 8g+7 - exactly an A
//             fragment of MFMA #2 (k slot j <-> tile column 8g + j), and row i's edges inside those eight
//             columns are ONE run of ef.  No data moves between lanes:
//             Y  += att * Xc    (MFMA #2)
// The gathered rows land lane-linear in LDS (lane (g, i) DMAs halves 32ks + 8g.. of row m = i of half
// sub to slot lane*16 of block (sub, ks)); MFMA #1 reads each lane's own slot back, MFMA #2 reads the
// same bytes through ds_read_b64_tr_b16.  One buffer is
// enough: every LDS read of tile t completes before the gather of tile t+1 is issued, and that
// gather is in flight while tile t is multiplied.
//   forward  (BWD = false): ef = scores (staged per row, flushed as contiguous runs, as in
//            sddmm_kernel), att = fl32(w * ef), max |ef| recorded for the backward call's scale.
//   backward (BWD = true):  att = fl32(w * ef_saved) (edge values DMA'd one tile ahead),
//            scores of dY are only reduced against the column ids: sum_e s[e] * (float)col(e).
// ------------------------------------------------------------------------------------------
static constexpr int kAgnnXcds = 8;   // workgroup b runs on XCD b % 8
struct AgnnArgs {
    const int64_t* wb_ptr;
    const int32_t* order;
    const int32_t* cols;
    const uint32_t* mask;
    const int32_t* ebase;
    const _Float16* x16;
    const uint32_t* hdr;
    const float* w;            // attention weight, device scalar
    float* ef;                 // forward: out [E]; backward: the saved scores (read only)
    uint32_t* ef_absmax;       // bit pattern of max |ef|: forward accumulates, backward reads
    float* y;                  // [N, D]
    double* partial;           // backward: one slot per workgroup
    int32_t N, Nc, row_off, Dpad, D, stride;
    int64_t E;
    const int32_t* rowptr;
    const uint32_t* bptr;      // range-major walk (MAXW > 0): per-window tile offsets of the column buckets
    int32_t nbuckets, gsel, nranges, nw, ngroups;
    int32_t big;               // fp16 image >= 4 GB: 64-bit lane addresses instead of the buffer descriptor
    int32_t nslices;           // > 0 (MAXW = 0 only): the XCD-sliced walk, see agnn_kernel
    int32_t valonly;           // backward kernel as an edge-valued SpMM (tcgnn_spmm_val on the sliced walk): Y = sum ef[e] X[col(e)], no scores, w = 1
};

static constexpr int agnn_wave_lds(int ks, bool bwd) {
    return 2 * ks * 1024 + kPadBytes + (bwd ? 2048 : 16 * kSddmmStageCap * 4 + 256);   // backward: two 1 KB edge-value blocks
}

// MAXW = 0: one workgroup per window, each wavefront a contiguous quarter of its tiles.
// MAXW > 0: persistent wavefronts that own MAXW windows (accumulators in registers) and walk the column
//           ranges in step, as spmm_blocked_kernel does, so the gathered rows stay L2-resident.
template <int NT, int WAVES, bool BWD, int MAXW>
__global__ __launch_bounds__(WAVES * 64, (NT <= 4 ? 3 : 2)) void agnn_kernel(const AgnnArgs a) {
    const bool valonly = BWD && a.valonly != 0;   // (kernel-uniform)
    if (valonly ? range_is_wide_val(a.hdr) : wide2_dense(a.hdr)) return;   // (range guard: the fp32 fallback launched behind this kernel does the work; a few dirty rows: wide_patch_kernel)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = (NT + 1) / 2;
    constexpr int WAVE_LDS = agnn_wave_lds(KS, BWD);
    constexpr int CAP = kSddmmStageCap;
    constexpr int NQ = 2 * KS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int64_t stride = a.stride;
    const int kx = scale_exp_from_bits(a.hdr[0]);
    const bool two_step = kx > 63 || kx < -63;                     // score = acc * 2^(-2kx), in two factors if needed
    const float inv_a = two_step ? pow2f(-kx) : pow2f(-2 * kx), inv_b = two_step ? pow2f(-kx) : 1.0f;
    const float wv = a.w ? a.w[0] : 1.0f;
    // power-of-two scale of the edge weights att = w * ef.  forward: |ef| <= Dpad * max|x|^2 (no pass over E);
    // backward: the recorded max |ef|.  Rounding to a 10-bit mantissa does not depend on the scale.
    float att_bound;
    if constexpr (BWD) att_bound = fabsf(wv) * __uint_as_float(a.ef_absmax[0]);
    else { const float xm = __uint_as_float(a.hdr[0]); att_bound = fabsf(wv) * (float)a.Dpad * xm * xm; }
    const int ka = scale_exp_from_bits(__float_as_uint(att_bound));
    const float c_val = wv * pow2f(ka);   // power-of-two scaling commutes with rounding: fl(x * w) * 2^ka == fl(x * (w * 2^ka))
    const half8 hz = {0, 0, 0, 0, 0, 0, 0, 0};

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x16, (short)(a.stride * 2), a.Nc + 1, 0x00020000);
    const MetaSource meta(a.cols, a.mask, a.ebase, lane);
    const uint32_t ring = (uint32_t)(uintptr_t)((LDS_AS char*)(smem + wave * WAVE_LDS));
    const uint32_t pad = ring + 2 * KS * 1024;
    const uint32_t aux = pad + kPadBytes;                          // forward: output staging; backward: edge-value pad
    uint32_t boff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) boff[ks] = (ks * 32 + 8 * g < a.Dpad) ? (uint32_t)(ks * 32 + 8 * g) * 2u : 0u;
    // pad: cols[32] | mask[16] | ebase[16]; this lane: ids of the tile columns it gathers for the two halves
    // (row m = i of half sub is tile column 8(i>>2) + 4sub + (i&3)), mask and edge offset of ROW i
    //
    // WL (KS = 2, rows of one 128-byte line; r03): WHOLE-LINE gathers.  The layout above makes an instruction take 64 bytes of each
    // of sixteen rows, so every line is asked for twice, and tools/gather_bench.hip measures what that costs once the rows come out
    // of L2: 9.2 TB/s at any depth against 14-18 TB/s for instructions that take eight whole rows.  Here instruction q takes the
    // rows of tile columns 8q .. 8q+7 - lane L the 16-byte chunk of row slot rho = L >> 3 that belongs at position p = L & 7 - into
    // block q, row-major.  Chunk c of row slot rho lies at position c ^ sw(rho, q), sw = 4 ((rho >> 1) & 1) + sigma(q),
    // sigma = (0, 2, 3, 1): with it the sixteen lanes of every ds_read_b128 lane group of MFMA #1's operand reads (row i, chunk
    // 4 ks + g) fall into the sixteen 16-byte bank slots, and so do the eight rows x two chunks a 32-lane group of the transposed
    // reads of MFMA #2 addresses.  All of it is per-lane constants: the tile loop issues the same instructions as before, plus two
    // more row ids read from the pad.
    // KS = 4 (rows of two lines, D = 97 .. 128): the same with four 256-byte rows per instruction, eight instructions, tile column tau in
    // row slot tau & 3 of block tau >> 2, sw = 8 ((q >> 1) & 1) + 2 rho.  (KS = 3 keeps the layout above: its 192-byte rows are
    // gathered at a 256-byte pitch, which a row-major image of the tile would have to be sized for.)
    constexpr bool WL = KS == 2 || KS == 4;
    constexpr int RB = KS == 2 ? 128 : 256, RPI = 1024 / RB, NI = WL ? 32 / RPI : 2, CPR = RB / 16;   // row bytes, rows per instruction, instructions, chunks per row
    constexpr int NV = WL ? NI + 2 : 4;                              // words read from the pad per tile: row ids, mask, edge offset
    auto wl_sw = [](uint32_t rho, uint32_t q) -> uint32_t {
        if constexpr (KS == 2) return 4u * ((rho >> 1) & 1u) + ((0x1320u >> (4u * q)) & 3u);
        else return 8u * ((q >> 1) & 1u) + 2u * rho;
    };
    auto wl_pos = [&](uint32_t tau, uint32_t c) -> uint32_t {        // byte position of chunk c of tile column tau inside the tile image
        const uint32_t q = tau / RPI, rho = tau % RPI;
        return q * 1024u + rho * (uint32_t)RB + ((c ^ wl_sw(rho, q)) * 16u);
    };
    const uint32_t pcol = (uint32_t)(8 * (i >> 2) + (i & 3));
    uint32_t idaddr[NV];
    if constexpr (WL) {
#pragma unroll
        for (int q = 0; q < NI; ++q) idaddr[q] = pad + (uint32_t)(RPI * q + lane / CPR) * 4u;
    } else { idaddr[0] = pad + pcol * 4u; idaddr[1] = pad + (pcol + 4u) * 4u; }
    idaddr[NV - 2] = pad + 128u + (uint32_t)i * 4u;
    idaddr[NV - 1] = pad + 192u + (uint32_t)i * 4u;
    [[maybe_unused]] uint32_t choff[WL ? NI : 1];                   // WL: byte offset inside the row this lane fetches with instruction q
    if constexpr (WL) {
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const uint32_t c = (uint32_t)(lane % CPR) ^ wl_sw((uint32_t)(lane / CPR), (uint32_t)q);
            choff[q] = ((int)(c * 8u) < a.Dpad) ? c * 16u : 0u;      // (a chunk past Dpad: chunk 0 instead - valid memory, multiplied by zeros)
        }
    }
    uint32_t qaddr[NQ];
#pragma unroll
    for (int k = 0; k < 2 * KS; ++k) {
        if constexpr (WL) {   // operand (sub, ks) of MFMA #1: row m = i of half sub (tile column 8 (i >> 2) + 4 sub + (i & 3)), chunk 4 ks + g
            const uint32_t sub = (uint32_t)(k / KS), ks = (uint32_t)(k % KS);
            qaddr[k] = ring + wl_pos(8u * (uint32_t)(i >> 2) + 4u * sub + (uint32_t)(i & 3), 4u * ks + (uint32_t)g);
        } else qaddr[k] = ring + (uint32_t)k * 1024u + (uint32_t)lane * 16u;
    }
    [[maybe_unused]] const uint32_t vaddr0 = aux + (uint32_t)lane * 16u;   // backward: this lane's run of saved scores (second block: + 1024)
    const uint32_t caddr[2] = {pad + 32u * (uint32_t)g, pad + 32u * (uint32_t)g + 16u};   // ids of my eight tile columns (backward)
    // transpose reads: this lane addresses k row j = i >> 2 (row m = 4g + j of half h), feature quad q = i & 3 of slice s
    uint32_t raddr[NT][2];
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (WL) {   // row m = 4g + (i >> 2) of half h = tile column 8g + 4h + (i >> 2); features 16 s + 4 (i & 3) ..
                raddr[s][h] = ring + wl_pos(8u * (uint32_t)g + 4u * (uint32_t)h + (uint32_t)(i >> 2), 2u * (uint32_t)s + (uint32_t)((i & 3) >> 1)) + 8u * (uint32_t)(i & 1);
            } else
                raddr[s][h] = ring + (uint32_t)((h * KS + (s >> 1)) * 1024 + ((2 * (s & 1) + ((i & 3) >> 1)) * 16 + 4 * g + (i >> 2)) * 16 + 8 * (i & 1));
        }
    const uint32_t stg_i = aux + (uint32_t)i * CAP * 4u;
    const uint32_t junk = aux + 16u * CAP * 4u + (uint32_t)lane * 4u;
    const uint32_t flush_base = aux + (uint32_t)lane * 4u;
    const uint32_t low8 = (1u << (8 * g)) - 1u;                                         // condensed columns left of my eight
    const uint32_t halfbits[2] = {0x0f0f0f0fu, 0xf0f0f0f0u};                            // tile columns of each half

    uint32_t emax = 0u;
    float dsum = 0.f;

    // one run: consecutive tiles t .. te-1 of window w, accumulated into acc
    auto run = [&](const int w, int64_t t, const int64_t te, floatx4 (&acc)[NT]) {
        if (t >= te) return;
        // B operand of MFMA #1: window row i, halves 32*ks + 8g .. +7 (rows past N read the zero sentinel row)
        int64_t arow = (int64_t)w * kWinRows + i;
        arow = arow < a.N ? arow + a.row_off : a.Nc;
        const _Float16* ap = a.x16 + arow * stride + 8 * g;
        half8 af[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[ks] = (ks * 32 + 8 * g < a.Dpad) ? *reinterpret_cast<const half8*>(ap + ks * 32) : hz;
        const int64_t wrow = (int64_t)w * kWinRows;
        const int64_t e_w0 = a.rowptr[wrow < a.N ? wrow : a.N];
        char* const ef_w = reinterpret_cast<char*>(a.ef + e_w0);
        uint32_t cnt = 0u, rstart = ~0u;                           // staged results of row i / window-relative position of the first
        [[maybe_unused]] bool flush_pending = false;

        auto flush = [&]() {
            uint32_t vals[16];
            lds_rows_block8<CAP * 4, 0>(flush_base, vals);
            lds_rows_block8<CAP * 4, 8>(flush_base, vals + 8);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t c_r = (uint32_t)__builtin_amdgcn_readlane((int)cnt, r);
                const uint32_t s_r = (uint32_t)__builtin_amdgcn_readlane((int)rstart, r);
                if ((uint32_t)lane < c_r) {
                    st_stream(reinterpret_cast<uint32_t*>(ef_w + ((s_r + (uint32_t)lane) << 2)), vals[r]);
                    const uint32_t ab = vals[r] & 0x7fffffffu;           // max |ef| for the backward call's scale
                    emax = ab > emax ? ab : emax;
                }
            }
            cnt = 0u; rstart = ~0u;
        };
        auto dma_b = [&](const uint32_t* cid) {
            if constexpr (WL) {   // NI instructions of RPI whole rows each
                if (MAXW == 0 && a.big) {
                    const char* const xb = reinterpret_cast<const char*>(a.x16);
#pragma unroll
                    for (int q = 0; q < NI; ++q)
                        __builtin_amdgcn_global_load_lds((GLB_AS const void*)(xb + (uint64_t)cid[q] * (uint64_t)(stride * 2) + choff[q]),
                                                         (LDS_AS void*)(uintptr_t)(ring + q * 1024), 16, 0, 0);
                    return;
                }
#pragma unroll
                for (int q = 0; q < NI; ++q)
                    __builtin_amdgcn_struct_ptr_buffer_load_lds(xrsrc, (LDS_AS void*)(uintptr_t)(ring + q * 1024), 16, (int)cid[q], (int)choff[q], 0, 0, 0);
                return;
            }
            if (MAXW == 0 && a.big) {   // (the range-major variant, on request only, is at its register limit: the launcher keeps big images off it)
                const char* const xb = reinterpret_cast<const char*>(a.x16);
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        __builtin_amdgcn_global_load_lds((GLB_AS const void*)(xb + (uint64_t)cid[sub] * (uint64_t)(stride * 2) + boff[ks]),
                                                         (LDS_AS void*)(uintptr_t)(ring + (sub * KS + ks) * 1024), 16, 0, 0);
                return;
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    __builtin_amdgcn_struct_ptr_buffer_load_lds(xrsrc, (LDS_AS void*)(uintptr_t)(ring + (sub * KS + ks) * 1024), 16,
                                                                (int)cid[sub], (int)boff[ks], 0, 0, 0);
        };
        struct Cur { uint32_t m, eb; int sh; bool wide; uintx4 c[2]; };
        // saved scores of row i inside my eight tile columns: one run of ef.  Four floats (clamped to stay inside ef)
        // cover it almost always; a second DMA fetches the next four when some lane's run is longer.
        auto dma_vals = [&](Cur& c) {
            // (32-bit arithmetic: edge offsets are int32 by the CSR's type)
            const int32_t e0 = (int32_t)c.eb + __popc(c.m & low8);
            int32_t lo = min(e0, (int32_t)a.E - 8);
            lo = max(lo, 0);
            c.sh = e0 - lo;
            c.wide = __any(__popc((c.m >> (8 * g)) & 0xffu) + c.sh > 4);
            __builtin_amdgcn_global_load_lds((GLB_AS const void*)(a.ef + lo), (LDS_AS void*)(uintptr_t)aux, 16, 0, 0);
            if (c.wide) __builtin_amdgcn_global_load_lds((GLB_AS const void*)(a.ef + lo + 4), (LDS_AS void*)(uintptr_t)(aux + 1024), 16, 0, 0);
        };
        auto stage = [&](Cur& cur, int64_t& tcur, int64_t& tn) -> bool {
            wait_vm0();
            uint32_t v[NV];
            uintx4 q[NQ];
            lds_ids_block<NV>(idaddr, v, qaddr[0], q[0]);         // next tile: its row ids for my lane, my row's mask and edge offset; + operand 0
            if (!valonly) lds_q_block<NQ - 1, 0>(qaddr + 1, q + 1);   // the other operands (values only: no scores, nothing reads them)
            // backward: the saved score of tile column j of my eight is word (edges of row i left of it in my run) of the run the
            // DMA fetched - read by ADDRESS (r03; a cascade of selects over eight registers cost ten VALU instructions per column
            // in a loop that is VALU-bound: 265 per tile, SQ_INSTS_VALU of profiles/r02).  Lanes without the edge read a
            // neighbouring word that the mask removes below.
            [[maybe_unused]] uint32_t sv[8];
            if constexpr (BWD) {
                const uint32_t byte0 = (cur.m >> (8 * g)) & 0xffu;
                uint32_t va = vaddr0 + ((uint32_t)cur.sh << 2), ad[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    ad[j] = va;
                    va -= (uint32_t)((int32_t)(byte0 << (31 - j)) >> 31) << 2;        // + 4 where the edge exists
                }
                if (cur.wide) {   // (wave-uniform, rare: some lane's run crosses into the second block of four)
#pragma unroll
                    for (int j = 0; j < 8; ++j) ad[j] += (((ad[j] - vaddr0) >> 4) & 1u) * 1008u;
                }
                lds_read8_b32(ad, sv);
            }
            Cur nx;
            nx.m = v[NV - 2]; nx.eb = v[NV - 1]; nx.sh = 0; nx.wide = false;
            if constexpr (BWD) lds_q_block<2, 0>(caddr, nx.c);
            half4 lo[NT], hi[NT];
            lds_tr_block<NT, 0>(raddr, lo, hi);
            const bool more = tn < te;
            const int64_t tnn = tn + 1;
            if (more) {
                dma_b(v);
                if constexpr (BWD) dma_vals(nx);
                if (tnn < te) meta.dma(tnn, pad);
            }
            if constexpr (!BWD) {   // (see sddmm_kernel: a pending flush goes right behind the gather)
                if (flush_pending) { flush(); flush_pending = false; }
            }
            // ---- tile tcur: scores
            floatx4 S[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                S[sub] = floatx4{0.f, 0.f, 0.f, 0.f};
                if (!valonly && __any((cur.m & halfbits[sub]) != 0u)) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        S[sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, q[sub * KS + ks]), af[ks], S[sub], 0, 0, 0);
                }
            }
            // ---- edge weights of row i for my eight tile columns (branch-free; 10-bit RNA rounding done in integer
            //      arithmetic, after which the round-toward-zero pack conversion is exact)
            const uint32_t byte = (cur.m >> (8 * g)) & 0xffu;
            // (the tile loop is VALU-bound - 160 VALU instructions per tile before this form - so the eight columns are
            // handled with masks instead of compares and selects: mk = all-ones where the edge exists; the second scale
            // factor is 1.0 unless the exponent needs two steps, and multiplying by it is exact)
            [[maybe_unused]] uint32_t wpos = stg_i + ((cnt + (uint32_t)__popc(cur.m & low8)) << 2);   // forward: its staging slot
            uint32_t rb[8];
            [[maybe_unused]] float scv[8];
            if constexpr (!BWD) {
#pragma unroll
                for (int j = 0; j < 8; ++j) scv[j] = S[j >> 2][j & 3] * inv_a;
                if (__builtin_expect(two_step, 0)) {   // (kernel-uniform, almost never taken; the empty asm keeps it a branch - if-converted
                                                       //  it costs a multiply and a select per column)
                    asm volatile("; second scale factor");
#pragma unroll
                    for (int j = 0; j < 8; ++j) scv[j] *= inv_b;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t mk = (uint32_t)((int32_t)(byte << (31 - j)) >> 31);   // (one v_bfe_i32)
                [[maybe_unused]] const float sraw = S[j >> 2][j & 3];
                [[maybe_unused]] float sc = 0.f;
                if constexpr (!BWD) sc = scv[j];
                float att_s;
                if constexpr (BWD) {
                    att_s = __uint_as_float(sv[j]) * c_val;                          // = fl32(w * ef) * 2^ka
                } else {
                    lds_write_b32(bitfield_select(mk, wpos, junk), sc);
                    asm("v_mad_i32_i24 %0, %1, -4, %0" : "+v"(wpos) : "v"(mk));           // wpos += 4 where the edge exists (mk = -1)
                    att_s = sc * c_val;                                              // = fl32(w * ef) * 2^ka
                }
                // (+ half an ulp of the 10-bit mantissa; the 13 bits below it are cut by the round-toward-zero pack conversion itself -
                //  in fp16's normal range exactly the bits `& 0xffffe000` would clear, below it a coarser cut toward zero of the same value)
                rb[j] = (__float_as_uint(att_s) & mk) + 0x1000u;
            }
            if constexpr (BWD) {
                if (!valonly) {   // sum_e s[e] * col(e) (the raw accumulators: their power-of-two scale is applied once, to the workgroup's sum)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t mk = (uint32_t)((int32_t)(byte << (31 - j)) >> 31);
                        dsum += __uint_as_float(__float_as_uint(S[j >> 2][j & 3]) & mk) * (float)(int32_t)cur.c[j >> 2][j & 3];
                    }
                }
            }
            half8 a16;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const auto pk = __builtin_amdgcn_cvt_pkrtz(__uint_as_float(rb[j]), __uint_as_float(rb[j + 1]));
                a16[j] = (_Float16)pk[0];
                a16[j + 1] = (_Float16)pk[1];
            }
            // ---- aggregation
#pragma unroll
            for (int s = 0; s < NT; ++s) {
                const half8 bf = __builtin_shufflevector(lo[s], hi[s], 0, 1, 2, 3, 4, 5, 6, 7);
                acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16, bf, acc[s], 0, 0, 0);
            }
            if constexpr (!BWD) {
                if (rstart == ~0u && cur.m != 0u) rstart = cur.eb - (uint32_t)e_w0;
                cnt += (uint32_t)__popc(cur.m);
                // a tile adds at most 32 results to a row: flush while every row still has room for one more tile
                if (!more) flush();
                else flush_pending = __any(cnt > (uint32_t)(CAP - 32));
            }
            cur = nx;
            tcur = tn;
            tn = tnn;
            return more;
        };

        // prologue: metadata of the first tile, its operands [and saved scores], metadata of the second
        meta.dma(t, pad);
        wait_vm0();
        Cur cur;
        {
            uint32_t v[NV];
            uintx4 dummy;
            lds_ids_block<NV>(idaddr, v, qaddr[0], dummy);
            cur.m = v[NV - 2]; cur.eb = v[NV - 1]; cur.sh = 0; cur.wide = false;
            if constexpr (BWD) lds_q_block<2, 0>(caddr, cur.c);
            dma_b(v);
            if constexpr (BWD) dma_vals(cur);
        }
        int64_t tn = t + 1;
        if (tn < te) meta.dma(tn, pad);
        while (stage(cur, t, tn)) {}
        wait_vm0();
    };

    const float inv1 = pow2f(-kx), inv2 = pow2f(-ka);
    auto store_rows = [&](const int w, const int s, const floatx4& v) {
        const int64_t row0 = (int64_t)w * kWinRows + 4 * g;
        const int colg = 16 * s + i;
        if (colg < a.D) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
                if (row0 + ii < a.N) a.y[(row0 + ii) * a.D + colg] = v[ii] * inv1 * inv2;
        }
    };

    [[maybe_unused]] int w0 = 0;
    [[maybe_unused]] floatx4 acc0[NT];
    if constexpr (MAXW == 0) {
#pragma unroll
        for (int s = 0; s < NT; ++s) acc0[s] = floatx4{0.f, 0.f, 0.f, 0.f};
        if (a.nslices > 0) {
            // XCD-sliced walk (r03).  Workgroups are dealt to the eight XCDs round-robin (workgroup b runs on XCD b % 8), and here
            // workgroup b only ever gathers rows of column slice b % nslices: an XCD's 4 MB L2 then holds the one slice of the fp16
            // image it is asked for (Reddit shape, D = 64: 29.8 MB / 8) instead of seeing all of it - tools/gather_bench.hip: rows out
            // of an L2-resident slice arrive at 14-17 TB/s against 8.3 TB/s for rows of the whole image, which is what the
            // per-window walk runs at.  Every wavefront takes the tiles of ONE window inside the slice (bptr: the plan's bucket
            // table) and stores its sums as that slice's addend of Y (a.y = nslices buffers), which agnn_slice_sum_kernel adds in
            // slice order: deterministic, and nothing depends on the placement being what is assumed here.
            // (more than eight slices - an image of 16 .. 32 MB - go in ROUNDS of eight: the grid's first 1 / rounds takes slices
            //  0 .. 7, the next one slices 8 .. 15, and workgroups start in grid order, so an XCD is asked for one slice at a time)
            const unsigned per_round = gridDim.x / (unsigned)(a.nslices / kAgnnXcds), b2 = blockIdx.x % per_round;
            const int slice = (int)(blockIdx.x / per_round) * kAgnnXcds + (int)(b2 % (unsigned)kAgnnXcds);
            const int wi = (int)(b2 / (unsigned)kAgnnXcds) * WAVES + wave;
            w0 = wi < a.nw ? __builtin_amdgcn_readfirstlane(a.order[wi]) : -1;
            if (w0 >= 0) {
                const int64_t tb = a.wb_ptr[w0];
                const uint32_t* bp = a.bptr + (int64_t)w0 * (a.nbuckets + 1);
                run(w0, tb + bp[slice * a.gsel], tb + bp[(slice + 1) * a.gsel], acc0);
            }
        } else {
            w0 = a.order[blockIdx.x];
            const int64_t tb = a.wb_ptr[w0], te_w = a.wb_ptr[w0 + 1];
            const int64_t chunk = (te_w - tb + WAVES - 1) / WAVES;         // contiguous share of this wavefront
            const int64_t t0 = tb + wave * chunk;
            run(w0, t0, t0 + chunk < te_w ? t0 + chunk : te_w, acc0);
        }
    } else {
        const int gw = blockIdx.x * WAVES + wave, gwn = gridDim.x * WAVES;
        for (int grp = gw; grp < a.ngroups; grp += gwn) {
            int wj[MAXW];
            int64_t tbj[MAXW];
            floatx4 acc[MAXW][NT];
#pragma unroll
            for (int j = 0; j < MAXW; ++j) {
                const int idx = grp + j * a.ngroups;   // strided picks from the heaviest-first order: balanced groups
                wj[j] = idx < a.nw ? __builtin_amdgcn_readfirstlane(a.order[idx]) : -1;
                tbj[j] = wj[j] >= 0 ? a.wb_ptr[wj[j]] : 0;
#pragma unroll
                for (int s = 0; s < NT; ++s) acc[j][s] = floatx4{0.f, 0.f, 0.f, 0.f};
            }
            for (int r = 0; r < a.nranges; ++r) {
#pragma unroll
                for (int j = 0; j < MAXW; ++j) {
                    if (wj[j] < 0) continue;
                    const uint32_t* bp = a.bptr + (int64_t)wj[j] * (a.nbuckets + 1);
                    run(wj[j], tbj[j] + bp[r * a.gsel], tbj[j] + bp[(r + 1) * a.gsel], acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < MAXW; ++j) {
                if (wj[j] < 0) continue;
#pragma unroll
                for (int s = 0; s < NT; ++s) store_rows(wj[j], s, acc[j][s]);
            }
        }
    }

    if constexpr (BWD) {
        // sum_e score(e) * col(e): per-wavefront double, then one slot per workgroup (fixed order -> deterministic)
        double d = (double)dsum * (double)inv_a * (double)inv_b;   // (dsum holds raw MFMA sums: scores * 2^(2 kx))
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) d += __shfl_xor(d, off, 64);
        __syncthreads();
        double* dred = reinterpret_cast<double*>(smem);
        if (lane == 0) dred[wave] = d;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int ww = 0; ww < WAVES; ++ww) tot += dred[ww];
            a.partial[blockIdx.x] = tot;
        }
    } else {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)emax, off, 64); emax = o > emax ? o : emax; }
        // one word for the whole launch: 600k same-address atomics serialise in one L2 channel (12 ns each - they cost
        // 3 ms on the ogbn-products shape).  The running maximum stops growing after a handful of wavefronts, so look first.
        if (lane == 0 && emax != 0u && emax > __hip_atomic_load(a.ef_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(a.ef_absmax, emax);
    }

    // ---- per-window workgroups: combine the wavefronts' partial sums in a fixed order and store
    if constexpr (MAXW == 0) {
        if (a.nslices > 0) {   // a wavefront = a window: its sums are one slice's addend
            if (w0 >= 0) {
                const unsigned per_round = gridDim.x / (unsigned)(a.nslices / kAgnnXcds);
                const int slice = (int)(blockIdx.x / per_round) * kAgnnXcds + (int)((blockIdx.x % per_round) % (unsigned)kAgnnXcds);
                const int64_t row0 = (int64_t)w0 * kWinRows + 4 * g;
                float* const yp = a.y + (int64_t)slice * a.N * a.D;
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    const int colg = 16 * s + i;
                    if (colg < a.D) {
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii)
                            if (row0 + ii < a.N) st_stream(&yp[(row0 + ii) * a.D + colg], acc0[s][ii] * inv1 * inv2);
                    }
                }
            }
        } else if constexpr (WAVES > 1) {
            __syncthreads(); // every wave is done with its LDS
            floatx4* red = reinterpret_cast<floatx4*>(smem);
#pragma unroll
            for (int s = 0; s < NT; ++s) red[(wave * NT + s) * 64 + lane] = acc0[s];
            __syncthreads();
            for (int s = wave; s < NT; s += WAVES) {
                floatx4 v = red[s * 64 + lane];
#pragma unroll
                for (int ww = 1; ww < WAVES; ++ww) {
                    const floatx4 o = red[(ww * NT + s) * 64 + lane];
                    v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
                }
                store_rows(w0, s, v);
            }
        } else {
#pragma unroll
            for (int s = 0; s < NT; ++s) store_rows(w0, s, acc0[s]);
        }
    }
}

// Y = sum of the XCD-sliced walk's addends, in slice order (float4 where the pointers allow)
__global__ __launch_bounds__(256) void agnn_slice_sum_kernel(const float* __restrict__ part, float* __restrict__ y, int64_t n, int64_t stride, int32_t nslices) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    if (((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 && ((n | stride) & 3) == 0) {
        const int64_t n4 = n >> 2, s4 = stride >> 2;
        const float4* p4 = reinterpret_cast<const float4*>(part);
        for (int64_t k = gid; k < n4; k += gsz) {
            float4 v = p4[k];
            for (int s = 1; s < nslices; ++s) { const float4 o = p4[(int64_t)s * s4 + k]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            reinterpret_cast<float4*>(y)[k] = v;
        }
    } else {
        for (int64_t k = gid; k < n; k += gsz) {
            float v = part[k];
            for (int s = 1; s < nslices; ++s) v += part[(int64_t)s * stride + k];
            y[k] = v;
        }
    }
}

// partial[0..n) -> out[0], fixed order.  One workgroup of 1024, four independent loads per thread and step: the sliced walk leaves
// 29 k partials on the Reddit shape, and 256 threads taking one dependent load per step spent 47 us on them (a memory round trip
// per step) - 2 x 47 us of an AGNN epoch.
static constexpr int kReduceThreads = 1024;
__global__ __launch_bounds__(kReduceThreads) void agnn_reduce_kernel(const double* __restrict__ partial, int32_t n, float* __restrict__ out, const double* __restrict__ extra = nullptr) {
    __shared__ double sh[kReduceThreads];
    double s = 0.0;
    int k = (int)threadIdx.x;
    for (; k + 3 * kReduceThreads < n; k += 4 * kReduceThreads) {
        const double a = partial[k], b = partial[k + kReduceThreads], c = partial[k + 2 * kReduceThreads], d = partial[k + 3 * kReduceThreads];
        s += (a + b) + (c + d);
    }
    for (; k < n; k += kReduceThreads) s += partial[k];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = kReduceThreads / 2; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(sh[0] + (extra ? extra[0] : 0.0));   // (extra: wide_patch_kernel's correction, header words 10-11; zero unless the call was patched)
}

// ------------------------------------------------------------------------------------------
// Small graphs: SpMM in ONE launch (binary A).
// Citeseer / Cora / Pubmed-sized inputs are launch-latency work (reference: 0.040 ms per call on an RTX 3090,
// logs/profile.csv:2): the fp16 path costs a memset + absmax + convert + the kernel.  Here one wavefront per window reads
// fp32 X directly, rounds each operand to a 10-bit mantissa like the reference's TF32 conversion (round_rna10: no range
// limit, so no scale pass) and multiplies on the fp32 matrix pipe: v_mfma_f32_16x16x4_f32, A = 16 rows x 4 condensed
// columns of the adjacency mask as 0.0 / 1.0, B = the 4 gathered rows x 16 feature columns.  1/16 of the fp16 MFMA rate,
// irrelevant at this size.  Accumulation in tile order, fp32, like the other kernels.
// ------------------------------------------------------------------------------------------
struct SpmmSmallArgs {
    const int64_t* wb_ptr;
    const int32_t* cols;
    const uint32_t* mask;
    const float* x;
    const float* gate;   // optional: operand element (r, c) counts only where gate[r, c] > 0 (ReLU backward mask)
    float* y;
    int32_t N, Nc, D, relu;
    const uint32_t* guard;   // nullptr: the kernel of small graphs.  Else the header of a staged image: this launch is the range guard's
                             // fallback behind an fp16-path kernel and returns at once unless that matrix is "wide" (range_is_wide)
};
typedef float floatx4s __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void spmm_small_kernel(const SpmmSmallArgs a) {
    if (a.guard && !range_is_wide(a.guard, 0)) return;
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    const int w = blockIdx.x;
    const int coloff = (int)blockIdx.y * 64;
    floatx4s acc[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[s] = floatx4s{0.f, 0.f, 0.f, 0.f};
    const int64_t tb = a.wb_ptr[w], te = a.wb_ptr[w + 1];
    for (int64_t t = tb; t < te; ++t) {
        const uint32_t m = a.mask[t * kWinRows + i];
        int32_t id[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) id[j] = a.cols[t * kWbCols + 4 * j + g];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float av = ((m >> (4 * j + g)) & 1u) ? 1.0f : 0.0f;
            const int64_t roff = (int64_t)id[j] * a.D + coloff + i;
            const float* row = a.x + roff;
            const bool live = id[j] < a.Nc;   // Nc = "no column": the zero sentinel of the packed stream
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float bv = (live && coloff + 16 * s + i < a.D) ? round_rna10(row[16 * s]) : 0.0f;
                if (a.gate && live && coloff + 16 * s + i < a.D && !(a.gate[roff + 16 * s] > 0.0f)) bv = 0.0f;
                acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[s], 0, 0, 0);
            }
        }
    }
    const int64_t row0 = (int64_t)w * kWinRows + 4 * g;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int colg = coloff + 16 * s + i;
        if (colg < a.D) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
                if (row0 + ii < a.N) a.y[(row0 + ii) * a.D + colg] = relu_if(a.relu, acc[s][ii]);
        }
    }
}
static constexpr int64_t kSmallMaxTiles = 8192;   // wide blocks up to which the single-launch kernel is used (mode 0)

// ------------------------------------------------------------------------------------------
// fallbacks for non-canonical CSR rows (unsorted or duplicated column ids)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spmm_val_csr_kernel(const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col,
                                                           const float* __restrict__ val,
                                                           const float* __restrict__ X, float* Y,
                                                           int32_t N, int32_t D) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
    for (int d = lane; d < D; d += 64) {
        float s = 0.f;
        for (int64_t e = e0; e < e1; ++e) s += round_rna10(val[e]) * round_rna10(X[(int64_t)col[e] * D + d]);
        Y[row * D + d] = s;
    }
}

__global__ __launch_bounds__(256) void sddmm_csr_kernel(const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ col,
                                                        const float* __restrict__ X, float* ef,
                                                        int32_t N, int32_t D, int32_t row_off) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    const float* xr = X + (row + row_off) * D;
    for (int64_t e = rowptr[row]; e < rowptr[row + 1]; ++e) {
        const float* xc = X + (int64_t)col[e] * D;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += round_rna10(xr[d]) * round_rna10(xc[d]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) ef[e] = s;
    }
}

// ---- fallbacks behind the range guard (range_is_wide): launched behind every fp16-path kernel, they return at once unless the
//      staged matrix is "wide".  Plain fp32, CSR order, operands rounded to a 10-bit mantissa exactly like the reference's
//      (round_rna10) with fp32's full exponent: correct for any magnitudes, far from fast - a wide matrix is a rare input.
// Y[row] = [relu] sum_e v_e * rna(X'[col e]) with v_e = 1 (binary), rna(val[e]) or rna(fl32(w * val[e])) (the AGNN edge weights);
// X' = X where gate > 0 (the fused ReLU backward mask).  ldx / ldy: row strides (column blocks of wider matrices).
__global__ __launch_bounds__(256) void spmm_wide_fallback_kernel(const uint32_t* __restrict__ hdr, int use_val_word, const int32_t* __restrict__ rowptr,
                                                                 const int32_t* __restrict__ col, const float* __restrict__ val, const float* __restrict__ wscale,
                                                                 const float* __restrict__ X, const float* __restrict__ gate, float* __restrict__ Y, int32_t N, int32_t D,
                                                                 int64_t ldx, int64_t ldy, int32_t relu, int32_t dedupe) {
    if (!(use_val_word ? range_is_wide_val(hdr) : range_is_wide(hdr, 0))) return;
    const int lane = threadIdx.x & 63;
    const float w = wscale ? wscale[0] : 1.0f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < N; row += (int64_t)gridDim.x * 4) {   // (a small grid: the usual launch returns above)
    const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
    for (int d = lane; d < D; d += 64) {
        float s = 0.f;
        for (int64_t e = e0; e < e1; ++e) {
            if (dedupe) {   // binary A on a non-canonical row: an edge listed twice counts once (TCGNN_kernel.cu:405)
                bool dup = false;
                for (int64_t e2 = e0; e2 < e; ++e2) dup = dup || col[e2] == col[e];
                if (dup) continue;
            }
            const int64_t xi = (int64_t)col[e] * ldx + d;
            float x = X[xi];
            if (gate && !(gate[xi] > 0.0f)) x = 0.0f;
            const float v = val ? round_rna10(wscale ? w * val[e] : val[e]) : 1.0f;
            s += v * round_rna10(x);
        }
        Y[row * ldy + d] = relu_if(relu, s);
    }
    }
}
// Y[row] = [relu] ((A X)[row]) W: the aggregated row goes through LDS, then every lane takes output columns (D_in, D_out <= 128)
__global__ __launch_bounds__(256) void spmm_gemm_wide_fallback_kernel(const uint32_t* __restrict__ hdr, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                      const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Y, int32_t N,
                                                                      int32_t Din, int32_t Dout, int32_t relu) {
    if (!range_is_wide(hdr, 0)) return;
    __shared__ float agg[4][128];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t row0 = (int64_t)blockIdx.x * 4; row0 < N; row0 += (int64_t)gridDim.x * 4) {
    const int64_t row = row0 + wv;
    __syncthreads();
    if (row < N) {
        const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
        for (int d = lane; d < Din; d += 64) {
            float s = 0.f;
            for (int64_t e = e0; e < e1; ++e) s += round_rna10(X[(int64_t)col[e] * Din + d]);
            agg[wv][d] = s;
        }
    }
    __syncthreads();
    if (row < N)
    for (int o = lane; o < Dout; o += 64) {
        float s = 0.f;
        for (int k = 0; k < Din; ++k) s = fmaf(agg[wv][k], W[(int64_t)k * Dout + o], s);
        Y[row * Dout + o] = relu_if(relu, s);
    }
    }
}
// ---- the sparse way through a wide matrix (see wide2_sparse): every edge (r, c) whose row of X or whose column's row of X is dirty
//      is recomputed in fp32 with the reference's operand rounding.
//   mode 0 (SDDMM):          ef[e] = <rna(x_r), rna(x_c)>
//   mode 1 (fused forward):  the same, max |ef| kept up to date, and Y[r] += rna(w ef_new) rna(x_c) - rna(w ef_old) img(x_c): the
//                            edge's contribution as the reference computes it minus what the MFMA kernel added (img = the fp16 image)
//   mode 2 (fused backward): X = dY, ef = the saved scores: G[r] += rna(w ef) (rna(x_c) - img(x_c)) and the edge's term of d_w,
//                            (<rna(x_r), rna(x_c)> - <img(x_r), img(x_c)>) col(e), into one extra double behind the partial sums
// Workgroups 0 .. kSparseRows - 1 take the edges OF dirty row b (a wavefront's lane = an edge); the others scan the column index
// for edges INTO a dirty row whose own row is clean.  Atomic adds: the order of the corrections of one row is not fixed - in a path
// that exists for a handful of rows per call.
struct PatchArgs {
    const uint32_t* hdr;
    const int32_t* rowptr; const int32_t* col; const int32_t* e2r;
    const float* X; const _Float16* x16; int32_t pitch;
    float* ef; const float* w; float* Y; uint32_t* efmax; double* dw_extra;
    int32_t N, Nc, D, row_off, mode; int64_t E;
};
__device__ __forceinline__ void patch_edge(const PatchArgs& a, int64_t e, int64_t r, int32_t c, bool c_dirty) {
    const float* xr = a.X + (r + a.row_off) * a.D;
    const float* xc = a.X + (int64_t)c * a.D;
    const _Float16* ir = a.x16 + (r + a.row_off) * a.pitch;
    const _Float16* ic = a.x16 + (int64_t)c * a.pitch;
    const float inv = pow2f(-scale_exp_from_bits(a.hdr[0]));
    float exact = 0.f, dimg = 0.f;
    for (int d = 0; d < a.D; ++d) { exact += round_rna10(xr[d]) * round_rna10(xc[d]); dimg += ((float)ir[d] * inv) * ((float)ic[d] * inv); }
    if (a.mode == 0) { a.ef[e] = exact; return; }
    const float w = a.w[0];
    if (a.mode == 1) {
        const float a_old = round_rna10(w * a.ef[e]), a_new = round_rna10(w * exact);
        a.ef[e] = exact;
        atomicMax(a.efmax, __float_as_uint(exact) & 0x7fffffffu);
        for (int d = 0; d < a.D; ++d) atomicAdd(&a.Y[r * a.D + d], a_new * round_rna10(xc[d]) - a_old * ((float)ic[d] * inv));
        return;
    }
    const float att = round_rna10(w * a.ef[e]);
    if (c_dirty)
        for (int d = 0; d < a.D; ++d) atomicAdd(&a.Y[r * a.D + d], att * (round_rna10(xc[d]) - (float)ic[d] * inv));
    atomicAdd(a.dw_extra, ((double)exact - (double)dimg) * (double)(float)c);
}
// the dense way through a wide matrix, inside the same launch (more than kSparseRows dirty rows: the MFMA kernel returned at once):
// plain fp32 in CSR order with the reference's operand rounding (as spmm_wide_fallback_kernel), a wavefront per row - every row's
// scores, then its aggregate from them, then (backward) its share of d_w into this workgroup's partial sum.
__device__ __forceinline__ void wide_dense_body(const PatchArgs& a, double* partial, int32_t npartial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = a.mode == 2 ? (int)min((uint32_t)gridDim.x, (uint32_t)npartial) : (int)gridDim.x;
    __shared__ double wsum[4];
    double acc = 0.0;
    uint32_t m = 0u;
    const float w = a.w ? a.w[0] : 1.0f;
    if ((int)blockIdx.x < nb)
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < a.N; row += (int64_t)nb * 4) {
        const float* xr = a.X + (row + a.row_off) * a.D;
        const int64_t e0 = a.rowptr[row], e1 = a.rowptr[row + 1];
        if (a.mode != 2 || a.dw_extra)
            for (int64_t e = e0; e < e1; ++e) {
                const float* xc = a.X + (int64_t)a.col[e] * a.D;
                float s = 0.f;
                for (int d = lane; d < a.D; d += 64) s += round_rna10(xr[d]) * round_rna10(xc[d]);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
                if (a.mode == 2) acc += (double)s * (double)(float)a.col[e];
                else if (lane == 0) { a.ef[e] = s; m = max(m, __float_as_uint(s) & 0x7fffffffu); }
            }
        if (a.mode == 0) continue;
        __threadfence_block();            // (mode 1: this wavefront reads back the scores its lane 0 has just written)
        for (int d = lane; d < a.D; d += 64) {
            float s = 0.f;
            for (int64_t e = e0; e < e1; ++e) s += round_rna10(w * a.ef[e]) * round_rna10(a.X[(int64_t)a.col[e] * a.D + d]);
            a.Y[row * a.D + d] = s;
        }
    }
    if (a.mode == 1 && a.efmax && lane == 0 && m) atomicMax(a.efmax, m);
    if (a.mode == 2) {
        if (lane == 0) wsum[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int k = (int)blockIdx.x; k < npartial; k += (int)gridDim.x) partial[k] = (k == (int)blockIdx.x && (int)blockIdx.x < nb) ? (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]) : 0.0;
    }
}
__global__ __launch_bounds__(256) void wide_patch_kernel(const PatchArgs a, double* partial, int32_t npartial) {
    if (!range_is_wide(a.hdr, 0)) return;
    if (a.hdr[8] > kSparseRows) { wide_dense_body(a, partial, npartial); return; }
    __shared__ int32_t dirty[kSparseRows];
    __shared__ int32_t ndirty;
    if (threadIdx.x == 0) {   // the list without duplicates, sorted (48 entries: insertion sort)
        const int n = (int)min(a.hdr[8], kSparseRows);
        int m = 0;
        for (int k = 0; k < n; ++k) {
            const int32_t v = (int32_t)a.hdr[16 + k];
            int j = 0;
            while (j < m && dirty[j] < v) ++j;
            if (j < m && dirty[j] == v) continue;
            for (int q = m; q > j; --q) dirty[q] = dirty[q - 1];
            dirty[j] = v; ++m;
        }
        ndirty = m;
    }
    __syncthreads();
    const int nd = ndirty;
    auto is_dirty = [&](int32_t x) { int lo = 0, hi = nd; while (lo < hi) { const int mid = (lo + hi) >> 1; if (dirty[mid] < x) lo = mid + 1; else hi = mid; } return lo < nd && dirty[lo] == x; };
    if (blockIdx.x < kSparseRows) {
        if ((int)blockIdx.x >= nd) return;
        const int64_t r = (int64_t)dirty[blockIdx.x] - a.row_off;       // the row of A this row of X belongs to
        if (r < 0 || r >= a.N) return;
        for (int64_t e = a.rowptr[r] + threadIdx.x; e < a.rowptr[r + 1]; e += blockDim.x) patch_edge(a, e, r, a.col[e], is_dirty(a.col[e]));
        return;
    }
    const int64_t first = (int64_t)(blockIdx.x - kSparseRows) * blockDim.x + threadIdx.x, step = (int64_t)(gridDim.x - kSparseRows) * blockDim.x;
    const int32_t lo_id = nd ? dirty[0] : 0, hi_id = nd ? dirty[nd - 1] : -1;
    for (int64_t e = first; e < a.E; e += step) {
        const int32_t c = a.col[e];
        if (c < lo_id || c > hi_id || !is_dirty(c)) continue;
        const int64_t r = a.e2r[e];
        if (r < 0 || r >= a.N || is_dirty((int32_t)(r + a.row_off))) continue;      // (its own row is dirty: the row's workgroup patches it)
        patch_edge(a, e, r, c, true);
    }
}
static hipError_t launch_wide_patch(const PatchArgs& a, hipStream_t stream, double* partial = nullptr, int32_t npartial = 0) {
    hipLaunchKernelGGL(wide_patch_kernel, dim3(kSparseRows + 1024), dim3(256), 0, stream, a, partial, npartial);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// launch tables
// ------------------------------------------------------------------------------------------
template <int NT, int WAVES, bool VAL>
static hipError_t launch_spmm_one(const SpmmArgs& args, int nwin, int nchunks, hipStream_t stream) {
    const size_t lds = (size_t)WAVES * TileWalker<NT, VAL>::WAVE_LDS + 4096;
    hipLaunchKernelGGL((spmm_kernel<NT, WAVES, VAL>), dim3((unsigned)nwin, (unsigned)nchunks), dim3(WAVES * 64), lds, stream, args);
    return hipGetLastError();
}

template <int WAVES, bool VAL>
static hipError_t launch_spmm_nt(int nt, const SpmmArgs& args, int nwin, int nchunks, hipStream_t stream) {
    switch (nt) {
        case 1: return launch_spmm_one<1, WAVES, VAL>(args, nwin, nchunks, stream);
        case 2: return launch_spmm_one<2, WAVES, VAL>(args, nwin, nchunks, stream);
        case 3: return launch_spmm_one<3, WAVES, VAL>(args, nwin, nchunks, stream);
        case 4: return launch_spmm_one<4, WAVES, VAL>(args, nwin, nchunks, stream);
        case 5: return launch_spmm_one<5, WAVES, VAL>(args, nwin, nchunks, stream);
        case 6: return launch_spmm_one<6, WAVES, VAL>(args, nwin, nchunks, stream);
        case 7: return launch_spmm_one<7, WAVES, VAL>(args, nwin, nchunks, stream);
        case 8: return launch_spmm_one<8, WAVES, VAL>(args, nwin, nchunks, stream);
        default: return hipErrorInvalidValue;
    }
}

static hipError_t launch_spmm_any(bool val, int waves, int nt, const SpmmArgs& args, int nwin, int nchunks, hipStream_t stream) {
    if (waves == 4) return val ? launch_spmm_nt<4, true>(nt, args, nwin, nchunks, stream) : launch_spmm_nt<4, false>(nt, args, nwin, nchunks, stream);
    return val ? launch_spmm_nt<1, true>(nt, args, nwin, nchunks, stream) : launch_spmm_nt<1, false>(nt, args, nwin, nchunks, stream);
}

// windows owned by one wavefront of the range-blocked kernel (accumulators: MAXW * NT * 4 registers)
static constexpr int blocked_maxw(int nt, bool val) { return (nt <= 4 && !val) ? 4 : 2; }

template <int NT, bool VAL>
static hipError_t launch_blocked_one(const SpmmBlockedArgs& args, int nwg, int nchunks, hipStream_t stream) {
    constexpr int MAXW = blocked_maxw(NT, VAL);
    const size_t lds = (size_t)4 * TileWalker<NT, VAL>::WAVE_LDS + 4096;
    hipLaunchKernelGGL((spmm_blocked_kernel<NT, MAXW, VAL>), dim3((unsigned)nwg, (unsigned)nchunks), dim3(256), lds, stream, args);
    return hipGetLastError();
}

static hipError_t launch_blocked_any(bool val, int nt, const SpmmBlockedArgs& args, int nwg, int nchunks, hipStream_t stream) {
#define TCGNN_BLK_CASE(n) case n: return val ? launch_blocked_one<n, true>(args, nwg, nchunks, stream) : launch_blocked_one<n, false>(args, nwg, nchunks, stream);
    switch (nt) {
        TCGNN_BLK_CASE(1) TCGNN_BLK_CASE(2) TCGNN_BLK_CASE(3) TCGNN_BLK_CASE(4)
        TCGNN_BLK_CASE(5) TCGNN_BLK_CASE(6) TCGNN_BLK_CASE(7) TCGNN_BLK_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef TCGNN_BLK_CASE
}

template <int WAVES, bool BLOCKED>
static hipError_t launch_sddmm_ks(int ks, const SddmmArgs& args, int nwg, hipStream_t stream) {
    const dim3 grid((unsigned)nwg), block(WAVES * 64);
    const size_t lds = (size_t)WAVES * sddmm_wave_lds(ks <= 4 ? ks : 1);
    switch (ks) {
        case 1: hipLaunchKernelGGL((sddmm_kernel<1, WAVES, BLOCKED>), grid, block, lds, stream, args); break;
        case 2: hipLaunchKernelGGL((sddmm_kernel<2, WAVES, BLOCKED>), grid, block, lds, stream, args); break;
        case 3: hipLaunchKernelGGL((sddmm_kernel<3, WAVES, BLOCKED>), grid, block, lds, stream, args); break;
        case 4: hipLaunchKernelGGL((sddmm_kernel<4, WAVES, BLOCKED>), grid, block, lds, stream, args); break;
        default: hipLaunchKernelGGL((sddmm_wide_kernel<WAVES>), grid, block, 0, stream, args); break;
    }
    return hipGetLastError();
}

static constexpr int kAgnnMaxW = 2;   // windows owned by a wavefront of the range-major fused kernel
template <int WAVES, bool BWD, int MAXW>
static hipError_t launch_agnn(int nt, const AgnnArgs& args, int nwg, hipStream_t stream) {
    const dim3 grid((unsigned)nwg), block(WAVES * 64);
    const size_t lds = (size_t)WAVES * agnn_wave_lds((nt + 1) / 2, BWD);
#define TCGNN_AGNN_CASE(n) case n: hipLaunchKernelGGL((agnn_kernel<n, WAVES, BWD, MAXW>), grid, block, lds, stream, args); break;
    switch (nt) {
        TCGNN_AGNN_CASE(1) TCGNN_AGNN_CASE(2) TCGNN_AGNN_CASE(3) TCGNN_AGNN_CASE(4)
        TCGNN_AGNN_CASE(5) TCGNN_AGNN_CASE(6) TCGNN_AGNN_CASE(7) TCGNN_AGNN_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef TCGNN_AGNN_CASE
    return hipGetLastError();
}

static int g_bucket_min_tiles = [] { const char* e = getenv("TCGNN_BUCKET_MIN_TILES"); return e ? atoi(e) : 2; }();   // tiles per (window, bucket) a bucket table needs
static int g_lds_auto = [] { const char* e = getenv("TCGNN_LDS_AUTO"); return e ? atoi(e) : 1; }();
static int g_lds_dbg = [] { const char* e = getenv("TCGNN_LDS_DBG"); return e ? atoi(e) : 0; }();
static int g_lds_fill_quota = [] { const char* e = getenv("TCGNN_LDS_FILL_QUOTA"); return e ? atoi(e) : 1; }();   // A/B aid (tcgnn_lds_flat.inc)
static int g_spmm_mode = [] { const char* e = getenv("TCGNN_SPMM_MODE"); return e ? atoi(e) : 0; }();
static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static constexpr size_t kBlockedMinBytes = 6u << 20;   // below this X16 is (nearly) L2-resident anyway
static constexpr size_t kRangeTargetBytes = 1u << 20;  // X16 bytes per column range: wavefronts drift by a range or two
                                                        // and the 4 MB L2 also streams metadata (sweep: 0.5-1.5 MB best at D=64)
static constexpr size_t kHdrBytes = 256;

// Row pitch of the fp16 image in halves: a gathered row should touch as few 128-byte lines as
// possible (a 96-byte row at pitch 96 straddles two lines three times out of four: D = 41..48 ran
// slower than D = 64), so rows up to 128 B are padded to a power of two and longer ones to whole lines.
static int x16_pitch(int dpad) {
    const int bytes = dpad * 2;
    if (bytes > 128) return ((bytes + 127) / 128 * 128) / 2;   // whole lines (D = 602: 1280 B, not the 2048 B of the next power of two)
    int p = 32;
    while (p < bytes) p <<= 1;   // power of two: 32 B .. 128 B inside one line
    return p / 2;
}
// The gather walks address X16 through a structured buffer descriptor whose record stride (the row pitch in bytes) is a
// 14-bit field: a wider row would wrap it (stride 0 + the swizzle bit set) and every gather would silently read the wrong row.
static constexpr int kMaxStructStride = 16383;
// ... and whose index * stride + offset is formed in 32 bits: an image of 4 GB or more (a papers100M-sized shard: 111 M rows of
// 128 B) wraps.  The kernels then form 64-bit lane addresses instead (one v_mad_u64_u32 per gathered piece).
static int32_t image_is_big(int32_t rows, int pitch_halves) { return ((uint64_t)rows + 1) * (uint64_t)pitch_halves * 2u >= (1ull << 32) ? 1 : 0; }
static bool pitch_fits_descriptor(int D) { return x16_pitch(round_up(D, 16)) * 2 <= kMaxStructStride; }

static size_t workspace_bytes_for(int32_t N, int32_t D) {
    const size_t dpad = (size_t)x16_pitch(round_up(D, 16));
    const size_t body = ((size_t)N + 1) * dpad * sizeof(_Float16);
    return kHdrBytes + ((body + 255) / 256) * 256;
}
// the fused AGNN backward keeps one double per workgroup (per window; the XCD-sliced walk: per slice and four windows) behind the fp16 image
static constexpr int kAgnnMaxSlices = 16;   // two rounds of eight (= XCDs)
static size_t agnn_partial_bytes(const tcgnn_plan* plan) {
    const size_t slots = std::max<size_t>((size_t)std::max(plan->nw_eff, 1), (size_t)kAgnnMaxSlices * (((size_t)std::max(plan->nw_eff, 1) + 3) / 4));
    return (slots * sizeof(double) + 255) / 256 * 256;
}

// enqueue absmax(X) [+ absmax(val)] + convert; returns the fp16 image pointer
// The range-blocked walks bind up to 4 windows to one persistent wavefront for the whole launch: a hub window (skewed
// degrees) then holds its wavefront far beyond the others (measured on a Reddit-sized graph with a 93 k-degree hub:
// 2.48 ms against 1.60 ms for the per-window walk, which spreads a window over 4 wavefronts).  Automatic mode only takes
// them when the longest window is within 8x the mean.
static bool has_locality(const tcgnn_plan* plan) { return plan->near_frac > 0.5; }
static bool windows_balanced(const tcgnn_plan* plan) {
    return plan->nw_eff > 0 && plan->max_wb * (int64_t)plan->nw_eff <= 8 * std::max<int64_t>(plan->total_wb, 1);
}
// ... and when the bucket table can cut the image into ranges an XCD's 4 MB L2 holds (ogbn-products at D = 128: 8 buckets
// of 78 MB - the range-major SDDMM then only pays for its bookkeeping: 5.70 ms against 4.93 ms per-window)
// ... and when there are enough windows for two workgroups of persistent wavefronts per CU at 4 windows each: with 3750 /
// 6250 windows the range-blocked walk left the chip a quarter full (0.33 / 1.03 ms against 0.13 / 0.66 ms per-window).
static bool ranges_fit_l2(const tcgnn_plan* plan, size_t x16_bytes) {
    return plan->nbuckets > 0 && x16_bytes / (size_t)plan->nbuckets <= ((size_t)8 << 20) && plan->nw_eff >= 32 * plan->num_cus;
}
// The fused AGNN kernel's walks beside the per-window one (agnn_kernel), for graphs whose numbering carries no locality of its own and
// whose windows are alike, when the fp16 image does not fit an XCD's 4 MB L2 but an eighth of it does:
//   XCD-sliced  - nslices addends of Y in the workspace and a pass that sums them;
//   range-major - persistent wavefronts owning two windows each (no addends; more registers).
// Measured on the Reddit shape (tools/bench_agnn.py, forward / backward ms; r03 with whole-line gathers at D = 64):
//   D = 16 (7.4 MB)  per-window 1.13 / 1.43   sliced 1.06 / 1.28   range-major 1.19 / 1.59
//   D = 32 (14.9 MB) per-window 1.46 / 1.65   sliced 1.20 / 1.38   range-major 1.26 / 1.65
//   D = 64 (29.8 MB) per-window 1.74 / 1.77   sliced 1.53 / 1.60   range-major 1.45-1.48 / 1.78   (sixteen slices in two rounds 1.81 / 1.85)
//   D = 128 (59.6 MB) per-window 3.48 / 3.53  sliced 2.61-2.67 / 2.71-2.73   range-major 2.68-2.73 / 2.93-2.95   (slices of 7.4 MB: they do
//                     not stay in a 4 MB L2, but an XCD that is asked for an eighth of the image still hits more often than one asked for all of it)
// so: sliced in both directions up to 16 MB; from there to 64 MB range-major forward (within 2 % of sliced, no addends) and sliced backward.
// What these walks are bound by is the memory system's throughput at their hit rate, not by what a wavefront has in flight nor by
// its instruction count (r03, measured on the sliced walk at D = 64): a quarter fewer VALU instructions per tile (103 -> 71 in the
// forward tile block) changed nothing; a second tile buffer with the gather running two tiles ahead (counted vmcnt, no extra
// registers) moved forward 1.53 -> 1.53 and backward 1.60 -> 1.57 and was taken out again; four wavefronts per SIMD instead of
// three (forward kernel squeezed from 130 to 128 registers, 12 bytes of scratch) 1.53 -> 1.43-1.45 sliced but 1.73 -> 1.79-1.87 per-window
// (more wavefronts thrash the L2 harder) - level with range-major's 1.45-1.48, so not kept either.
// TCGNN_AGNN_SLICED (read per call: tests switch it): 0 per-window only, 1 the rule above, 2 sliced whenever possible, 16 two rounds.
static constexpr size_t kAgnnSliceBytes = (size_t)4 << 20;
enum { kAgnnPerWindow = 0, kAgnnSliced = 1, kAgnnRangeMajor = 2 };
static int agnn_walk(const tcgnn_plan* plan, int32_t D, bool bwd, int* nslices_out) {
    *nslices_out = 0;
    const char* const env = getenv("TCGNN_AGNN_SLICED");
    const int knob = env ? atoi(env) : 1;
    if (!knob || plan->waves != 4 || plan->nbuckets < 8 || plan->nw_eff < 1 || plan->nbuckets % kAgnnXcds) return kAgnnPerWindow;
    const int pitch = x16_pitch(round_up(D, 16));
    if (image_is_big(plan->Nc, pitch)) return kAgnnPerWindow;
    const size_t x16_bytes = ((size_t)plan->Nc + 1) * pitch * sizeof(_Float16);
    if (knob >= 2) { *nslices_out = (knob == 16 && plan->nbuckets % 16 == 0) ? 16 : kAgnnXcds; return kAgnnSliced; }   // (forced)
    if (!(x16_bytes > kBlockedMinBytes && x16_bytes <= 2 * (size_t)kAgnnXcds * kAgnnSliceBytes && plan->nw_eff >= 8 * plan->num_cus &&
          windows_balanced(plan) && !has_locality(plan))) return kAgnnPerWindow;
    if (x16_bytes > (size_t)kAgnnXcds * kAgnnSliceBytes && !bwd) return kAgnnRangeMajor;             // 32 - 64 MB: forward
    if (!bwd && x16_bytes > (size_t)kAgnnXcds * (kAgnnSliceBytes / 2)) return kAgnnRangeMajor;       // 16 - 32 MB: forward
    // (the sliced walk wants every window's tiles spread evenly over the slices: workgroups are handed to the XCDs round-robin and
    //  in order, so where a window has most of its tiles in one slice - the calibrated SBM graph: 22.5 % of the edges inside the
    //  window's own community, near_frac 0.3 - the XCD of that slice holds the others up: backward 1.81 -> 2.40 ms there)
    if (plan->near_frac > 0.2) return kAgnnPerWindow;
    *nslices_out = kAgnnXcds;
    return kAgnnSliced;
}
// (the workspace is sized for whichever direction slices)
static int agnn_slices(const tcgnn_plan* plan, int32_t D) {
    int nf = 0, nb = 0;
    (void)agnn_walk(plan, D, false, &nf); (void)agnn_walk(plan, D, true, &nb);
    return std::max(nf, nb);
}
static size_t agnn_slice_bytes(const tcgnn_plan* plan, int32_t D) {
    return ((size_t)agnn_slices(plan, D) * (size_t)plan->N * D * sizeof(float) + 255) / 256 * 256;
}

// ---- range guard parameters (range_is_wide): cap = how many lost-precision terms one result can collect at most - the longest row
// of the graph (SpMM) or 2 D (SDDMM / fused AGNN) - and the power of max|X| in the error bound.  cap 0 = guard off
// (tcgnn_set_range_guard(0), TCGNN_RANGE_GUARD=0).
static int g_range_guard = [] { const char* e = getenv("TCGNN_RANGE_GUARD"); return e ? atoi(e) : 2; }();   // (r04: every operator - the usual wide input costs one patch launch)
struct Guard { uint32_t cap, pow; };
static Guard guard_spmm(const tcgnn_plan* p) { return {g_range_guard ? (uint32_t)std::max(p->max_degree, 1) : 0u, 1u}; }
// (level 1, the default: the aggregation operators - binary and edge-valued SpMM, the fused dense update - whose bound is linear in
//  max|X| and which a training epoch never reaches; level 2 adds SDDMM and the fused AGNN pair, whose bound is QUADRATIC in max|X|:
//  an AGNN epoch of the reference's unscaled recipe crosses 2^14.5 with a single lost element now and then, and each such call
//  costs ~25 ms in the CSR fallbacks against 2 ms - so those two answer to the documented bound unless asked to be strict)
static Guard guard_sddmm(int D) { return {g_range_guard >= 2 ? (uint32_t)(2 * std::max(D, 1)) : 0u, 2u}; }

// ldx > 0: X (and the gate) is a column block of a wider row-major matrix with that row stride; the scale words in the
// header were then computed over the WHOLE matrix by the caller (block_of_wider = true: no memset, no absmax pass here), so
// every block is rounded exactly as the undivided call would round it.
static int stage_features(const tcgnn_plan* plan, const float* d_X, const float* d_val, int32_t D,
                          void* ws, size_t ws_bytes, hipStream_t stream, const uint32_t** hdr_out,
                          const _Float16** x16_out, int* dpad_out, int* pitch_out, bool planar = false, const float* d_gate = nullptr,
                          int64_t ldx = 0, bool block_of_wider = false, const uint32_t* hdr_from = nullptr, const Guard* guard = nullptr) {
    const Guard gx = guard ? *guard : guard_spmm(plan);   // (the binary SpMM's bound unless the caller's operator has its own)
    // hdr_from: the scale words of an image of the same matrix staged a moment ago (the planar one of a plan with a cold remainder):
    // copied instead of recomputed, so both images are rounded with the same scale without a second pass over X
    const size_t need = workspace_bytes_for(plan->Nc, D);
    if (!ws || ws_bytes < need || (reinterpret_cast<uintptr_t>(ws) & 255))
        return fail(TCGNN_ERR_WORKSPACE, "workspace: need %zu bytes 256-aligned, got %zu at %p", need, ws_bytes, ws);
    uint32_t* hdr = static_cast<uint32_t*>(ws);
    _Float16* x16 = reinterpret_cast<_Float16*>(static_cast<char*>(ws) + kHdrBytes);
    const int dpad = round_up(D, 16);
    const int pitch = x16_pitch(dpad);
    if (hdr_from) { HIP_TRY(hipMemcpyAsync(hdr, hdr_from, 32, hipMemcpyDeviceToDevice, stream)); block_of_wider = true; }
    else if (!block_of_wider) HIP_TRY(hipMemsetAsync(hdr, 0, 64, stream));   // (words 0 .. 7: range words; 8: dirty-row count)
    const int64_t nx = block_of_wider ? 0 : (int64_t)plan->Nc * D;
    if (nx > 0) {
        const int grid = absmax_grid(nx);
        if (d_gate) hipLaunchKernelGGL(absmax_gated_kernel, dim3(grid), dim3(kAbsmaxThreads), 0, stream, d_X, d_gate, nx, hdr, hdr + 2, gx.cap, gx.pow);
        else hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(kAbsmaxThreads), 0, stream, d_X, nx, hdr, hdr + 2, gx.cap, gx.pow);
    }
    if (d_val && plan->E > 0 && !block_of_wider) {
        const int grid = absmax_grid(plan->E);
        hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(kAbsmaxThreads), 0, stream, d_val, plan->E, hdr + 1, hdr + 3, guard_spmm(plan).cap, 0u);
    }
    const int64_t chunks = ((int64_t)plan->Nc + 1) * (dpad / 8);
    const unsigned cgrid = (unsigned)((chunks + 255) / 256);
    uint32_t* const tiny = hdr_from ? nullptr : hdr + 6;   // (a second image of the same matrix: its elements are counted already)
    const bool vec = (D % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_X) & 15) == 0);
    if (planar) {   // [dpad / 16 planes][Nc + 1][16 halves] for the LDS-resident range kernel (same chunk count: no pitch padding)
        if (vec && D % 16 == 0 && (!d_gate || (reinterpret_cast<uintptr_t>(d_gate) & 15) == 0)) {
            const int64_t threads = ((int64_t)plan->Nc + 1) * (D / 4);
            hipLaunchKernelGGL(convert_planar_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, d_X, plan->Nc, D, x16, hdr, d_gate, tiny);
        } else if (vec) hipLaunchKernelGGL((convert_planar_kernel<true>), dim3(cgrid), dim3(256), 0, stream, d_X, plan->Nc, D, dpad / 16, x16, hdr, d_gate, tiny);
        else if ((size_t)D * 64 * sizeof(float) <= 48 * 1024)
            hipLaunchKernelGGL(convert_planar_tiled_kernel, dim3((unsigned)(((int64_t)plan->Nc + 1 + 63) / 64)), dim3(256), (size_t)D * 64 * sizeof(float), stream,
                               d_X, plan->Nc, D, dpad / 16, x16, hdr, d_gate, tiny);
        else     hipLaunchKernelGGL((convert_planar_kernel<false>), dim3(cgrid), dim3(256), 0, stream, d_X, plan->Nc, D, dpad / 16, x16, hdr, d_gate, tiny);
    } else {
        uint32_t* const dirty = (tiny && gx.pow == 2u && gx.cap) ? hdr : nullptr;   // (SDDMM / fused AGNN at guard level 2: dirty rows for wide_patch_kernel)
        if (vec) hipLaunchKernelGGL((convert_kernel<true>), dim3(cgrid), dim3(256), 0, stream, d_X, plan->Nc, D, dpad, pitch, x16, hdr, d_gate, ldx, tiny, dirty);
        else     hipLaunchKernelGGL((convert_kernel<false>), dim3(cgrid), dim3(256), 0, stream, d_X, plan->Nc, D, dpad, pitch, x16, hdr, d_gate, ldx, tiny, dirty);
    }
    HIP_TRY(hipGetLastError());
    *hdr_out = hdr; *x16_out = x16; *dpad_out = dpad; *pitch_out = pitch;
    return TCGNN_OK;
}

// Cell stream of the LDS-resident column-range SpMM: per (workgroup, range, wavefront, window slot) the window's
// condensed columns inside the range, re-tiled 32 to a tile.  Built from the packed tile stream (cols / mask).
static int g_lds_maxw = [] { const char* e = getenv("TCGNN_LDS_MAXW"); const int v = e ? atoi(e) : 0; return (v == 4 || v == 8) ? v : 0; }();   // 0: by width
// Passes of the LDS-resident kernel over a matrix of dpad columns.  Whole 64-column chunks go as two 32-column passes of the
// 8-windows-per-wavefront layout (half the workgroups stream each plane pair, 760-row ranges: Reddit D = 64 0.56 vs 0.79 ms),
// what is left over (1-3 planes) as one pass of the 4-window layout.  TCGNN_LDS_MAXW = 4 / 8 forces one layout for every
// pass (tests, timing).
struct LdsPass { int maxw, nt, chunk0, nchunks; };
static int lds_passes(int dpad, LdsPass (&passes)[2]) {
    int n = 0;
    if (g_lds_maxw) {
        const int cd = lds_chunk_dims(g_lds_maxw);
        if (dpad / cd) passes[n++] = {g_lds_maxw, cd / 16, 0, dpad / cd};
        if (dpad % cd) passes[n++] = {g_lds_maxw, (dpad % cd) / 16, dpad / cd, 1};
    } else {
        // (r03: a remainder of THREE planes - Reddit's 41 classes - goes as one more pair of 32-column chunks of the 8-window layout,
        //  the fourth plane lying beyond the matrix and filled with zeros: 0.45 ms against 0.51 for the 3-plane pass of the 4-window
        //  layout, which streams 22 MB per CU and has no room for the in-kernel cold remainder)
        const int full = dpad / 64, rem = (dpad % 64) / 16;
        if (full || rem == 3) passes[n++] = {kLdsMaxW2, 2, 0, 2 * (full + (rem == 3 ? 1 : 0))};
        if (rem && rem != 3) passes[n++] = {kLdsMaxW, rem, full, 1};
    }
    return n;
}
// workgroups of one pass: enough to hold every window, spread over every CU a pass can have (with 8 windows per wavefront a
// 64-column chunk takes two passes, hence half the CUs each)
static bool lds_has_hubs(const tcgnn_plan* p) {
    int64_t mx = 0;
    for (int w = 0; w < p->nw_eff; ++w) mx = std::max<int64_t>(mx, p->h_bp[(size_t)w]);
    return mx * p->nw_eff > 4 * std::max<int64_t>(p->tc_blocks, 1);
}
// Placements of a cell stream (lds_place_windows): contiguous weight-balanced blocks per workgroup (locality: the hot / cold split,
// wavefronts of a workgroup busy in the same ranges), the same with hub windows split, or longest-first over the whole graph with
// hub windows split.  Graphs without hubs take the first; graphs with hubs build the count tables of the other two and keep the
// one whose estimated time is lower (build_lds_cells).  TCGNN_LDS_PLACE=local|localsplit|global forces one.
enum { kPlaceLocal = 0, kPlaceLocalSplit = 1, kPlaceGlobal = 2 };
static int lds_place_forced() {
    const char* env = getenv("TCGNN_LDS_PLACE");
    if (!env) return -1;
    return !strcmp(env, "global") ? kPlaceGlobal : (!strcmp(env, "localsplit") ? kPlaceLocalSplit : kPlaceLocal);
}
static int lds_buf_rows_for_maxw(int maxw) { return maxw == kLdsMaxW2 ? 768 : 512; }   // (the shortest ranges of the layout: the finest spread)
// weight of a window = the tiles it is likely to cost the LDS-resident walk: its condensed columns spread over the column
// ranges (a cell with a handful of columns still costs a whole tile step), not the column count alone
static double lds_window_weight(const tcgnn_plan* p, int w, double nranges_d) {
    const double cols = 8.0 * std::max(p->h_bp[(size_t)w], 1);
    return std::ceil(cols / 32.0 + nranges_d * (1.0 - std::exp(-cols / nranges_d)));
}
static int lds_workgroups_unsplit(const tcgnn_plan* p, int maxw, int extra_slots) {
    const int per_wg = kLdsWaves * maxw;
    const int64_t slots = (int64_t)p->nw_eff + extra_slots;
    int nwg = (int)((slots + per_wg - 1) / per_wg);
    const int cu_target = maxw == kLdsMaxW2 ? std::max(1, p->num_cus / 2) : p->num_cus;
    if (nwg < cu_target) nwg = std::max(nwg, std::min(cu_target, (p->nw_eff + kLdsWaves - 1) / kLdsWaves));
    return nwg;
}
// Hub windows: a wavefront owns whole windows, so a window whose tiles exceed a wavefront's fair share of the workgroup's work
// is the critical path of every range (R-MAT, Reddit shape: the window of the sixteen top hubs holds 3.6 wavefront shares).
// Such a window is SPLIT: k wavefronts of one workgroup each take every k-th run of its tiles in every range and their partial
// sums are added through LDS, in a fixed order, at the end of the kernel.  parts[w] = k (1: whole).  Only on graphs that take the
// graphs with hubs (lds_has_hubs); TCGNN_LDS_SPLIT=0 switches it off.
static constexpr int kLdsMaxParts = 8, kLdsMaxFollowers = 32;   // (followers of a workgroup: 32 x NT KB of LDS scratch)
static int lds_split_parts(const tcgnn_plan* p, int maxw, std::vector<uint8_t>* parts, const std::vector<double>* exact = nullptr) {
    static const int enabled = [] { const char* e = getenv("TCGNN_LDS_SPLIT"); return e ? atoi(e) : 1; }();
    const int nw = p->nw_eff;
    if (parts) parts->assign((size_t)nw, 1);
    if (!enabled || nw <= 0 || !lds_has_hubs(p)) return 0;
    const double nranges_d = std::max(1.0, std::ceil((double)p->Nc / (lds_buf_rows_for_maxw(maxw) - 8)));
    auto weight = [&](int w) { return exact ? (*exact)[(size_t)w] : lds_window_weight(p, w, nranges_d); };
    double total = 0;
    for (int w = 0; w < nw; ++w) total += weight(w);
    const double share = total / ((double)lds_workgroups_unsplit(p, maxw, 0) * kLdsWaves);
    int extra = 0;
    for (int w = 0; w < nw; ++w) {
        const double wt = weight(w);
        if (wt <= 1.5 * share) continue;
        const int k = (int)std::min<double>(kLdsMaxParts, std::ceil(wt / share));
        if (k < 2) continue;
        if (parts) (*parts)[(size_t)w] = (uint8_t)k;
        extra += k - 1;
    }
    return extra;
}
static int lds_workgroups(const tcgnn_plan* p, int maxw) {
    const int which = maxw == kLdsMaxW2 ? 1 : 0;
    int extra = p->lds_extra[which].load(std::memory_order_relaxed);
    if (extra < 0) { extra = lds_split_parts(p, maxw, nullptr); p->lds_extra[which].store(extra, std::memory_order_relaxed); }
    return lds_workgroups_unsplit(p, maxw, extra);
}

// Kernel-time models behind the automatic choice between the LDS-resident kernel and the gather walks, microseconds on
// MI355X (tools/check_lds_threshold.py: nine graphs x two widths; the estimates land within ~15 % of the measured times).
// The LDS kernel's time follows the column ranges it walks, almost whatever the edge count: per range a fixed part (barrier,
// DMA issue, metadata; grows with the bytes a range streams) plus ~0.135 us per tile a wavefront multiplies; workgroups
// beyond one per CU run in further rounds.  The gather walks' time follows the edge count.
static double lds_estimate_us(const tcgnn_plan* p, int dpad) {
    LdsPass passes[2];
    const int np = lds_passes(dpad, passes);
    const double cols_per_window = 32.0 * (double)p->total_wb / std::max(p->nw_eff, 1);
    double t = 25.0;
    for (int i = 0; i < np; ++i) {
        const int maxw = passes[i].maxw, nt = passes[i].nt;
        const int rows = lds_stream_buf_rows(lds_stream_of(nt, maxw)) - 8;
        const double nranges = std::ceil((double)p->Nc / rows);
        const int nwg = lds_workgroups(p, maxw);
        const double rounds = std::ceil((double)nwg * passes[i].nchunks / std::max(p->num_cus, 1));
        const double c = cols_per_window / nranges;                                   // distinct columns of a cell
        const double tiles_per_cell = c <= 24.0 ? 1.0 - std::exp(-c) : c / 32.0 + 0.5;
        const double windows_per_wave = (double)p->nw_eff / ((double)nwg * kLdsWaves);
        static const double fixed4[4] = {0.47, 0.60, 0.89, 1.16};
        const double a = maxw == kLdsMaxW2 ? 0.80 : fixed4[std::min(nt, 4) - 1];
        t += rounds * nranges * (a + 0.135 * windows_per_wave * tiles_per_cell);
    }
    return t;
}
static double gather_estimate_us(const tcgnn_plan* p, int dpad) {
    const double image = ((double)p->Nc + 1) * x16_pitch(dpad) * 2.0;
    double ps;   // picoseconds per edge
    if (image <= (double)kBlockedMinBytes) ps = 4.6 + std::max(0, dpad - 16) * (1.8 / 48.0);        // L2-resident image, per-window walk
    else if (dpad <= 32) ps = 8.5;
    else if (dpad <= 64) ps = 8.5 + (dpad - 32) * (1.1 / 32.0);
    else ps = 9.6 + (dpad - 64) * (6.9 / 64.0);
    // the gathered image leaves first the L2s, then the Infinity Cache (row shards of the multi-GPU workload: 60 / 119 / 239 MB
    // -> 10.0 / 12.4 / 14.7 ps per edge at 64 columns)
    if (image > 45.0e6) ps *= std::pow(image / 45.0e6, 0.3);
    return 20.0 + (double)p->E * ps * 1e-6;
}
// automatic mode: the LDS-resident kernel when its estimate is clearly the lower one (decided once per plan and width)
static bool lds_chosen(const tcgnn_plan* p, int dpad) {
    if (!g_lds_auto || p->nw_eff <= 0 || p->total_wb <= 0) return false;
    const int k = dpad / 16;
    if (k <= 64 && p->lds_choice[k] >= 0) return p->lds_choice[k] != 0;
    LdsPass passes[2];
    const int np = lds_passes(dpad, passes);
    double cells = 0;   // cell-table entries of the streams this width needs (host scan + device memory)
    for (int i = 0; i < np; ++i) {
        const int rows = lds_stream_buf_rows(lds_stream_of(passes[i].nt, passes[i].maxw)) - 8;
        cells += (double)lds_workgroups(p, passes[i].maxw) * kLdsWaves * passes[i].maxw * std::ceil((double)p->Nc / rows);
    }
    const bool yes = cells < 2.0e8 && lds_estimate_us(p, dpad) <= 0.95 * gather_estimate_us(p, dpad);
    if (k <= 64) p->lds_choice[k] = yes ? 1 : 0;
    return yes;
}

// Window slots of a cell stream (order[cell_position(wg, wave, j)] = window id or -1): the windows, in their own order, are cut into
// nwg contiguous blocks of about equal weight (blockPartition = condensed columns) and at most 16 x maxw windows; inside a block
// they go heaviest-first to the least loaded wavefront that still has a free slot.
// One workgroup's windows (heaviest first; a split window's parts weigh wt / k each and are placed when it comes up) go to the
// least loaded wavefront with a free slot, the parts of one window to different wavefronts.  parts_out (if sized) receives
// part | parts << 8 | scratch index << 16 for the slots of split windows: scratch index of a follower = its own, of part 0 = its first follower's.
static bool lds_deal_workgroup(int g, int maxw, const std::vector<int32_t>& items, const std::vector<uint8_t>& k, const std::vector<double>& wt,
                               std::vector<int32_t>& order, std::vector<uint32_t>& parts_out, int& nsplit) {
    double load[kLdsWaves] = {0};
    int used[kLdsWaves] = {0};
    uint32_t next_fol = 0;
    for (const int32_t w : items) {
        const int kk = k.empty() ? 1 : k[(size_t)w];
        uint32_t taken = 0u;                         // wavefronts that hold a part of this window
        const uint32_t fol0 = next_fol;
        for (int part = 0; part < kk; ++part) {
            int best = -1;
            for (int v = 0; v < kLdsWaves; ++v)
                if (used[v] < maxw && !((taken >> v) & 1u) && (best < 0 || load[v] < load[best])) best = v;
            if (best < 0)   // (every free slot sits on a wavefront that already holds a part of this window)
                for (int v = 0; v < kLdsWaves; ++v) if (used[v] < maxw && (best < 0 || load[v] < load[best])) best = v;
            if (best < 0) return false;   // more items than slots: the caller's accounting is off - it falls back to a placement without parts
            const size_t pos = (size_t)cell_position(g, best, used[best], maxw);
            order[pos] = w;
            if (kk > 1) parts_out[pos] = (uint32_t)part | ((uint32_t)kk << 8) | ((part == 0 ? fol0 : fol0 + (uint32_t)part - 1u) << 16);
            taken |= 1u << best;
            load[best] += wt[(size_t)w] / kk;
            ++used[best];
        }
        if (kk > 1) { next_fol += (uint32_t)kk - 1u; ++nsplit; }
    }
    return true;
}
static bool lds_place_windows(const tcgnn_plan* p, int nwg, int maxw, std::vector<int32_t>& order, std::vector<uint32_t>& parts_out, int& nsplit,
                              const std::vector<double>& exact, int mode) {
    const int nw = p->nw_eff, cap = kLdsWaves * maxw;
    order.assign((size_t)nwg * cap, -1);
    parts_out.clear();
    nsplit = 0;
    // Hub windows (power-law graphs numbered by degree) sit next to each other: a contiguous block of them fills a few wavefronts of
    // its workgroup and leaves the rest idle, so the workgroup runs several times longer than the mean (R-MAT, Reddit shape: 1.20 ms
    // against 0.85 ms).  When the heaviest window is far above the mean the windows are instead dealt heaviest first, boustrophedon-wise
    // over workgroups and wavefronts - every workgroup gets one hub and a share of the light windows.  TCGNN_LDS_PLACE=global|local forces it.
    if (mode == kPlaceGlobal) {
        std::vector<uint8_t> k;
        const int extra = lds_split_parts(p, maxw, &k, &exact);
        const char* const place_env = getenv("TCGNN_LDS_PLACE_DEAL");   // A/B aid: the r01 boustrophedon deal when no window is split
        if (extra == 0 && place_env && atoi(place_env) > 0) {
            std::vector<int32_t> idx((size_t)nw);
            std::iota(idx.begin(), idx.end(), 0);
            std::stable_sort(idx.begin(), idx.end(), [&](int32_t x, int32_t y) { return p->h_bp[(size_t)x] > p->h_bp[(size_t)y]; });
            for (int q = 0; q < nw; ++q) {
                const int row = q / nwg, c = q % nwg;
                const int wg = (row & 1) ? nwg - 1 - c : c, j = row / kLdsWaves, wv = row % kLdsWaves;
                order[(size_t)cell_position(wg, (j & 1) ? kLdsWaves - 1 - wv : wv, j, maxw)] = idx[(size_t)q];
            }
            return true;
        }
        // With split windows: longest-processing-time placement.  Windows (a split one with all its parts) go heaviest first to the
        // least loaded workgroup that has the slots; inside a workgroup the items (whole windows and parts) go heaviest first to
        // the least loaded wavefront with a free slot, the parts of one window to different wavefronts.
        const std::vector<double>& wt = exact;
        std::vector<int32_t> idx((size_t)nw);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int32_t x, int32_t y) { return wt[(size_t)x] > wt[(size_t)y]; });
        std::vector<double> wg_load((size_t)nwg, 0.0);
        std::vector<int> wg_free((size_t)nwg, cap), wg_fol((size_t)nwg, 0);
        std::vector<std::vector<int32_t>> wg_items((size_t)nwg);
        for (int q = 0; q < nw; ++q) {
            const int w = idx[(size_t)q];
            int kk = k[(size_t)w];
            int best = -1;
            for (int pass = 0; pass < 2 && best < 0; ++pass) {   // (second pass: the window whole, wherever one slot is free)
                if (pass == 1) { kk = 1; k[(size_t)w] = 1; }
                for (int g = 0; g < nwg; ++g)
                    if (wg_free[(size_t)g] >= kk && wg_fol[(size_t)g] + kk - 1 <= kLdsMaxFollowers && (best < 0 || wg_load[(size_t)g] < wg_load[(size_t)best])) best = g;
            }
            if (best < 0) return false;
            wg_load[(size_t)best] += wt[(size_t)w];
            wg_free[(size_t)best] -= kk;
            wg_fol[(size_t)best] += kk - 1;
            wg_items[(size_t)best].push_back(w);
        }
        if (extra > 0) parts_out.assign((size_t)nwg * cap, 0u); else k.clear();
        for (int g = 0; g < nwg; ++g)
            if (!lds_deal_workgroup(g, maxw, wg_items[(size_t)g], k, wt, order, parts_out, nsplit)) return false;
        if (nsplit == 0) parts_out.clear();
        return true;
    }
    // weight of a window = the tiles it is likely to cost the LDS-resident walk: its condensed columns spread over the column
    // ranges (a cell with a handful of columns still costs a whole tile step), not the column count alone
    const std::vector<double>& wt = exact;
    double total = 0;
    for (int w = 0; w < nw; ++w) total += wt[(size_t)w];
    std::vector<uint8_t> k;
    const int extra = mode == kPlaceLocalSplit ? lds_split_parts(p, maxw, &k, &exact) : 0;
    if (extra > 0) {
        // the stream has nwg * cap slots (lds_workgroups counted the parts with the modelled weights; the exact ones may ask for
        // more): take parts back, from the most divided windows first, until everything fits
        int64_t slots = 0;
        for (int w = 0; w < nw; ++w) slots += k[(size_t)w];
        for (int level = kLdsMaxParts; slots > (int64_t)nwg * cap && level > 1; --level)
            for (int w = 0; w < nw && slots > (int64_t)nwg * cap; ++w)
                if (k[(size_t)w] == level) { --k[(size_t)w]; --slots; }
        parts_out.assign((size_t)nwg * cap, 0u);
    } else k.clear();
    std::vector<int64_t> slots_after((size_t)nw + 1, 0);   // slots window w and the windows after it need (an upper bound: parts may still be taken back)
    for (int w = nw - 1; w >= 0; --w) slots_after[(size_t)w] = slots_after[(size_t)w + 1] + (k.empty() ? 1 : k[(size_t)w]);
    std::vector<int32_t> blk;   // windows of the block being dealt
    double acc = 0;
    int wg = 0, blk_slots = 0, blk_fol = 0;
    bool ok = true;
    static const int natural = [] { const char* e = getenv("TCGNN_LDS_DEAL_NATURAL"); return e ? atoi(e) : 1; }();
    auto deal = [&]() {
        // Even windows are dealt in their own order (least loaded wavefront first = round-robin): a block that spans two communities
        // then gives every wavefront its share of both, and in a range of either community all sixteen are busy.  Sorting by
        // weight first is for blocks with windows far above the others.
        double mx = 0, sum = 0;
        for (const int32_t w : blk) { mx = std::max(mx, wt[(size_t)w]); sum += wt[(size_t)w]; }
        if (!natural) std::stable_sort(blk.begin(), blk.end(), [&](int32_t x, int32_t y) { return wt[(size_t)x] > wt[(size_t)y]; });
        else if (mx * (double)blk.size() > 1.5 * sum) {   // the windows far above the mean first, heaviest first; the others keep their order
            const double heavy = 2.0 * sum / (double)blk.size();
            auto mid = std::stable_partition(blk.begin(), blk.end(), [&](int32_t x) { return wt[(size_t)x] > heavy; });
            std::stable_sort(blk.begin(), mid, [&](int32_t x, int32_t y) { return wt[(size_t)x] > wt[(size_t)y]; });
        }
        ok = lds_deal_workgroup(wg, maxw, blk, k, wt, order, parts_out, nsplit) && ok;
        blk.clear(); blk_slots = 0; blk_fol = 0;
    };
    // invariant: the slots still needed fit into the free slots of this block plus the workgroups left
    for (int w = 0; w < nw && ok; ++w) {
        int kk = k.empty() ? 1 : k[(size_t)w];
        const int wgs_left0 = nwg - (wg + 1);
        if (kk > 1 && blk_fol + kk - 1 > kLdsMaxFollowers) { k[(size_t)w] = 1; kk = 1; }            // (scratch of the workgroup is full: this one stays whole)
        if (blk_slots + kk > cap) {
            // closing the block leaves its free slots unused: only if what is left still fits into the workgroups behind it
            if (wgs_left0 > 0 && slots_after[(size_t)w] <= (int64_t)wgs_left0 * cap) { deal(); ++wg; }
            else { if (kk > 1) k[(size_t)w] = 1; kk = 1; }
            if (blk_slots + kk > cap) { ok = false; break; }
        }
        blk.push_back(w);
        blk_slots += kk; blk_fol += kk - 1;
        acc += wt[(size_t)w];
        const int wgs_left = nwg - (wg + 1);
        const bool full = blk_slots == cap;
        const bool heavy_enough = acc >= total * (double)(wg + 1) / nwg;
        if (wgs_left > 0 && (full || (heavy_enough && slots_after[(size_t)w + 1] <= (int64_t)wgs_left * cap))) { deal(); ++wg; }
    }
    if (ok && !blk.empty()) deal();
    if (nsplit == 0) parts_out.clear();
    return ok;
}

static int build_lds_cells(tcgnn_plan* p, hipStream_t stream, int slot) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (p->lds[slot].nranges > 0) return TCGNN_OK;
    if (p->nw_eff <= 0 || p->Nc <= 0) return fail(TCGNN_ERR_INVALID_ARG, "LDS-range SpMM: empty graph");
    // the stream is cut from the packed tile stream (cols / mask) or, slot kLdsValSlot, from the single-edge one, whose K slots also
    // carry the CSR position of their edge: that index follows every slot into the tiles (s_eidx / d_eidx)
    const bool val = slot == kLdsValSlot;
    if (val && !p->d_xwb_ptr) return fail(TCGNN_ERR_INVALID_ARG, "edge-valued LDS-resident SpMM: the single-edge stream has not been built");
    const int64_t* const s_wb = val ? p->d_xwb_ptr : p->d_wb_ptr;
    const int32_t* const s_cols = val ? p->d_xcols : p->d_cols;
    const uint32_t* const s_mask = val ? p->d_xmask : p->d_mask;
    const int32_t* const s_eidx = val ? p->d_xeidx : nullptr;
    int32_t *d_eidx = nullptr, *d_cold_eidx = nullptr;
    const int maxw = lds_stream_maxw(slot);
    const int rows = lds_stream_buf_rows(slot) - 8;          // data rows of a range
    const int nranges = (p->Nc + rows - 1) / rows;
    const int per_wg = kLdsWaves * maxw;
    const int nwg = lds_workgroups(p, maxw);
    const int nw = nwg * per_wg;                              // window SLOTS of this stream
    const int64_t ncell = (int64_t)nwg * nranges * per_wg;
    uint32_t *d_cnt = nullptr, *d_firstq = nullptr, *d_tiles = nullptr;
    int32_t* d_sorder = nullptr;
    uint32_t* d_parts_guard = nullptr;   // (set below; freed by bail)
    auto bail = [&](int rc) { (void)hipFree(d_cnt); (void)hipFree(d_firstq); (void)hipFree(d_tiles); (void)hipFree(d_sorder); (void)hipFree(d_parts_guard); return rc; };
    // exact weight of every window in THIS stream: the tiles it will cost (at least one, so empty windows still take a slot's worth)
    std::vector<double> exact((size_t)p->nw_eff, 1.0);
    {
        uint32_t* d_wt = nullptr;
        std::vector<uint32_t> h_wt((size_t)p->nw_eff, 0u);
        hipError_t e0 = hipMalloc(&d_wt, h_wt.size() * sizeof(uint32_t));
        if (e0 == hipSuccess) e0 = hipMemsetAsync(d_wt, 0, h_wt.size() * sizeof(uint32_t), stream);
        if (e0 == hipSuccess) {
            const int64_t nth = (int64_t)p->nw_eff * nranges;
            hipLaunchKernelGGL(window_tiles_kernel, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, stream, s_wb, s_cols, p->nw_eff, nranges, p->Nc, rows, d_wt);
            e0 = hipGetLastError();
        }
        if (e0 == hipSuccess) e0 = hipMemcpyAsync(h_wt.data(), d_wt, h_wt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        if (e0 == hipSuccess) e0 = hipStreamSynchronize(stream);
        (void)hipFree(d_wt);
        if (e0 != hipSuccess) return fail(e0 == hipErrorOutOfMemory ? TCGNN_ERR_OOM : TCGNN_ERR_HIP, "window weights: %s", hipGetErrorString(e0));
        for (size_t w = 0; w < h_wt.size(); ++w) exact[w] = std::max<double>(1.0, h_wt[w]);
    }
    // ---- candidate placements: window slots, columns per cell, columns and busiest-wavefront tiles per (workgroup, range) pair,
    //      and from those an estimate of the kernel time (+ the cold remainder's); the cheapest is built
    const int64_t npairs_all = (int64_t)nwg * nranges;
    struct Cand {
        int mode = kPlaceLocal, nsplit = 0;
        uint32_t hot_min = 1u;
        double est_us = 0;
        std::vector<int32_t> sorder;
        std::vector<uint32_t> sparts, paircols, pairmax, pairover, pairtiles;   // pairover: [2][pairs] columns beyond 32 / 64 per cell
        uint32_t *d_cellcols = nullptr, *d_firstq = nullptr, *d_parts = nullptr;
        int32_t* d_sorder = nullptr;
        void release() { (void)hipFree(d_cellcols); (void)hipFree(d_firstq); (void)hipFree(d_parts); (void)hipFree(d_sorder); d_cellcols = d_firstq = d_parts = nullptr; d_sorder = nullptr; }
    };
    const char* const verbose_env0 = getenv("TCGNN_VERBOSE");
    const bool verbose0 = verbose_env0 && atoi(verbose_env0) > 0;
    auto prepare = [&](int mode, Cand& c) -> hipError_t {
        c.mode = mode;
        if (!lds_place_windows(p, nwg, maxw, c.sorder, c.sparts, c.nsplit, exact, mode)) {   // (a placement with parts that ran out of slots)
            c.mode = kPlaceLocal;
            if (!lds_place_windows(p, nwg, maxw, c.sorder, c.sparts, c.nsplit, exact, kPlaceLocal)) return hipErrorInvalidValue;
        }
        // threshold: a range step costs a workgroup ~1.7 us (8-window layout, two passes over 256 CUs: ~13 ns of chip time) or
        // ~1 us (4-window layout, one pass: ~4 ns), a column in the gather walk ~10 ps of chip time.  Forcing the LDS-resident walk
        // (mode 3: tests, timing) keeps every pair that holds a column, and so does the graph-wide placement (no locality to split
        // on: every pair holds about the same share); TCGNN_LDS_HOT_COLS overrides.
        c.hot_min = maxw == kLdsMaxW2 ? 1000u : 400u;
        if (g_spmm_mode == 3 || c.mode == kPlaceGlobal) c.hot_min = 1u;
        if (const char* env = getenv("TCGNN_LDS_HOT_COLS")) c.hot_min = (uint32_t)std::max(1, atoi(env));
        uint32_t *d_pc = nullptr, *d_pm = nullptr, *d_po = nullptr, *d_pt = nullptr;
        hipError_t e = hipMalloc(&c.d_cellcols, (size_t)(ncell + 1) * sizeof(uint32_t));
        if (e == hipSuccess && c.nsplit > 0) {
            e = hipMalloc(&c.d_parts, c.sparts.size() * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMemcpyAsync(c.d_parts, c.sparts.data(), c.sparts.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream);
        }
        if (e == hipSuccess) e = hipMalloc(&c.d_firstq, (size_t)nw * nranges * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&c.d_sorder, c.sorder.size() * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc(&d_pc, (size_t)npairs_all * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&d_pm, (size_t)npairs_all * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&d_po, (size_t)npairs_all * 2 * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&d_pt, (size_t)npairs_all * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpyAsync(c.d_sorder, c.sorder.data(), c.sorder.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipMemsetAsync(c.d_cellcols, 0, (size_t)(ncell + 1) * sizeof(uint32_t), stream);
        std::vector<uint32_t>& pairmax = c.pairmax;
        pairmax.resize((size_t)npairs_all);
        c.paircols.resize((size_t)npairs_all);
        c.pairover.resize((size_t)npairs_all * 2);
        c.pairtiles.resize((size_t)npairs_all);
        if (e == hipSuccess) {
            const int64_t nthreads = (int64_t)nw * nranges;
            hipLaunchKernelGGL(cell_count_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, stream, s_wb, c.d_sorder, s_cols, nw, nwg,
                               nranges, p->Nc, maxw, rows, c.d_cellcols, c.d_firstq, c.d_parts);
            hipLaunchKernelGGL(cell_pair_cols_kernel, dim3((unsigned)((npairs_all + 255) / 256)), dim3(256), 0, stream, c.d_cellcols, npairs_all, per_wg, maxw, d_pc, d_pm, d_po, d_pt);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(c.paircols.data(), d_pc, c.paircols.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(c.pairover.data(), d_po, c.pairover.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(c.pairtiles.data(), d_pt, c.pairtiles.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(pairmax.data(), d_pm, pairmax.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(d_pc); (void)hipFree(d_pm); (void)hipFree(d_po); (void)hipFree(d_pt);
        if (e != hipSuccess) return e;
        // every range ends at a barrier: a workgroup's time is the sum over its hot ranges of a fixed part and its busiest
        // wavefront's tiles (constants of lds_estimate_us); the kernel lasts as long as its slowest workgroup, times the rounds
        // beyond one workgroup per CU; the cold remainder's columns go through a gather walk at ~20 ps each.
        const double rounds = std::max(1.0, std::ceil((double)nwg * (maxw == kLdsMaxW2 ? 2 : 1) / std::max(p->num_cus, 1)));
        auto estimate = [&](uint32_t hot_min) {
            double worst = 0, cold = 0;
            for (int wg = 0; wg < nwg; ++wg) {
                double t = 0;
                for (int r = 0; r < nranges; ++r) {
                    const size_t k = (size_t)wg * nranges + r;
                    if (c.paircols[k] >= hot_min) t += 0.80 + 0.135 * pairmax[k]; else cold += c.paircols[k];
                }
                worst = std::max(worst, t);
            }
            return worst * rounds + cold * 20e-6;
        };
        c.est_us = estimate(c.hot_min);
        if (verbose0 && c.mode != kPlaceGlobal)
            for (uint32_t h : {1u, 125u, 250u, 500u, 1000u, 2000u}) fprintf(stderr, "[tcgnn]   placement %d, hot threshold %u: estimated %.0f us\n", c.mode, h, estimate(h));
        return hipSuccess;
    };
    Cand best;
    {
        const int forced = lds_place_forced();
        const bool hubs = lds_has_hubs(p);
        hipError_t e0 = prepare(forced >= 0 ? forced : (hubs ? kPlaceGlobal : kPlaceLocal), best);
        if (e0 == hipSuccess && forced < 0 && hubs) {
            Cand other;
            e0 = prepare(kPlaceLocalSplit, other);
            if (verbose0) fprintf(stderr, "[tcgnn] cell stream %d: estimated %.0f us longest-first over the graph, %.0f us contiguous blocks (+ cold remainder)\n", slot, best.est_us, other.est_us);
            if (e0 == hipSuccess && other.est_us < best.est_us) std::swap(best, other);
            other.release();
        }
        if (e0 != hipSuccess) { best.release(); return fail(e0 == hipErrorOutOfMemory ? TCGNN_ERR_OOM : TCGNN_ERR_HIP, "cell table: %s", hipGetErrorString(e0)); }
    }
    std::vector<int32_t>& sorder = best.sorder;
    std::vector<uint32_t>& sparts = best.sparts;
    const int nsplit = best.nsplit;
    const uint32_t hot_min = best.hot_min;
    const std::vector<uint32_t>& paircols = best.paircols;
    uint32_t* d_parts = best.d_parts;
    d_parts_guard = d_parts;
    d_firstq = best.d_firstq;
    d_sorder = best.d_sorder;
    uint32_t* d_cellcols = best.d_cellcols;   // (the dense table of step one holds columns per cell)
    hipError_t e = hipSuccess;
    // ---- hot / cold: a (workgroup, range) pair is worth a range fill only if enough of the workgroup's columns fall into it
    uint32_t* d_paircols = nullptr;
    int32_t *d_kmap = nullptr, *d_rbase = nullptr, *d_rlist = nullptr;
    uint32_t* d_coldcols = nullptr;
    int64_t* d_cold_ptr = nullptr;
    int32_t* d_ccols = nullptr;
    uint32_t* d_cmask = nullptr;
    uint32_t* d_flat = nullptr;
    auto bail2 = [&](int rc) {
        (void)hipFree(d_flat); (void)hipFree(d_paircols); (void)hipFree(d_kmap); (void)hipFree(d_rbase); (void)hipFree(d_rlist); (void)hipFree(d_cellcols); (void)hipFree(d_coldcols);
        (void)hipFree(d_cold_ptr); (void)hipFree(d_ccols); (void)hipFree(d_cmask); (void)hipFree(d_eidx); (void)hipFree(d_cold_eidx);
        return bail(rc);
    };
    e = hipMalloc(&d_kmap, (size_t)npairs_all * sizeof(int32_t));
    if (e != hipSuccess) return bail2(fail(e == hipErrorOutOfMemory ? TCGNN_ERR_OOM : TCGNN_ERR_HIP, "cell table: %s", hipGetErrorString(e)));
    std::vector<int32_t> kmap((size_t)npairs_all, -1), rbase((size_t)nwg + 1, 0), rlist;
    int64_t hot_cols = 0, cold_cols = 0;
    for (int wg = 0; wg < nwg; ++wg) {
        rbase[(size_t)wg] = (int32_t)rlist.size();
        for (int r = 0; r < nranges; ++r) {
            const uint32_t c = paircols[(size_t)wg * nranges + r];
            if (c >= hot_min) { kmap[(size_t)wg * nranges + r] = (int32_t)rlist.size(); rlist.push_back(r); hot_cols += c; }
            else cold_cols += c;
        }
    }
    rbase[(size_t)nwg] = (int32_t)rlist.size();
    const int64_t npairs = (int64_t)rlist.size();
    rlist.resize(rlist.size() + 4, 0);
    const int64_t ncell_hot = npairs * per_wg;
    e = hipMalloc(&d_cnt, (size_t)(ncell_hot + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&d_rbase, rbase.size() * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&d_rlist, rlist.size() * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, (size_t)(ncell_hot + 1) * sizeof(uint32_t), stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_kmap, kmap.data(), kmap.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rbase, rbase.data(), rbase.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rlist, rlist.data(), rlist.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return bail2(fail(e == hipErrorOutOfMemory ? TCGNN_ERR_OOM : TCGNN_ERR_HIP, "cell table: %s", hipGetErrorString(e)));
    // ---- flat or ordinary (tcgnn_lds_flat.inc): a flat stream gives every cell of a hot pair exactly tpc tiles and sends what
    //      a cell holds beyond 32 tpc columns to the cold remainder.  Taken when that costs few columns (the cold remainder of a
    //      flat stream is added by a latency-bound kernel) and no more tile steps than the ordinary stream's; never with split
    //      hub windows (their cells are nowhere near uniform).  TCGNN_LDS_FLAT=0 / 1 / 2 forces ordinary / one / two tiles per cell.
    int flat_tpc = 0;
    int64_t over_cols = 0;
    {
        // (what a range lasts is its busiest wavefront's tile steps - every range ends at a barrier: pairmax for the ordinary
        //  stream, maxw x tpc for a flat one, whose empty window slots are skipped but whose wavefronts all wait for a full one)
        int64_t classic_tiles = 0, classic_steps = 0, over[2] = {0, 0};
        for (int64_t k = 0; k < npairs_all; ++k)
            if (kmap[(size_t)k] >= 0) { classic_tiles += best.pairtiles[(size_t)k]; classic_steps += best.pairmax[(size_t)k];
                                        over[0] += best.pairover[(size_t)k]; over[1] += best.pairover[(size_t)(npairs_all + k)]; }
        const char* fenv = getenv("TCGNN_LDS_FLAT");
        const int forced_flat = fenv ? atoi(fenv) : -1;
        if (nsplit == 0 && npairs > 0 && forced_flat != 0) {
            for (int tpc = 1; tpc <= 2 && !flat_tpc; ++tpc) {
                if (maxw * tpc > 16) break;
                const int64_t flat_tiles = ncell_hot * tpc;
                const bool few_cold = (double)(cold_cols + over[tpc - 1]) <= 0.04 * (double)std::max<int64_t>(hot_cols + cold_cols, 1);
                (void)flat_tiles;
                // (a flat step costs ~0.75 of an ordinary one - 1.46 against 1.84 us per range of ~8 steps on the Reddit shape - and a
                //  stream of nearly-empty cells, a wide row shard, ties on the count: 627 712 flat steps against 627 699)
                const bool no_more_steps = (double)(npairs * maxw * tpc) <= 1.2 * (double)std::max<int64_t>(classic_steps, 1);
                if (forced_flat == tpc || (forced_flat < 0 && few_cold && no_more_steps)) { flat_tpc = tpc; over_cols = over[tpc - 1]; }
            }
        }
        if (verbose0) fprintf(stderr, "[tcgnn] cell stream %d: %lld tiles ordinary, %lld steps of the busiest wavefronts; flat would take %lld steps (+%lld columns cold) / %lld (+%lld): %s\n", slot,
                              (long long)classic_tiles, (long long)classic_steps, (long long)(npairs * maxw), (long long)over[0], (long long)(2 * npairs * maxw), (long long)over[1],
                              flat_tpc ? (flat_tpc == 1 ? "flat, 1 tile per cell" : "flat, 2 tiles per cell") : "ordinary");
    }
    if (flat_tpc && !getenv("TCGNN_LDS_FLAT")) {
        // ... nor where ONE window would leave a long remainder behind (a hub row: its cells overflow in every range, and the
        // remainder of a window is one wavefront's serial work - a 24 k-degree hub cost the flat walk 0.72 ms against 0.61):
        // the remainder is counted with the flat cap before anything is built
        uint32_t* d_trial = nullptr;
        std::vector<uint32_t> trial((size_t)p->nw_eff, 0u);
        e = hipMalloc(&d_trial, trial.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemsetAsync(d_trial, 0, trial.size() * sizeof(uint32_t), stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(cell_cold_count_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, stream, d_cellcols, d_kmap, d_sorder, nw, nranges, maxw, d_trial, (uint32_t)(32 * flat_tpc));
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(trial.data(), d_trial, trial.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(d_trial);
        if (e != hipSuccess) return bail2(fail(e == hipErrorOutOfMemory ? TCGNN_ERR_OOM : TCGNN_ERR_HIP, "cell table: %s", hipGetErrorString(e)));
        uint32_t worst = 0;
        for (uint32_t c : trial) worst = std::max(worst, (c + 31u) / 32u);
        if (worst > 16u) {
            if (verbose0) fprintf(stderr, "[tcgnn] cell stream %d: ordinary after all - one window would leave %u cold tiles behind\n", slot, worst);
            flat_tpc = 0; over_cols = 0;
        }
    }
    if (val && flat_tpc != 1)   // (spmm_lds_val_kernel walks flat streams with one tile per cell only; the caller keeps the gather walks)
        return bail2(fail(TCGNN_ERR_UNSUPPORTED, "edge-valued LDS-resident SpMM: the single-edge cells of this graph are not uniform enough for a flat stream"));
    hot_cols -= over_cols; cold_cols += over_cols;
    if (ncell > 0) hipLaunchKernelGGL(cell_compact_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, stream, d_cellcols, d_kmap, npairs_all, per_wg, d_cnt);
    std::vector<uint32_t> cnt((size_t)ncell_hot + 1);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(cnt.data(), d_cnt, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return bail2(fail(TCGNN_ERR_HIP, "cell count: %s", hipGetErrorString(e)));
    uint64_t run = 0;
    if (flat_tpc) run = (uint64_t)ncell_hot * (uint64_t)flat_tpc;   // (positions are computed: the table is not kept)
    else for (size_t k = 0; k < cnt.size(); ++k) { const uint32_t c = cnt[k]; cnt[k] = (uint32_t)run; run += c; }
    if (run >= (1ull << 32)) return bail2(fail(TCGNN_ERR_BAD_GRAPH, "LDS-range SpMM: %llu tiles overflow the 32-bit cell table", (unsigned long long)run));
    const int64_t ntiles = (int64_t)run;
    const int64_t nwords = std::max<int64_t>(ntiles, 1) * kCellWords;
    e = hipMalloc(&d_tiles, (size_t)nwords * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpyAsync(d_cnt, cnt.data(), cnt.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return bail2(fail(e == hipErrorOutOfMemory ? TCGNN_ERR_OOM : TCGNN_ERR_HIP, "cell stream (%lld tiles): %s", (long long)ntiles, hipGetErrorString(e)));
    hipLaunchKernelGGL(cell_init_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, stream, d_tiles, nwords, rows);
    if (val) {   // (flat only: the caller gives the stream up otherwise)
        e = hipMalloc(&d_eidx, (size_t)std::max<int64_t>(ntiles, 1) * 32 * sizeof(int32_t));
        if (e == hipSuccess) e = hipMemsetAsync(d_eidx, 0xff, (size_t)std::max<int64_t>(ntiles, 1) * 32 * sizeof(int32_t), stream);
        if (e != hipSuccess) return bail2(fail(e == hipErrorOutOfMemory ? TCGNN_ERR_OOM : TCGNN_ERR_HIP, "edge index of the cell stream: %s", hipGetErrorString(e)));
    }
    hipLaunchKernelGGL(cell_fill_kernel, dim3((unsigned)nw), dim3(256), 0, stream, s_wb, d_sorder, s_cols, s_mask, nwg, nranges, p->Nc,
                       maxw, rows, d_cnt, d_firstq, d_tiles, d_kmap, d_parts, d_cellcols, flat_tpc, s_eidx, d_eidx);
    if (ntiles > 0) hipLaunchKernelGGL(cell_optimize_kernel, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, stream, d_tiles, ntiles, rows, d_eidx);
    e = hipGetLastError();
    if (e == hipSuccess && flat_tpc) {   // tiles -> per-(pair, wavefront) metadata blocks; the ordinary tiles and the cell table go
        e = hipMalloc(&d_flat, (size_t)nwords * sizeof(uint32_t));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(flat_transpose_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, stream, d_tiles, d_flat, npairs * kLdsWaves, maxw * flat_tpc);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(d_tiles); d_tiles = nullptr;
        (void)hipFree(d_cnt); d_cnt = nullptr;
        if (e != hipSuccess) return bail2(fail(e == hipErrorOutOfMemory ? TCGNN_ERR_OOM : TCGNN_ERR_HIP, "flat cell stream: %s", hipGetErrorString(e)));
    }
    // ---- the cold remainder, re-condensed per window for the gather walk
    int64_t cold_tiles = 0, cold_max = 0;
    size_t cold_bytes = 0;
    std::vector<int64_t> cptr;
    if (e == hipSuccess && cold_cols > 0) {
        const int nwe = p->nw_eff;
        std::vector<uint32_t> coldc((size_t)nwe, 0);
        e = hipMalloc(&d_coldcols, (size_t)nwe * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemsetAsync(d_coldcols, 0, (size_t)nwe * sizeof(uint32_t), stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(cell_cold_count_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, stream, d_cellcols, d_kmap, d_sorder, nw, nranges, maxw, d_coldcols,
                               (uint32_t)(32 * flat_tpc));
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(coldc.data(), d_coldcols, coldc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        cptr.assign((size_t)nwe + 1, 0);
        for (int w = 0; w < nwe; ++w) { cptr[(size_t)w + 1] = cptr[(size_t)w] + (coldc[(size_t)w] + 31) / 32; cold_max = std::max<int64_t>(cold_max, (coldc[(size_t)w] + 31) / 32); }
        cold_tiles = cptr[(size_t)nwe];
        const size_t b_ptr = cptr.size() * sizeof(int64_t), b_c = (size_t)std::max<int64_t>(cold_tiles, 1) * kWbCols * 4, b_m = (size_t)std::max<int64_t>(cold_tiles, 1) * kWinRows * 4;
        if (e == hipSuccess) e = hipMalloc(&d_cold_ptr, b_ptr);
        if (e == hipSuccess) e = hipMalloc(&d_ccols, b_c);
        if (e == hipSuccess) e = hipMalloc(&d_cmask, b_m);
        if (e == hipSuccess) e = hipMemcpyAsync(d_cold_ptr, cptr.data(), b_ptr, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_cmask, 0, b_m, stream);
        if (e == hipSuccess) {
            // (padding columns of a window's last tile point at the all-zero sentinel row, like pack_kernel's)
            std::vector<int32_t> fillv((size_t)std::max<int64_t>(cold_tiles, 1) * kWbCols, p->Nc);
            e = hipMemcpyAsync(d_ccols, fillv.data(), b_c, hipMemcpyHostToDevice, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
        }
        if (e == hipSuccess && val) {
            e = hipMalloc(&d_cold_eidx, (size_t)std::max<int64_t>(cold_tiles, 1) * 32 * sizeof(int32_t));
            if (e == hipSuccess) e = hipMemsetAsync(d_cold_eidx, 0xff, (size_t)std::max<int64_t>(cold_tiles, 1) * 32 * sizeof(int32_t), stream);
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(cell_cold_fill_kernel, dim3((unsigned)nw), dim3(256), 0, stream, s_wb, d_sorder, s_cols, s_mask, d_kmap, nranges, p->Nc,
                               maxw, rows, d_cold_ptr, d_ccols, d_cmask, d_parts, d_firstq, (uint32_t)(32 * flat_tpc), s_eidx, d_cold_eidx);
            e = hipGetLastError();
        }
        cold_bytes = b_ptr + b_c + b_m;
    }
    // ---- a flat stream whose layout leaves 16 x NT KB of LDS: the remainder as per-wavefront record lists, multiplied inside the
    //      flat kernel one tile per range (tcgnn_lds_flat.inc: cold_step) instead of by spmm_cold_planar_kernel afterwards
    int32_t* d_wcold_ptr = nullptr;
    uint32_t *d_wcold = nullptr, *d_wlist = nullptr;
    static const int wcold_enabled = [] { const char* en = getenv("TCGNN_LDS_COLD_INSIDE"); return en ? atoi(en) : 1; }();
    if (e == hipSuccess && !val && flat_tpc && cold_tiles > 0 && cold_tiles < (1ll << 28) && wcold_enabled && flat_cold_fits(lds_stream_nt(slot), maxw, flat_tpc)) {
        std::vector<int32_t> wptr((size_t)nwg * kLdsWaves + 1, 0);
        std::vector<uint32_t> wlist;
        wlist.reserve((size_t)cold_tiles);
        for (int g2 = 0; g2 < nwg; ++g2)
            for (int v = 0; v < kLdsWaves; ++v) {
                wptr[(size_t)g2 * kLdsWaves + v] = (int32_t)wlist.size();
                for (int j = 0; j < maxw; ++j) {
                    const int w = sorder[(size_t)cell_position(g2, v, j, maxw)];
                    if (w < 0) continue;
                    for (int64_t t = cptr[(size_t)w]; t < cptr[(size_t)w + 1]; ++t) wlist.push_back((uint32_t)t | ((uint32_t)j << 28));
                }
            }
        wptr[(size_t)nwg * kLdsWaves] = (int32_t)wlist.size();
        e = hipMalloc(&d_wcold_ptr, wptr.size() * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc(&d_wlist, std::max<size_t>(wlist.size(), 1) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&d_wcold, std::max<size_t>(wlist.size(), 1) * 64 * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpyAsync(d_wcold_ptr, wptr.data(), wptr.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess && !wlist.empty()) e = hipMemcpyAsync(d_wlist, wlist.data(), wlist.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess && !wlist.empty()) {
            const int64_t nwordsw = (int64_t)wlist.size() * 64;
            hipLaunchKernelGGL(flat_cold_records_kernel, dim3((unsigned)((nwordsw + 255) / 256)), dim3(256), 0, stream, d_wlist, (int64_t)wlist.size(), d_ccols, d_cmask, d_wcold);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(d_wlist);
        if (e != hipSuccess) { (void)hipFree(d_wcold_ptr); (void)hipFree(d_wcold); d_wcold_ptr = nullptr; d_wcold = nullptr; }
        else cold_bytes += wptr.size() * sizeof(int32_t) + wlist.size() * 256;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);   // host vectors must outlive their copies
    if (e != hipSuccess) return bail2(fail(e == hipErrorOutOfMemory ? TCGNN_ERR_OOM : TCGNN_ERR_HIP, "cell fill: %s", hipGetErrorString(e)));
    (void)hipFree(d_firstq); (void)hipFree(d_paircols); (void)hipFree(d_kmap); (void)hipFree(d_cellcols); (void)hipFree(d_coldcols);
    tcgnn_plan::CellStream& cs = p->lds[slot];
    cs.d_order = d_sorder;
    cs.d_cell_ptr = d_cnt; cs.d_cell_tiles = d_tiles;
    cs.flat_tpc = flat_tpc; cs.d_flat = d_flat;
    cs.d_wcold_ptr = d_wcold_ptr; cs.d_wcold = d_wcold;
    cs.d_eidx = d_eidx; cs.d_cold_eidx = d_cold_eidx;
    cs.nwg = nwg; cs.tiles = ntiles;
    cs.npairs = (int32_t)npairs; cs.d_rbase = d_rbase; cs.d_rlist = d_rlist;
    cs.cold_tiles = cold_tiles; cs.hot_cols = hot_cols; cs.cold_cols = cold_cols; cs.cold_max = cold_max;
    cs.d_parts = d_parts; cs.nsplit = nsplit;
    const char* const verbose_env = getenv("TCGNN_VERBOSE");   // (read per build: builds are rare, and tests switch it on)
    const bool verbose = verbose_env && atoi(verbose_env) > 0;
    if (verbose)
        fprintf(stderr, "[tcgnn] cell stream %d (%d windows per wavefront, %d-row ranges): %d workgroups x %d ranges, %lld of %lld pairs hot (>= %u columns), "
                        "%lld columns hot / %lld cold, %lld tiles + %lld cold gather tiles, placement %s\n",
                slot, maxw, rows, nwg, nranges, (long long)npairs, (long long)npairs_all, hot_min, (long long)hot_cols, (long long)cold_cols,
                (long long)ntiles, (long long)cold_tiles, best.mode == kPlaceGlobal ? "longest-first over the graph" : "contiguous blocks");
    if (verbose && flat_tpc) fprintf(stderr, "[tcgnn]   flat: %d tile(s) per cell, %lld columns beyond the cells' tiles moved to the cold remainder (%s)\n", flat_tpc, (long long)over_cols,
                                     d_wcold_ptr ? "multiplied inside the kernel, one tile per range" : "added by spmm_cold_planar_kernel");
    if (verbose && nsplit) fprintf(stderr, "[tcgnn]   %d windows split over several wavefronts\n", nsplit);
    cs.d_cold_ptr = d_cold_ptr; cs.d_cold_cols = d_ccols; cs.d_cold_mask = d_cmask;
    p->bytes += (flat_tpc ? 0 : (size_t)(ncell_hot + 1) * sizeof(uint32_t)) + (size_t)nwords * sizeof(uint32_t) + sorder.size() * sizeof(int32_t) +
                (rbase.size() + rlist.size()) * sizeof(int32_t) + cold_bytes + (nsplit ? sparts.size() * sizeof(uint32_t) : 0) +
                (val ? (size_t)(std::max<int64_t>(ntiles, 1) + (cold_tiles > 0 ? cold_tiles : 0)) * 32 * sizeof(int32_t) : 0);
    cs.nranges = nranges;
    return TCGNN_OK;
}

// ---- the edge-valued LDS-resident walk (tcgnn_lds_val.inc): single-edge tile stream + its flat cell stream (slot kLdsValSlot).
// Built on the first edge-valued call that would use it (allocates and synchronises: never inside a graph capture); the answer -
// usable or not - is remembered in plan->val_choice.
static int build_val_stream(tcgnn_plan* p, hipStream_t stream) {
    {
        static std::mutex mu;
        std::lock_guard<std::mutex> lock(mu);
        if (p->val_choice.load(std::memory_order_acquire) >= 0) return TCGNN_OK;
        if (!p->d_xwb_ptr) {
            const int nw = p->nw_eff;
            int32_t *d_ew = nullptr, *d_flags = nullptr;
            std::vector<int32_t> ew((size_t)nw, 0);
            hipError_t e = hipMalloc(&d_ew, (size_t)nw * sizeof(int32_t));
            if (e == hipSuccess) e = hipMalloc(&d_flags, sizeof(int32_t));
            if (e == hipSuccess) e = hipMemsetAsync(d_flags, 0, sizeof(int32_t), stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(window_edges_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, stream, p->rowptr, p->N, nw, d_ew);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(ew.data(), d_ew, (size_t)nw * sizeof(int32_t), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            (void)hipFree(d_ew);
            std::vector<int64_t> xptr((size_t)nw + 1, 0);
            for (int w = 0; w < nw; ++w) xptr[(size_t)w + 1] = xptr[(size_t)w] + (ew[(size_t)w] + kWbCols - 1) / kWbCols;
            const int64_t total = std::max<int64_t>(xptr[(size_t)nw], 1);
            int64_t* d_xp = nullptr; int32_t *d_xc = nullptr, *d_xe = nullptr; uint32_t* d_xm = nullptr;
            auto drop = [&]() { (void)hipFree(d_xp); (void)hipFree(d_xc); (void)hipFree(d_xe); (void)hipFree(d_xm); (void)hipFree(d_flags); };
            if (e == hipSuccess) e = hipMalloc(&d_xp, xptr.size() * sizeof(int64_t));
            if (e == hipSuccess) e = hipMalloc(&d_xc, (size_t)total * kWbCols * sizeof(int32_t));
            if (e == hipSuccess) e = hipMalloc(&d_xe, (size_t)total * kWbCols * sizeof(int32_t));
            if (e == hipSuccess) e = hipMalloc(&d_xm, (size_t)total * kWinRows * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMemcpyAsync(d_xp, xptr.data(), xptr.size() * sizeof(int64_t), hipMemcpyHostToDevice, stream);
            if (e == hipSuccess) e = hipMemsetAsync(d_xe, 0xff, (size_t)total * kWbCols * sizeof(int32_t), stream);
            if (e == hipSuccess) e = hipMemsetAsync(d_xm, 0, (size_t)total * kWinRows * sizeof(uint32_t), stream);
            if (e == hipSuccess) {
                std::vector<int32_t> fillv((size_t)total * kWbCols, p->Nc);   // (padding slots point at the all-zero sentinel row, like pack_kernel's)
                e = hipMemcpyAsync(d_xc, fillv.data(), fillv.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream);
                if (e == hipSuccess) e = hipStreamSynchronize(stream);
            }
            int32_t bad = 0;
            if (e == hipSuccess && nw > 0) {
                hipLaunchKernelGGL(expand_edges_kernel, dim3((unsigned)nw), dim3(256), 0, stream, p->d_wb_ptr, p->d_cols, p->d_mask, p->d_ebase, d_xp, d_xc, d_xm, d_xe, p->Nc, d_flags);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_flags, sizeof(int32_t), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess || bad) {
                drop();
                p->val_choice.store(0, std::memory_order_release);
                return e == hipSuccess ? TCGNN_OK : fail(e == hipErrorOutOfMemory ? TCGNN_ERR_OOM : TCGNN_ERR_HIP, "single-edge tile stream: %s", hipGetErrorString(e));
            }
            (void)hipFree(d_flags);
            p->d_xwb_ptr = d_xp; p->d_xcols = d_xc; p->d_xmask = d_xm; p->d_xeidx = d_xe; p->total_xwb = xptr[(size_t)nw];
            p->bytes += xptr.size() * sizeof(int64_t) + (size_t)total * (2 * kWbCols * sizeof(int32_t) + kWinRows * sizeof(uint32_t));
        }
    }
    const int rc = build_lds_cells(p, stream, kLdsValSlot);
    tcgnn_plan::CellStream& cs = p->lds[kLdsValSlot];
    bool ok = rc == TCGNN_OK && cs.nranges > 0 && cs.flat_tpc == 1 && cs.nsplit == 0 && cs.d_eidx;
    if (ok) {   // CSR positions -> 16-bit offsets into each window's run of edges (half the index bytes per call and in the plan)
        const size_t nt = (size_t)std::max<int64_t>(cs.tiles, 1), nc = (size_t)std::max<int64_t>(cs.cold_tiles, 1);
        int32_t* d_bad = nullptr; int32_t bad = 0;
        hipError_t e = hipMalloc(&cs.d_eidx16, nt * 32 * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMalloc(&cs.d_cold_eidx16, nc * 32 * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMalloc(&d_bad, sizeof(int32_t));
        if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof(int32_t), stream);
        if (e == hipSuccess) e = hipMemsetAsync(cs.d_eidx16, 0xff, nt * 32 * sizeof(uint16_t), stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(val_index16_kernel, dim3((unsigned)(cs.nwg * kLdsWaves * kLdsMaxW2)), dim3(256), 0, stream, p->rowptr, cs.d_order, cs.d_eidx, cs.d_rbase, cs.d_eidx16, p->N,
                               kLdsMaxW2, cs.cold_tiles > 0 ? cs.d_cold_ptr : nullptr, cs.d_cold_eidx, cs.d_cold_eidx16, d_bad);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(d_bad);
        ok = e == hipSuccess && !bad;
        if (ok) p->bytes += (nt + nc) * 32 * sizeof(uint16_t);
    }
    if (cs.d_eidx) {
        (void)hipStreamSynchronize(stream);
        p->bytes -= (size_t)(std::max<int64_t>(cs.tiles, 1) + (cs.cold_tiles > 0 ? cs.cold_tiles : 0)) * 32 * sizeof(int32_t);
        (void)hipFree(cs.d_eidx); (void)hipFree(cs.d_cold_eidx); cs.d_eidx = nullptr; cs.d_cold_eidx = nullptr;
    }
    // (the single-edge source is only needed to cut the cell stream: 0.9 GB on the Reddit shape, released here)
    (void)hipStreamSynchronize(stream);
    p->bytes -= ((size_t)p->nw_eff + 1) * sizeof(int64_t) + (size_t)std::max<int64_t>(p->total_xwb, 1) * (2 * kWbCols * sizeof(int32_t) + kWinRows * sizeof(uint32_t));
    (void)hipFree(p->d_xwb_ptr); (void)hipFree(p->d_xcols); (void)hipFree(p->d_xmask); (void)hipFree(p->d_xeidx);
    p->d_xwb_ptr = nullptr; p->d_xcols = nullptr; p->d_xmask = nullptr; p->d_xeidx = nullptr;
    p->val_choice.store(ok ? 1 : 0, std::memory_order_release);
    return (rc == TCGNN_ERR_OOM || rc == TCGNN_ERR_HIP) ? rc : TCGNN_OK;
}
// bytes the per-call slot values take behind the planar image (the stream's tiles and its cold tiles, 64 bytes each)
static size_t val_stream_bytes(const tcgnn_plan* p) {
    const tcgnn_plan::CellStream& cs = p->lds[kLdsValSlot];
    if (p->val_choice.load(std::memory_order_acquire) != 1) return 0;
    return ((size_t)(std::max<int64_t>(cs.tiles, 1) + std::max<int64_t>(cs.cold_tiles, 0)) * 64 + 255) / 256 * 256;
}

// columns one gather-walk launch may cover: the widest row whose pitch the structured descriptor can express, in whole
// 128-column chunks.  Wider matrices go through the gather walks as independent column blocks (ld = the full row length).
static constexpr int kMaxGatherBlockDims = 4096;

static bool agnn_supported(const tcgnn_plan* plan, int32_t D);
static int run_spmm(const tcgnn_plan* plan, const float* d_X, const float* d_val, float* d_Y, int32_t D,
                    void* ws, size_t ws_bytes, void* stream_v, int relu = 0, const float* d_gate = nullptr, const void* d_staged = nullptr,
                    int64_t ld = 0, bool block_of_wider = false, const float* d_W = nullptr, int32_t D_out = 0) {
    if (!plan || D < 1 || (plan->N > 0 && ((!d_X && !d_staged) || !d_Y))) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_spmm: null argument or D < 1");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    if (plan->N == 0) return TCGNN_OK;
    if (ld == 0) ld = D;
    if (!block_of_wider && (int64_t)plan->nw_eff * kWinRows < plan->N) // windows the caller did not describe stay zero, like zeros_like
        HIP_TRY(hipMemsetAsync(d_Y, 0, (size_t)plan->N * (d_W ? D_out : D) * sizeof(float), stream));
    if (d_val && (!plan->canonical || plan->E < 4)) {
        hipLaunchKernelGGL(spmm_val_csr_kernel, dim3((unsigned)((plan->N + 3) / 4)), dim3(256), 0, stream,
                           plan->rowptr, plan->col, d_val, d_X, d_Y, plan->N, D);
        HIP_TRY(hipGetLastError());
        return TCGNN_OK;
    }
    const int mode = g_spmm_mode; // 0 auto, 1 plain, 2 blocked, 3 LDS-resident ranges, 4 single-launch fp32 kernel
    if (!d_val && !d_staged && !d_W && plan->nw_eff > 0 && (mode == 4 || (mode == 0 && plan->total_wb <= kSmallMaxTiles))) {
        const SpmmSmallArgs sa{plan->d_wb_ptr, plan->d_cols, plan->d_mask, d_X, d_gate, d_Y, plan->N, plan->Nc, D, relu, nullptr};
        KernelTimer timer(plan, stream, "spmm_small_kernel");
        hipLaunchKernelGGL(spmm_small_kernel, dim3((unsigned)plan->nw_eff, (unsigned)((D + 63) / 64)), dim3(64), 0, stream, sa);
        HIP_TRY(hipGetLastError());
        return TCGNN_OK;
    }
    const uint32_t* hdr; const _Float16* x16; int dpad, pitch;
    // (the planar image is addressed as planes * rows 32-byte records through one buffer descriptor: 31 bits of record index)
    bool lds = !d_val && !d_staged && !block_of_wider && plan->nw_eff > 0 && (mode == 3 || (mode == 0 && lds_chosen(plan, round_up(D, 16)))) &&
               (int64_t)((D + 15) / 16) * ((int64_t)plan->Nc + 1) * 32 < ((int64_t)1 << 32);   // records * 32 B inside the descriptor's 32-bit offset
    // f3 on the LDS-resident kernel: one pass stores its product, the two 32-column passes of a 64-column matrix ADD theirs into a
    // zeroed Y (two addends: the sum does not depend on their order); wider inputs would need an ordered reduction - gather walk
    if (lds && d_W && !(round_up(D, 16) <= 64 && !g_lds_maxw)) lds = false;
    if (lds && block_of_wider) lds = false;
    LdsPass passes[2]; int npass = 0;
    if (lds) {
        // every (layout, pass width) has its own cell stream, built the first time it is needed (plan creation builds the
        // one a 64-column matrix uses; a first call with another width synchronises the stream once) - before the staging
        // pass, which lays the image out for the kernel that will run, and before the timer starts
        npass = lds_passes(round_up(D, 16), passes);
        tcgnn_plan* mp = const_cast<tcgnn_plan*>(plan);
        for (int i = 0; i < npass && lds; ++i) {
            const int slot = lds_stream_of(passes[i].nt, passes[i].maxw);
            if (plan->lds[slot].nranges > 0) continue;
            // a stream that is being captured into a HIP graph cannot allocate or synchronise: this call takes the gather walks
            // (nothing is remembered: the next call outside a capture builds the cell stream)
            hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) {
                if (mode == 3) return fail(TCGNN_ERR_INVALID_ARG, "LDS-resident SpMM: the cell stream of this width has to be built before the call is captured into a graph");
                lds = false;
                break;
            }
            const int b = build_lds_cells(mp, stream, slot);
            if (b && mode == 3) return b;
            if (b) {   // automatic mode: the gather walks need no stream (e.g. no memory left for it); remember the answer
                lds = false;
                if (round_up(D, 16) / 16 <= 64) plan->lds_choice[round_up(D, 16) / 16] = 0;
            }
        }
    }
    // ---- a plan with locality: the (workgroup, range) pairs too thin for a range fill were left out of the cell stream and sit
    //      in a re-condensed remainder that the gather walk ADDS afterwards.  One stream must serve every pass of the call (the
    //      remainder is per stream), the fused dense update cannot span two kernels, and a stream that kept less than half of the
    //      columns is not worth its range fills at all.
    const tcgnn_plan::CellStream* cold = nullptr;
    if (lds) {
        const tcgnn_plan::CellStream& c0 = plan->lds[lds_stream_of(passes[0].nt, passes[0].maxw)];
        bool any_cold = false, thin = false, flat_cold = false;
        for (int i = 0; i < npass; ++i) {
            const tcgnn_plan::CellStream& ci = plan->lds[lds_stream_of(passes[i].nt, passes[i].maxw)];
            // (a flat stream's remainder is added per pass from the planar image by spmm_cold_planar_kernel: no second image, any
            //  number of passes - but not under the fused dense update, which needs the whole sum before it multiplies)
            if (ci.flat_tpc) flat_cold = flat_cold || (ci.cold_tiles > 0 && !ci.d_wcold_ptr);   // (a remainder multiplied inside the flat kernel needs nothing from here)
            else any_cold = any_cold || ci.cold_tiles > 0;
            thin = thin || ci.hot_cols * 2 < ci.hot_cols + ci.cold_cols;
        }
        if ((any_cold && (npass > 1 || d_W)) || (flat_cold && d_W)) {
            lds = false;
            // (a width whose passes can never share one remainder: settle the choice, so later calls - and tcgnn_workspace_bytes, which
            //  reserves a second image for LDS-chosen widths - stop coming back here)
            if (any_cold && npass > 1 && mode != 3 && round_up(D, 16) / 16 <= 64) plan->lds_choice[round_up(D, 16) / 16] = 0;
        }
        else if (thin && mode != 3) { lds = false; if (round_up(D, 16) / 16 <= 64) plan->lds_choice[round_up(D, 16) / 16] = 0; }
        else if (any_cold) cold = &c0;
    }
    if (!lds && !pitch_fits_descriptor(D)) {
        // a row too long for the gather walks' buffer descriptor: independent column blocks, every one rounded
        // with the scale of the whole matrix (absmax over all of X / the edge values here, once)
        if (d_staged) return fail(TCGNN_ERR_UNSUPPORTED, "tcgnn_spmm_staged: rows of %d columns exceed the %d-byte descriptor stride", D, kMaxStructStride);
        if (block_of_wider) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_spmm: nested column blocks");
        const size_t need = workspace_bytes_for(plan->Nc, D);
        if (!ws || ws_bytes < need || (reinterpret_cast<uintptr_t>(ws) & 255))
            return fail(TCGNN_ERR_WORKSPACE, "workspace: need %zu bytes 256-aligned, got %zu at %p", need, ws_bytes, ws);
        uint32_t* whdr = static_cast<uint32_t*>(ws);
        HIP_TRY(hipMemsetAsync(whdr, 0, 32, stream));
        const int64_t nx = (int64_t)plan->Nc * D;
        const int grid = absmax_grid(nx);
        if (d_gate) hipLaunchKernelGGL(absmax_gated_kernel, dim3(grid), dim3(kAbsmaxThreads), 0, stream, d_X, d_gate, nx, whdr, whdr + 2, guard_spmm(plan).cap, 1u);
        else hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(kAbsmaxThreads), 0, stream, d_X, nx, whdr, whdr + 2, guard_spmm(plan).cap, 1u);
        if (d_val && plan->E > 0) {
            const int g2 = absmax_grid(plan->E);
            hipLaunchKernelGGL(absmax_kernel, dim3(g2), dim3(kAbsmaxThreads), 0, stream, d_val, plan->E, whdr + 1, whdr + 3, guard_spmm(plan).cap, 0u);
        }
        HIP_TRY(hipGetLastError());
        for (int c0 = 0; c0 < D; c0 += kMaxGatherBlockDims) {
            const int db = std::min(kMaxGatherBlockDims, D - c0);
            const int rc = run_spmm(plan, d_X + c0, d_val, d_Y + c0, db, ws, ws_bytes, stream_v, relu, d_gate ? d_gate + c0 : nullptr, nullptr, D, true);
            if (rc) return rc;
        }
        return TCGNN_OK;
    }
    // ---- edge values on the LDS-resident flat walk (r04, tcgnn_lds_val.inc): whole 64-column chunks, on graphs the binary SpMM's
    //      time model sends to the LDS-resident kernel, canonical CSR (the single-edge stream is cut with the packed edge offsets).
    //      The first such call builds the stream (unless it is being captured into a graph) and still takes a gather walk - its
    //      workspace was sized before the stream existed; later calls find tcgnn_workspace_bytes grown by the slot values.
    bool val_lds = false;
    {
        const int dp = round_up(D, 16);
        if (d_val && !d_staged && !block_of_wider && !d_gate && !relu && !d_W && (mode == 0 || mode == 3) && dp % 64 == 0 && dp <= 2 * kMaxChunkDims && plan->canonical &&
            plan->nw_eff > 0 && (int64_t)(dp / 16) * ((int64_t)plan->Nc + 1) * 32 < ((int64_t)1 << 32) && (mode == 3 || lds_chosen(plan, dp))) {
            if (plan->val_choice.load(std::memory_order_acquire) < 0) {
                hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
                if (!(hipStreamIsCapturing(stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone)) {
                    const int b = build_val_stream(const_cast<tcgnn_plan*>(plan), stream);
                    if (b) return b;
                }
            }
            val_lds = plan->val_choice.load(std::memory_order_acquire) == 1 && ws_bytes >= workspace_bytes_for(plan->Nc, D) + val_stream_bytes(plan);
        }
    }
    if (d_staged) {   // the caller built the (row-major) fp16 image itself: tcgnn_spmm_staged
        hdr = static_cast<const uint32_t*>(d_staged);
        x16 = reinterpret_cast<const _Float16*>(static_cast<const char*>(d_staged) + kHdrBytes);
        dpad = round_up(D, 16);
        pitch = x16_pitch(dpad);
    } else {
        const int rc = stage_features(plan, d_X, d_val, D, ws, ws_bytes, stream, &hdr, &x16, &dpad, &pitch, lds || val_lds, d_gate, block_of_wider ? ld : 0, block_of_wider);
        if (rc) return rc;
    }
    if (plan->nw_eff == 0) return TCGNN_OK;
    // the range guard's fallback, launched behind the fp16-path kernels of this call (returns at once unless the staged matrix is
    // "wide": range_is_wide).  An image the caller staged itself carries no range words: no guard.
    auto wide_fallback = [&]() -> int {
        if (d_staged) return TCGNN_OK;
        const unsigned grid = (unsigned)std::min<int64_t>(((int64_t)plan->N + 3) / 4, 4096);
        if (d_W) hipLaunchKernelGGL(spmm_gemm_wide_fallback_kernel, dim3(grid), dim3(256), 0, stream, hdr, plan->rowptr, plan->col, d_X, d_W, d_Y, plan->N, D, D_out, relu);
        else if (!d_val && ld == D) {   // binary A, whole rows: the fp32-MFMA walk small graphs take anyway (10-bit operands, fp32's exponent)
            const SpmmSmallArgs sa{plan->d_wb_ptr, plan->d_cols, plan->d_mask, d_X, d_gate, d_Y, plan->N, plan->Nc, D, relu, hdr};
            hipLaunchKernelGGL(spmm_small_kernel, dim3((unsigned)plan->nw_eff, (unsigned)((D + 63) / 64)), dim3(64), 0, stream, sa);
        }
        else hipLaunchKernelGGL(spmm_wide_fallback_kernel, dim3(grid), dim3(256), 0, stream, hdr, d_val ? 1 : 0, plan->rowptr, plan->col, d_val, (const float*)nullptr, d_X, d_gate, d_Y,
                                plan->N, D, (int64_t)ld, (int64_t)ld, relu, (!d_val && !plan->canonical) ? 1 : 0);
        HIP_TRY(hipGetLastError());
        return TCGNN_OK;
    };
    // ---- edge values on the LDS-resident flat walk (r04, tcgnn_lds_val.inc; decided above, before the staging pass)
    if (val_lds) {
        const tcgnn_plan::CellStream& cs = plan->lds[kLdsValSlot];
        const size_t image = workspace_bytes_for(plan->Nc, D);
        _Float16* const vals = reinterpret_cast<_Float16*>(static_cast<char*>(ws) + image);
        _Float16* const cvals = vals + (size_t)std::max<int64_t>(cs.tiles, 1) * 32;
        KernelTimer timer(plan, stream, cs.cold_tiles > 0 ? "val_permute_kernel + spmm_lds_val_kernel + spmm_cold_val_kernel (cold remainder)" : "val_permute_kernel + spmm_lds_val_kernel");
        {
            static bool attr_set = false;
            if (!attr_set) { HIP_TRY(hipFuncSetAttribute((const void*)val_permute_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kValSpanHalves * 2)); attr_set = true; }
            hipLaunchKernelGGL(val_permute_kernel, dim3((unsigned)(cs.nwg * kLdsWaves * (kLdsMaxW2 / kValWpb))), dim3(kValThreads), kValSpanHalves * 2, stream, d_val, plan->rowptr, cs.d_order, cs.d_eidx16, cs.d_rbase, hdr,
                               vals, plan->N, cs.cold_tiles > 0 ? cs.d_cold_ptr : nullptr, cs.d_cold_eidx16, cvals);
            HIP_TRY(hipGetLastError());
        }
        const SpmmValArgs va{cs.d_flat, vals, cs.d_order, x16, hdr, d_Y, plan->N, D, dpad / 16, 0, plan->Nc + 1, plan->nw_eff, cs.nwg, cs.d_rbase, cs.d_rlist};
        HIP_TRY(launch_lds_val(va, dpad / 32, stream));
        if (cs.cold_tiles > 0) {
            const ColdValArgs ca{cs.d_cold_ptr, cs.d_cold_cols, cs.d_cold_mask, cvals, x16, hdr, d_Y, plan->N, D, plan->Nc + 1, plan->nw_eff, 0, dpad};
            hipLaunchKernelGGL(spmm_cold_val_kernel, dim3((unsigned)((plan->nw_eff + 3) / 4), (unsigned)((dpad + 63) / 64)), dim3(256), 0, stream, ca);
            HIP_TRY(hipGetLastError());
        }
        timer.stop();
        return wide_fallback();
    }
    // ---- edge values on the fused AGNN kernel's XCD-sliced walk (r03).  Where the fused pair's backward pass takes that walk (graphs
    //      without locality of their own, windows alike, an fp16 image of 16 - 64 MB: agnn_walk) the edge-valued SpMM is the same
    //      gather with less to do per tile, so it runs as that kernel with the score half switched off (AgnnArgs::valonly: w = 1,
    //      ef = the caller's values, their abs-max from this call's header): 76 % L2 hits and 5.8 GB of fabric reads instead of the
    //      per-window walk's 31 % and 12.2 GB on the Reddit shape at D = 64.  Same operand rounding and scales; the sums run in slice order.
    if (d_val && !d_staged && !block_of_wider && !d_gate && !relu && mode == 0 && dpad > 32 && dpad <= kMaxChunkDims && agnn_supported(plan, D)) {
        int ns = 0;
        const size_t need = workspace_bytes_for(plan->Nc, D) + agnn_partial_bytes(plan) + agnn_slice_bytes(plan, D);
        if (agnn_walk(plan, D, true, &ns) == kAgnnSliced && ns > 0 && ws_bytes >= need) {
            double* partial = reinterpret_cast<double*>(static_cast<char*>(ws) + workspace_bytes_for(plan->Nc, D));
            float* const ypart = reinterpret_cast<float*>(static_cast<char*>(ws) + workspace_bytes_for(plan->Nc, D) + agnn_partial_bytes(plan));
            AgnnArgs a{plan->d_wb_ptr, plan->d_order, plan->d_cols, plan->d_mask, plan->d_ebase, x16, hdr, nullptr, const_cast<float*>(d_val), const_cast<uint32_t*>(hdr) + 1, ypart, partial,
                       plan->N, plan->Nc, plan->row_off, dpad, D, pitch, plan->E, plan->rowptr, plan->d_bptr, plan->nbuckets, plan->nbuckets / ns, 0, plan->nw_eff, 0,
                       image_is_big(plan->Nc, pitch), ns, 1};
            {
                KernelTimer timer(plan, stream, "agnn_kernel (XCD-sliced, values only) + agnn_slice_sum_kernel");
                HIP_TRY((launch_agnn<4, true, 0>(dpad / 16, a, ns * ((plan->nw_eff + 3) / 4), stream)));
                const int64_t nsum = std::min<int64_t>(plan->N, (int64_t)plan->nw_eff * kWinRows) * D;   // (rows beyond the windows were zeroed above)
                const unsigned sg = (unsigned)std::min<int64_t>(2048, (nsum / 4 + 255) / 256 + 1);
                hipLaunchKernelGGL(agnn_slice_sum_kernel, dim3(sg), dim3(256), 0, stream, ypart, d_Y, nsum, (int64_t)plan->N * D, ns);
                HIP_TRY(hipGetLastError());
            }
            return wide_fallback();
        }
    }
    if (lds) {
        int total_chunks = 0;
        for (int i = 0; i < npass; ++i) total_chunks += passes[i].nchunks;
        const int accumulate = (d_W && total_chunks > 1) ? 1 : 0;
        if (accumulate) {
            if (relu) return fail(TCGNN_ERR_UNSUPPORTED, "tcgnn_spmm_gemm: ReLU cannot be fused when the product is accumulated over column passes");
            HIP_TRY(hipMemsetAsync(d_Y, 0, (size_t)plan->N * D_out * sizeof(float), stream));
        }
        const _Float16* x16_rows = nullptr;
        if (cold) {   // the remainder's gather walk reads the row-major image: staged behind the planar one, with its scale words
            const size_t image = workspace_bytes_for(plan->Nc, D);
            if (ws_bytes < 2 * image) return fail(TCGNN_ERR_WORKSPACE, "tcgnn_spmm: a plan with a cold remainder stages two images: %zu bytes, got %zu", 2 * image, ws_bytes);
            const uint32_t* hdr2; int dpad2, pitch2;
            const int rc = stage_features(plan, d_X, nullptr, D, static_cast<char*>(ws) + image, ws_bytes - image, stream, &hdr2, &x16_rows, &dpad2, &pitch2, false, d_gate, 0,
                                          false, hdr);
            if (rc) return rc;
        }
        const tcgnn_plan::CellStream& cs0 = plan->lds[lds_stream_of(passes[0].nt, passes[0].maxw)];
        KernelTimer timer(plan, stream, cold ? "spmm_lds_kernel + spmm_kernel (cold remainder)" :
                                        (cs0.flat_tpc ? (cs0.cold_tiles > 0 && !cs0.d_wcold_ptr ? "spmm_lds_flat_kernel + spmm_cold_planar_kernel (cold remainder)" : "spmm_lds_flat_kernel") : "spmm_lds_kernel"));
        for (int i = 0; i < npass; ++i) {
            const tcgnn_plan::CellStream& cs = plan->lds[lds_stream_of(passes[i].nt, passes[i].maxw)];
            if (cs.flat_tpc) {
                const bool has_cold = cs.cold_tiles > 0 && !cs.d_wcold_ptr;
                SpmmFlatArgs f{cs.d_flat, cs.d_order, x16, hdr, d_Y, plan->N, D, dpad / 16, passes[i].chunk0, plan->Nc + 1, plan->nw_eff, cs.nwg, g_lds_dbg,
                               has_cold ? 0 : relu, cs.d_rbase, cs.d_rlist, d_W, D_out, accumulate, g_lds_fill_quota, cs.d_wcold_ptr, cs.d_wcold};
                HIP_TRY(launch_flat_any(passes[i].maxw, passes[i].nt, cs.flat_tpc, f, passes[i].nchunks, stream));
                if (has_cold && !(g_lds_dbg & 16)) {
                    const int cd = lds_chunk_dims(passes[i].maxw);
                    const int col0 = passes[i].chunk0 * cd, ncols = std::min(passes[i].nchunks * cd, dpad - col0);
                    const ColdPlanarArgs ca{cs.d_cold_ptr, cs.d_cold_cols, cs.d_cold_mask, x16, hdr, d_Y, plan->N, D, plan->Nc + 1, plan->nw_eff, col0, ncols, relu};
                    hipLaunchKernelGGL(spmm_cold_planar_kernel, dim3((unsigned)((plan->nw_eff + 3) / 4), (unsigned)((ncols + 63) / 64)), dim3(256), 0, stream, ca);
                    HIP_TRY(hipGetLastError());
                }
                continue;
            }
            SpmmLdsArgs l{cs.d_cell_ptr, cs.d_cell_tiles, cs.d_order, x16, hdr, d_Y, plan->N, D, dpad / 16, passes[i].chunk0, plan->Nc + 1,
                          cs.nranges, plan->nw_eff, cs.nwg, g_lds_dbg, cold ? 0 : relu, cs.d_rbase, cs.d_rlist, d_W, D_out, accumulate, cs.d_parts};
            HIP_TRY(launch_lds_any(passes[i].maxw, passes[i].nt, l, passes[i].nchunks, stream));
        }
        if (cold) {
            const int pitch_r = x16_pitch(dpad);
            // (the per-tile metadata DMA fetches 16 edge-offset words too; binary SpMM never looks at them: the mask array stands in)
            SpmmArgs a{cold->d_cold_ptr, plan->d_order, cold->d_cold_cols, cold->d_cold_mask, reinterpret_cast<const int32_t*>(cold->d_cold_mask), x16_rows, nullptr, hdr, d_Y, plan->N, D, pitch_r, 0, plan->E,
                       plan->Nc + 1, relu, (int32_t)D, image_is_big(plan->Nc, pitch_r), nullptr, 0, 1};
            static const int cold_w4 = [] { const char* e = getenv("TCGNN_COLD_W4"); return e ? atoi(e) : 48; }();   // tiles per window from which 4 wavefronts share it (SBM Reddit shape, 25 cold tiles per window: 169 us with one wavefront, 202 with four)
            // (and a window with hundreds of cold tiles - a hub - is 0.25 us per tile of serial work for one wavefront)
            const int waves = (cold->cold_tiles >= (int64_t)cold_w4 * plan->nw_eff || cold->cold_max >= 512) ? 4 : 1;
            const int nfull = dpad / kMaxChunkDims, rem = (dpad % kMaxChunkDims) / 16;
            if (nfull) { a.chunk0 = 0; HIP_TRY(launch_spmm_any(false, waves, 8, a, plan->nw_eff, nfull, stream)); }
            if (rem) { a.chunk0 = nfull; HIP_TRY(launch_spmm_any(false, waves, rem, a, plan->nw_eff, 1, stream)); }
        }
        timer.stop();
        return wide_fallback();
    }
    SpmmArgs a{plan->d_wb_ptr, plan->d_order, plan->d_cols, plan->d_mask, plan->d_ebase, x16, d_val, hdr, d_Y, plan->N, D, pitch, 0, plan->E, plan->Nc + 1, relu, (int32_t)ld,
               image_is_big(plan->Nc, pitch), d_W, D_out, 0};
    if (d_W) a.ldy = D_out;
    const int nfull = dpad / kMaxChunkDims, rem = (dpad % kMaxChunkDims) / 16;
    // range-blocked walk when the fp16 image of X overflows L2 and the windows are long enough to cut
    const size_t x16_bytes = ((size_t)plan->Nc + 1) * pitch * sizeof(_Float16);
    // (a numbering with locality keeps the per-window walk: in XCD-contiguous order its co-resident workgroups share their gathered
    //  rows in L2 - 50-community Reddit shape, edge values: 0.92 ms against 1.18 ms range-blocked; the range-blocked walk is for
    //  graphs without it, where it wins by 1.4x)
    const bool blocked = !d_W && plan->nbuckets > 0 && mode != 1 && (mode == 2 || (x16_bytes > kBlockedMinBytes && windows_balanced(plan) && ranges_fit_l2(plan, x16_bytes) && !has_locality(plan) &&
                                                                                     // (edge values: two windows per wavefront instead of four; at Reddit's 30 MB image the per-window walk
                                                                                     //  in contiguous order is 6 % faster - 1.70 against 1.79 ms per call - so only images beyond the Infinity Cache's reach)
                                                                                     (!d_val || x16_bytes > ((size_t)64 << 20))));
    KernelTimer timer(plan, stream, blocked ? "spmm_blocked_kernel" : "spmm_kernel");
    if (blocked) {
        size_t range_bytes = kRangeTargetBytes;
        if (const char* e = getenv("TCGNN_RANGE_KB")) range_bytes = (size_t)atol(e) << 10;   // tuning experiments only
        int nranges = 1;
        while (nranges < plan->nbuckets && x16_bytes / nranges > range_bytes) nranges <<= 1;
        SpmmBlockedArgs b{a, plan->d_bptr, plan->nbuckets, plan->nbuckets / nranges, nranges, plan->nw_eff, 0};
        auto wgs = [&](int nt) {   // persistent grid = what is resident at once: LDS per workgroup (4 wavefronts: tile buffers,
                                   // pads, + the 4 KB A table) against 160 KB, and the register budget (4 or 2 workgroups per CU)
            const bool val = d_val != nullptr;
            const int maxw = blocked_maxw(nt, val);
            b.ngroups = (plan->nw_eff + maxw - 1) / maxw;
            const int lds_wg = 4 * (2 * nt * 1024 + kPadBytes + (val ? 1024 : 0)) + 4096;
            const int per_cu = std::max(1, std::min(nt <= 4 ? 4 : 2, (160 * 1024) / lds_wg));
            return std::min((b.ngroups + 3) / 4, plan->num_cus * per_cu);
        };
        if (nfull) { b.base.chunk0 = 0; const int n = wgs(8); HIP_TRY(launch_blocked_any(d_val != nullptr, 8, b, n, nfull, stream)); }
        if (rem) { b.base.chunk0 = nfull; const int n = wgs(rem); HIP_TRY(launch_blocked_any(d_val != nullptr, rem, b, n, 1, stream)); }
        timer.stop();
        return wide_fallback();
    }
    if (nfull) { a.chunk0 = 0; HIP_TRY(launch_spmm_any(d_val != nullptr, plan->waves, 8, a, plan->nw_eff, nfull, stream)); }
    if (rem) { a.chunk0 = nfull; HIP_TRY(launch_spmm_any(d_val != nullptr, plan->waves, rem, a, plan->nw_eff, 1, stream)); }
    timer.stop();
    return wide_fallback();
}

static bool agnn_supported(const tcgnn_plan* plan, int32_t D) {
    return plan && plan->canonical && D >= 1 && D <= kMaxChunkDims && plan->E >= 8;
}

static int run_agnn(const tcgnn_plan* plan, const float* d_X, const float* d_w, float* d_ef, uint32_t* d_absmax, float* d_Y,
                    float* d_dw, int32_t D, void* ws, size_t ws_bytes, void* stream_v, bool bwd) {
    const char* name = bwd ? "tcgnn_agnn_backward" : "tcgnn_agnn_forward";
    if (!plan || D < 1 || !d_w || !d_absmax || (bwd && !d_dw) || (plan->N > 0 && (!d_X || !d_Y)) || (plan->E > 0 && !d_ef))
        return fail(TCGNN_ERR_INVALID_ARG, "%s: null argument or D < 1", name);
    if (!agnn_supported(plan, D))
        return fail(TCGNN_ERR_UNSUPPORTED, "%s: needs a canonical plan, D <= %d and E >= 8 (canonical=%d, D=%d, E=%lld)", name,
                    kMaxChunkDims, plan->canonical, D, (long long)plan->E);
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    int nslices = 0;
    const int walk = agnn_walk(plan, D, bwd, &nslices);
    const bool sliced = walk == kAgnnSliced;
    const size_t need = workspace_bytes_for(plan->Nc, D) + agnn_partial_bytes(plan) + agnn_slice_bytes(plan, D);
    if (!ws || ws_bytes < need) return fail(TCGNN_ERR_WORKSPACE, "%s: workspace needs %zu bytes, got %zu", name, need, ws_bytes);
    if ((int64_t)plan->nw_eff * kWinRows < plan->N) {   // rows the caller's windows do not cover stay zero
        HIP_TRY(hipMemsetAsync(d_Y, 0, (size_t)plan->N * D * sizeof(float), stream));
        if (!bwd) HIP_TRY(hipMemsetAsync(d_ef, 0, (size_t)plan->E * sizeof(float), stream));
    }
    if (!bwd) HIP_TRY(hipMemsetAsync(d_absmax, 0, sizeof(uint32_t), stream));
    const uint32_t* hdr; const _Float16* x16; int dpad, pitch;
    const Guard gsd = guard_sddmm(D);
    int rc = stage_features(plan, d_X, nullptr, D, ws, ws_bytes, stream, &hdr, &x16, &dpad, &pitch, false, nullptr, 0, false, nullptr, &gsd);
    if (rc) return rc;
    double* partial = reinterpret_cast<double*>(static_cast<char*>(ws) + workspace_bytes_for(plan->Nc, D));
    if (plan->nw_eff == 0) {
        if (bwd) HIP_TRY(hipMemsetAsync(d_dw, 0, sizeof(float), stream));
        return TCGNN_OK;
    }
    AgnnArgs a{plan->d_wb_ptr, plan->d_order, plan->d_cols, plan->d_mask, plan->d_ebase, x16, hdr, d_w, d_ef, d_absmax, d_Y, partial,
               plan->N, plan->Nc, plan->row_off, dpad, D, pitch, plan->E, plan->rowptr, plan->d_bptr, plan->nbuckets, 0, 0, plan->nw_eff, 0,
               image_is_big(plan->Nc, pitch), 0, 0};
    const int nt = dpad / 16;
    const size_t x16_bytes = ((size_t)plan->Nc + 1) * pitch * sizeof(_Float16);
    float* const ypart = reinterpret_cast<float*>(static_cast<char*>(ws) + workspace_bytes_for(plan->Nc, D) + agnn_partial_bytes(plan));
    // The range-major variant (bit-compatible scores, sums in another order): slower than the per-window walk while the kernel
    // asked for every 128-byte line twice (r02: D = 64 1.87 vs 1.80 ms forward); with whole-line gathers (r03) its forward pass
    // is the fastest form at D = 64 (1.45-1.48 against 1.74 per-window, 1.53 sliced) - agnn_walk picks it there; mode 2 forces it.
    const bool blocked = plan->nbuckets > 0 && (g_spmm_mode == 2 || (g_spmm_mode == 0 && walk == kAgnnRangeMajor)) && x16_bytes > 0 && !a.big;
    int nwg = plan->nw_eff;
    {
        KernelTimer timer(plan, stream, (sliced && !blocked) ? "agnn_kernel (XCD-sliced) + agnn_slice_sum_kernel" : "agnn_kernel");
        hipError_t e;
        if (sliced && !blocked) {
            a.nslices = nslices;
            a.gsel = plan->nbuckets / nslices;
            a.y = ypart;
            nwg = nslices * ((plan->nw_eff + 3) / 4);
            e = bwd ? launch_agnn<4, true, 0>(nt, a, nwg, stream) : launch_agnn<4, false, 0>(nt, a, nwg, stream);
            if (e == hipSuccess) {
                const int64_t nsum = std::min<int64_t>(plan->N, (int64_t)plan->nw_eff * kWinRows) * D;   // (rows beyond the windows were zeroed above)
                const unsigned sg = (unsigned)std::min<int64_t>(2048, (nsum / 4 + 255) / 256 + 1);
                hipLaunchKernelGGL(agnn_slice_sum_kernel, dim3(sg), dim3(256), 0, stream, ypart, d_Y, nsum, (int64_t)plan->N * D, nslices);
                e = hipGetLastError();
            }
        } else if (blocked) {
            size_t range_bytes = 4 * kRangeTargetBytes;
            if (const char* env = getenv("TCGNN_RANGE_KB")) range_bytes = (size_t)atol(env) << 10;
            int nranges = 1;
            while (nranges < plan->nbuckets && x16_bytes / nranges > range_bytes) nranges <<= 1;
            a.nranges = nranges;
            a.gsel = plan->nbuckets / nranges;
            a.ngroups = (plan->nw_eff + kAgnnMaxW - 1) / kAgnnMaxW;
            const int lds_wg = 4 * agnn_wave_lds((nt + 1) / 2, bwd);
            const int per_cu = std::max(1, std::min(nt <= 4 ? 3 : 2, (160 * 1024) / lds_wg));
            nwg = std::min((a.ngroups + 3) / 4, plan->num_cus * per_cu);
            e = bwd ? launch_agnn<4, true, kAgnnMaxW>(nt, a, nwg, stream) : launch_agnn<4, false, kAgnnMaxW>(nt, a, nwg, stream);
        } else if (plan->waves == 4) {
            e = bwd ? launch_agnn<4, true, 0>(nt, a, nwg, stream) : launch_agnn<4, false, 0>(nt, a, nwg, stream);
        } else {
            e = bwd ? launch_agnn<1, true, 0>(nt, a, nwg, stream) : launch_agnn<1, false, 0>(nt, a, nwg, stream);
        }
        HIP_TRY(e);
    }
    // (the d_w correction of the patch: a double in header words 10-11, zeroed with the header by the staging pass)
    double* const dw_extra = reinterpret_cast<double*>(const_cast<uint32_t*>(hdr) + 10);
    if (g_range_guard >= 2) {
        // a few dirty rows (what training produces): the MFMA kernel above ran, the edges that touch them are recomputed here
        const PatchArgs pa{hdr, plan->rowptr, plan->col, plan->e2r, d_X, x16, pitch, d_ef, d_w, d_Y, d_absmax, dw_extra, plan->N, plan->Nc, D, plan->row_off, bwd ? 2 : 1, plan->E};
        // (many: the same launch does all the work in plain fp32 - wide_dense_body; one launch per call either way, returning at once
        //  unless the staged matrix is "wide")
        HIP_TRY(launch_wide_patch(pa, stream, partial, nwg));
    }
    if (bwd) {
        hipLaunchKernelGGL(agnn_reduce_kernel, dim3(1), dim3(kReduceThreads), 0, stream, partial, nwg, d_dw, g_range_guard >= 2 ? dw_extra : (const double*)nullptr);
        HIP_TRY(hipGetLastError());
    }
    return TCGNN_OK;
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int tcgnn_agnn_supported(const tcgnn_plan* plan, int32_t D) { return agnn_supported(plan, D) ? 1 : 0; }

int tcgnn_agnn_forward(const tcgnn_plan* plan, const float* d_X, const float* d_w, float* d_ef, uint32_t* d_ef_absmax, float* d_Y,
                       int32_t D, void* ws, size_t ws_bytes, void* stream) {
    return run_agnn(plan, d_X, d_w, d_ef, d_ef_absmax, d_Y, nullptr, D, ws, ws_bytes, stream, false);
}

int tcgnn_agnn_backward(const tcgnn_plan* plan, const float* d_dY, const float* d_w, const float* d_ef, const uint32_t* d_ef_absmax,
                        float* d_G, float* d_dw, int32_t D, void* ws, size_t ws_bytes, void* stream) {
    return run_agnn(plan, d_dY, d_w, const_cast<float*>(d_ef), const_cast<uint32_t*>(d_ef_absmax), d_G, d_dw, D, ws, ws_bytes, stream, true);
}

int tcgnn_plan_destroy(tcgnn_plan* plan) {
    if (!plan) return TCGNN_OK;
    (void)hipFree(plan->d_wb_ptr); (void)hipFree(plan->d_order); (void)hipFree(plan->d_cols);
    (void)hipFree(plan->d_mask); (void)hipFree(plan->d_ebase); (void)hipFree(plan->d_bptr);
    for (auto& cs : plan->lds) {
        (void)hipFree(cs.d_cell_ptr); (void)hipFree(cs.d_cell_tiles); (void)hipFree(cs.d_order); (void)hipFree(cs.d_rbase); (void)hipFree(cs.d_rlist);
        (void)hipFree(cs.d_cold_ptr); (void)hipFree(cs.d_cold_cols); (void)hipFree(cs.d_cold_mask); (void)hipFree(cs.d_parts); (void)hipFree(cs.d_flat);
        (void)hipFree(cs.d_wcold_ptr); (void)hipFree(cs.d_wcold); (void)hipFree(cs.d_eidx); (void)hipFree(cs.d_cold_eidx); (void)hipFree(cs.d_eidx16); (void)hipFree(cs.d_cold_eidx16);
    }
    (void)hipFree(plan->d_xwb_ptr); (void)hipFree(plan->d_xcols); (void)hipFree(plan->d_xmask); (void)hipFree(plan->d_xeidx);
    for (hipEvent_t e : plan->ev) (void)hipEventDestroy(e);
    delete plan;
    return TCGNN_OK;
}

int tcgnn_plan_create_sharded(const int32_t* d_nodePointer, const int32_t* d_edgeList,
                              const int32_t* d_blockPartition, const int32_t* d_edgeToColumn,
                              const int32_t* d_edgeToRow, int32_t num_rows, int32_t num_cols,
                              int32_t row_offset, int64_t num_edges, int32_t num_windows,
                              void* stream_v, tcgnn_plan** plan_out) {
    if (!plan_out) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_plan_create: plan_out is null");
    *plan_out = nullptr;
    const int32_t num_nodes = num_rows;
    if (num_cols < 0 || row_offset < 0 || (int64_t)row_offset + num_rows > (int64_t)num_cols)
        return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_plan_create: rows [%d, %d) do not fit in %d feature rows", row_offset, row_offset + num_rows, num_cols);
    if (num_nodes < 0 || num_edges < 0 || num_windows < 0 || !d_nodePointer ||
        (num_windows > 0 && !d_blockPartition) || (num_edges > 0 && (!d_edgeList || !d_edgeToColumn || !d_edgeToRow)))
        return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_plan_create: null array or negative size");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    tcgnn_plan* p = new (std::nothrow) tcgnn_plan();
    if (!p) return fail(TCGNN_ERR_OOM, "tcgnn_plan_create: host allocation failed");
    p->N = num_nodes; p->Nc = num_cols; p->row_off = row_offset; p->E = num_edges; p->nw = num_windows;
    p->nw_eff = (int32_t)std::min<int64_t>(num_windows, ((int64_t)num_nodes + kWinRows - 1) / kWinRows);
    p->rowptr = d_nodePointer; p->col = d_edgeList; p->bp = d_blockPartition; p->e2c = d_edgeToColumn; p->e2r = d_edgeToRow;
    const int nw = p->nw_eff;
    std::vector<int32_t> bp((size_t)std::max(nw, 1));
    auto bail = [&](int rc) { tcgnn_plan_destroy(p); return rc; };
    if (nw > 0) {
        uint32_t* d_maxdeg = nullptr;   // the longest row: what the range guard's bound follows (guard_spmm)
        uint32_t h_maxdeg = 0;
        hipError_t e = hipMalloc(&d_maxdeg, sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemsetAsync(d_maxdeg, 0, sizeof(uint32_t), stream);
        if (e == hipSuccess && num_nodes > 0) {
            hipLaunchKernelGGL(max_degree_kernel, dim3((unsigned)std::min<int64_t>(1024, ((int64_t)num_nodes + 255) / 256)), dim3(256), 0, stream, d_nodePointer, num_nodes, d_maxdeg);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&h_maxdeg, d_maxdeg, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(bp.data(), d_blockPartition, (size_t)nw * sizeof(int32_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(d_maxdeg);
        if (e != hipSuccess) return bail(fail(TCGNN_ERR_HIP, "read blockPartition: %s", hipGetErrorString(e)));
        p->max_degree = (int32_t)std::min<uint32_t>(h_maxdeg, 0x7fffffffu);
    }
    std::vector<int64_t> wb_ptr((size_t)nw + 1, 0);
    for (int w = 0; w < nw; ++w) {
        if (bp[(size_t)w] < 0) return bail(fail(TCGNN_ERR_BAD_GRAPH, "blockPartition[%d] = %d is negative", w, bp[(size_t)w]));
        p->tc_blocks += bp[(size_t)w];
        wb_ptr[(size_t)w + 1] = wb_ptr[(size_t)w] + (bp[(size_t)w] + 3) / 4;
        p->max_wb = std::max<int64_t>(p->max_wb, (bp[(size_t)w] + 3) / 4);
    }
    p->total_wb = wb_ptr[(size_t)nw];
    p->h_bp.assign(bp.begin(), bp.begin() + nw);
    std::vector<int32_t> order((size_t)std::max(nw, 1));
    std::iota(order.begin(), order.begin() + nw, 0);
    std::stable_sort(order.begin(), order.begin() + nw, [&](int32_t x, int32_t y) { return bp[(size_t)x] > bp[(size_t)y]; });
    // Block -> window map of the per-window gather walks.  Heaviest first keeps a hub window from starting last; but when no window
    // is far above the mean the order is free, and then locality decides: workgroup b runs on XCD b % 8 (observed dispatch, used for
    // speed only), so XCD x takes the x-th contiguous eighth of the windows in their own order - the workgroups resident on one
    // XCD at one time are neighbours in the graph's numbering and share their gathered rows in that XCD's L2 (communities).
    {
        int64_t mx = 0;
        for (int w = 0; w < nw; ++w) mx = std::max<int64_t>(mx, bp[(size_t)w]);
        static const int order_mode = [] { const char* e = getenv("TCGNN_ORDER"); return e ? atoi(e) : 0; }();   // 0 automatic, 1 heaviest first, 2 XCD-contiguous
        const bool balanced = nw > 0 && mx * nw <= 4 * std::max<int64_t>(p->tc_blocks, 1);
        // A few hubs over an otherwise even graph (communities + hubs): the K windows more than 4x the mean start first, heaviest first
        // (the round-robin dispatch spreads them over the XCDs), the rest follows in XCD-contiguous order.  A continuous skew
        // (R-MAT: the weight falls with the id, the rest is not even either) keeps heaviest-first throughout.
        int K = 0;
        if (!balanced && nw >= 64) {
            const int64_t mean_x4 = 4 * std::max<int64_t>(p->tc_blocks, 1) / nw + 1;
            while (K < nw && bp[(size_t)order[(size_t)K]] > mean_x4) ++K;
            int64_t rest = 0, rest_max = 0;
            for (int q = K; q < nw; ++q) { rest += bp[(size_t)order[(size_t)q]]; rest_max = std::max<int64_t>(rest_max, bp[(size_t)order[(size_t)q]]); }
            if (K > nw / 16 || rest_max * (int64_t)(nw - K) > 3 * std::max<int64_t>(rest, 1)) K = -1;   // not "a few hubs": keep heaviest-first
        }
        if (order_mode == 2 || (order_mode == 0 && nw >= 64 && (balanced || K > 0))) {
            if (K < 0 || order_mode == 2) K = order_mode == 2 ? 0 : K;
            std::vector<int32_t> rest;                                   // the windows behind the hubs, in their own order
            {
                std::vector<char> is_hub((size_t)nw, 0);
                for (int q = 0; q < K; ++q) is_hub[(size_t)order[(size_t)q]] = 1;
                for (int w = 0; w < nw; ++w) if (!is_hub[(size_t)w]) rest.push_back(w);
            }
            // every XCD takes one contiguous eighth: the eighths must weigh about the same (a degree that falls with the id -
            // R-MAT - would hand XCD 0 the heavy end: ogbn-products shape 3.96 -> 4.54 ms), else heaviest-first stays
            bool even = true;
            if (order_mode != 2) {
                int64_t part[8] = {0}, all = 0;
                for (size_t q = 0; q < rest.size(); ++q) { part[q * 8 / rest.size()] += bp[(size_t)rest[q]]; all += bp[(size_t)rest[q]]; }
                for (int x = 0; x < 8; ++x) even = even && part[x] * 8 <= all + all / 8;
            }
            if (even) {
            int cnt[8] = {0}, start[9] = {0}, seen[8] = {0};
            for (int b = K; b < nw; ++b) ++cnt[b % 8];                   // positions XCD x gets behind the hubs
            for (int x = 0; x < 8; ++x) start[x + 1] = start[x] + cnt[x];
            for (int b = K; b < nw; ++b) { const int x = b % 8; order[(size_t)b] = rest[(size_t)(start[x] + seen[x]++)]; }
            }
        }
    }
    p->waves = (nw > 0 && p->total_wb >= (int64_t)6 * nw) ? 4 : 1;

    const size_t n_wb = (size_t)std::max<int64_t>(p->total_wb, 1);
    const size_t b_ptr = ((size_t)nw + 1) * sizeof(int64_t), b_ord = (size_t)std::max(nw, 1) * sizeof(int32_t);
    const size_t b_cols = n_wb * kWbCols * sizeof(int32_t), b_mask = n_wb * kWinRows * sizeof(uint32_t), b_eb = n_wb * kWinRows * sizeof(int32_t);
    int32_t* d_flags = nullptr;
    hipError_t e = hipMalloc(&p->d_wb_ptr, b_ptr);
    if (e == hipSuccess) e = hipMalloc(&p->d_order, b_ord);
    if (e == hipSuccess) e = hipMalloc(&p->d_cols, b_cols);
    if (e == hipSuccess) e = hipMalloc(&p->d_mask, b_mask);
    if (e == hipSuccess) e = hipMalloc(&p->d_ebase, b_eb);
    if (e == hipSuccess) e = hipMalloc(&d_flags, 2 * sizeof(int32_t));
    if (e != hipSuccess) { (void)hipFree(d_flags); return bail(fail(e == hipErrorOutOfMemory ? TCGNN_ERR_OOM : TCGNN_ERR_HIP, "plan allocation (%zu bytes): %s", b_ptr + b_ord + b_cols + b_mask + b_eb, hipGetErrorString(e))); }
    p->bytes = b_ptr + b_ord + b_cols + b_mask + b_eb;
    int32_t flags[2] = {0, 0};
    e = hipMemcpyAsync(p->d_wb_ptr, wb_ptr.data(), b_ptr, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess && nw > 0) e = hipMemcpyAsync(p->d_order, order.data(), (size_t)nw * sizeof(int32_t), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_flags, 0, 2 * sizeof(int32_t), stream);
    if (e == hipSuccess && nw > 0) {
        hipLaunchKernelGGL(pack_kernel, dim3((unsigned)nw), dim3(256), 0, stream, d_nodePointer, d_edgeList, d_edgeToColumn,
                           d_edgeToRow, p->d_wb_ptr, num_rows, num_cols, p->d_cols, p->d_mask, p->d_ebase, d_flags);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(flags, d_flags, sizeof flags, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream); // host vectors above must outlive the copies
    (void)hipFree(d_flags);
    if (e != hipSuccess) return bail(fail(TCGNN_ERR_HIP, "plan build: %s", hipGetErrorString(e)));
    if (flags[0]) return bail(fail(TCGNN_ERR_BAD_GRAPH, "edgeToColumn / edgeToRow / edgeList hold ids outside the window, blockPartition or node range"));
    p->canonical = flags[1] ? 0 : 1;
    if (nw > 0) {
        unsigned long long* d_loc = nullptr;
        unsigned long long h_loc[2] = {0, 0};
        e = hipMalloc(&d_loc, sizeof h_loc);
        if (e == hipSuccess) e = hipMemsetAsync(d_loc, 0, sizeof h_loc, stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(locality_kernel, dim3((unsigned)nw), dim3(256), 0, stream, p->d_wb_ptr, p->d_cols, nw, num_cols, row_offset, std::max(num_cols / 16, kWinRows), d_loc);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h_loc, d_loc, sizeof h_loc, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(d_loc);
        if (e != hipSuccess) return bail(fail(TCGNN_ERR_HIP, "plan build (locality): %s", hipGetErrorString(e)));
        p->near_frac = h_loc[1] ? (double)h_loc[0] / (double)h_loc[1] : 0.0;
        if (const char* v = getenv("TCGNN_VERBOSE")) if (atoi(v) > 0) fprintf(stderr, "[tcgnn] plan: %.0f %% of the condensed columns lie within num_cols / 16 rows of their window\n", 100.0 * p->near_frac);
    }
    {   // column buckets for the range-blocked SpMM: only when windows are long (>= 2 tiles per bucket on
        // average) and numerous enough to fill the chip with one wavefront per 4 windows (below)
        hipDeviceProp_t prop;
        int devid = 0;
        if (hipGetDevice(&devid) == hipSuccess && hipGetDeviceProperties(&prop, devid) == hipSuccess) p->num_cus = prop.multiProcessorCount;
        int nb = 8;
        while (nb < 128 && (int64_t)num_cols / nb > 4096) nb <<= 1;
        // wide column spaces with short windows (a row shard of a multi-GPU graph: Reddit's 243 tiles per window spread over
        // N x 232 965 columns): fewer, longer buckets rather than no table - without it the shard falls back to the per-window
        // walk (measured 1.77 ms against 0.87 ms for the unsharded graph)
        while (nb > 8 && p->total_wb < (int64_t)g_bucket_min_tiles * nb * nw) nb >>= 1;
        if (nw >= 4 * p->num_cus && p->total_wb >= (int64_t)g_bucket_min_tiles * nb * nw) {
            p->nbuckets = nb;
            p->bucket_rows = (int32_t)(((int64_t)num_cols + nb - 1) / nb);
            if (p->bucket_rows < 1) p->bucket_rows = 1;
            const size_t b_bp = (size_t)nw * (nb + 1) * sizeof(uint32_t);
            e = hipMalloc(&p->d_bptr, b_bp);
            if (e == hipSuccess) {
                const int64_t total = (int64_t)nw * (nb + 1);
                hipLaunchKernelGGL(bucket_ptr_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p->d_wb_ptr, p->d_cols, nw, nb,
                                   p->bucket_rows, p->d_bptr);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipStreamSynchronize(stream);
            }
            if (e != hipSuccess) return bail(fail(TCGNN_ERR_HIP, "bucket table: %s", hipGetErrorString(e)));
            p->bytes += b_bp;
        }
    }
    // Cell stream of the LDS-resident column-range SpMM (tcgnn_lds_spmm.inc) when the time models pick that kernel for a
    // 64-column matrix: built now rather than inside the first call.  Other widths decide, and build, at their first call.
    // TCGNN_LDS_AUTO=0 disables the automatic choice.
    if (lds_chosen(p, 64)) {
        LdsPass passes[2];
        const int np = lds_passes(64, passes);
        for (int i = 0; i < np; ++i) {
            const int rc = build_lds_cells(p, stream, lds_stream_of(passes[i].nt, passes[i].maxw));
            if (rc == TCGNN_ERR_OOM) { p->lds_choice[4] = 0; break; }   // (the gather walks need no stream)
            if (rc) return bail(rc);
        }
    }
    *plan_out = p;
    return TCGNN_OK;
}

int tcgnn_plan_create(const int32_t* d_nodePointer, const int32_t* d_edgeList,
                      const int32_t* d_blockPartition, const int32_t* d_edgeToColumn,
                      const int32_t* d_edgeToRow, int32_t num_nodes, int64_t num_edges,
                      int32_t num_windows, void* stream, tcgnn_plan** plan_out) {
    return tcgnn_plan_create_sharded(d_nodePointer, d_edgeList, d_blockPartition, d_edgeToColumn, d_edgeToRow, num_nodes,
                                     num_nodes, 0, num_edges, num_windows, stream, plan_out);
}

int tcgnn_plan_get_info(const tcgnn_plan* plan, tcgnn_plan_info* info) {
    if (!plan || !info) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_plan_get_info: null argument");
    info->num_nodes = plan->N; info->num_windows = plan->nw; info->num_edges = plan->E;
    info->tc_blocks = plan->tc_blocks; info->wide_blocks = plan->total_wb; info->plan_bytes = (int64_t)plan->bytes;
    info->canonical = plan->canonical; info->waves_per_window = plan->waves;
    info->column_buckets = plan->nbuckets; info->lds_ranges = 0;
    for (int i = 0; i < kLdsStreams; ++i) if (plan->lds[i].nranges > info->lds_ranges) info->lds_ranges = plan->lds[i].nranges;   // finest stream built so far
    return TCGNN_OK;
}

int tcgnn_plan_prepare(tcgnn_plan* plan, int32_t D, void* stream_v) {
    if (!plan || D < 1) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_plan_prepare: null plan or D < 1");
    if (plan->nw_eff <= 0 || plan->N == 0) return TCGNN_OK;
    const int dpad = round_up(D, 16);
    if (!(g_spmm_mode == 3 || (g_spmm_mode == 0 && plan->total_wb > kSmallMaxTiles && lds_chosen(plan, dpad)))) return TCGNN_OK;   // the gather walks need nothing built
    if ((int64_t)(dpad / 16) * ((int64_t)plan->Nc + 1) * 32 >= ((int64_t)1 << 32)) return TCGNN_OK;
    LdsPass passes[2];
    const int np = lds_passes(dpad, passes);
    for (int i = 0; i < np; ++i) {
        const int slot = lds_stream_of(passes[i].nt, passes[i].maxw);
        if (plan->lds[slot].nranges > 0) continue;
        const int rc = build_lds_cells(plan, static_cast<hipStream_t>(stream_v), slot);
        if (rc && g_spmm_mode == 3) return rc;
        if (rc && dpad / 16 <= 64) plan->lds_choice[dpad / 16] = 0;   // (as the hot path would: no memory for the stream -> the gather walks)
    }
    return TCGNN_OK;
}

int tcgnn_set_range_guard(int32_t level) {
    if (level < 0 || level > 2) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_set_range_guard: 0 (off), 1 (SpMM operators, default) or 2 (every operator)");
    g_range_guard = level;
    return TCGNN_OK;
}

int tcgnn_range_mode(const void* d_workspace, void* stream_v, int32_t* wide_x, int32_t* wide_val) {
    if (!d_workspace || !wide_x) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_range_mode: null argument");
    uint32_t h[9];
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    HIP_TRY(hipMemcpyAsync(h, d_workspace, sizeof(h), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    auto spread = [&](int k, int& emax) {
        emax = (int)(h[k] >> 23);
        if (h[k + 2] == 0u || h[k] == 0u || h[k] >= 0x7f800000u) return false;
        return emax - (int)((0x7f800000u - h[k + 2]) >> 23) > 28;
    };
    auto clog2 = [](uint32_t k) { int c = 0; while (c < 32 && (1ull << c) < k) ++c; return c; };   // (the host mirror of range_is_wide / range_is_wide_val)
    int ex = 0, ea = 0;
    const bool sx = spread(0, ex), sa = spread(1, ea);
    *wide_x = (sx && h[4] != 0u && h[6] != 0u && (int)h[7] * (ex - 127) >= 29 - clog2(std::min(h[4], h[6]))) ? 1 : 0;
    if (*wide_x && h[7] == 2u && h[8] <= kSparseRows) *wide_x = 2;   // (SDDMM / fused AGNN with a few dirty rows: the MFMA kernel + wide_patch_kernel)
    if (wide_val) {
        const uint32_t k = (sa || h[6] >= h[5]) ? h[5] : std::max(h[6], 1u);
        *wide_val = ((sx || sa) && h[5] != 0u && h[0] != 0u && h[1] != 0u && (ex - 127) + (ea - 127) >= 28 - clog2(k)) ? 1 : 0;
    }
    return TCGNN_OK;
}

int tcgnn_set_spmm_mode(int32_t mode) {
    if (mode < 0 || mode > 4) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_set_spmm_mode: 0 (auto), 1 (plain), 2 (range-blocked), 3 (LDS-resident ranges) or 4 (single-launch fp32 kernel)");
    g_spmm_mode = mode;
    return TCGNN_OK;
}

const char* tcgnn_plan_last_kernel(const tcgnn_plan* plan) { return plan ? plan->last_kernel.load(std::memory_order_relaxed) : ""; }

int tcgnn_plan_set_timing(tcgnn_plan* plan, int32_t max_calls) {
    if (!plan || max_calls < 0) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_plan_set_timing: bad argument");
    for (hipEvent_t e : plan->ev) (void)hipEventDestroy(e);
    plan->ev.clear();
    plan->ev_used = 0;
    for (int i = 0; i < 2 * max_calls; ++i) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        plan->ev.push_back(e);
    }
    return TCGNN_OK;
}

int tcgnn_plan_read_timing(tcgnn_plan* plan, float* ms_out, int32_t capacity, int32_t* count) {
    if (!plan || !count || (capacity > 0 && !ms_out)) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_plan_read_timing: null argument");
    int n = 0;
    const int used = std::min(plan->ev_used.load(), (int)plan->ev.size() / 2);
    for (int i = 0; i < used && n < capacity; ++i) {
        HIP_TRY(hipEventSynchronize(plan->ev[2 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&ms_out[n], plan->ev[2 * i], plan->ev[2 * i + 1]));
        ++n;
    }
    *count = n;
    plan->ev_used = 0;
    return TCGNN_OK;
}

size_t tcgnn_workspace_bytes(const tcgnn_plan* plan, int32_t D) {
    if (!plan || D < 1) return 0;
    // (the fused AGNN calls' reduction slots and the partial score streams of the LDS-resident SDDMM ride along)
    // ... and the second (row-major) image of a plan whose LDS-resident walk leaves a cold remainder to the gather walk
    const size_t image = workspace_bytes_for(plan->Nc, D);
    // (before the width's streams exist the answer is the conservative one; once built, only an ORDINARY stream with a cold
    //  remainder stages the second image - a flat stream's remainder reads the planar one)
    bool two_images = plan->nw_eff > 0 && (g_spmm_mode == 3 || (g_spmm_mode == 0 && lds_chosen(plan, round_up(D, 16))));
    if (two_images) {
        LdsPass passes[2];
        const int np = lds_passes(round_up(D, 16), passes);
        bool all_built = np > 0, cold_rows = false;
        for (int i = 0; i < np; ++i) {
            const tcgnn_plan::CellStream& ci = plan->lds[lds_stream_of(passes[i].nt, passes[i].maxw)];
            all_built = all_built && ci.nranges > 0;
            cold_rows = cold_rows || (ci.nranges > 0 && !ci.flat_tpc && ci.cold_tiles > 0);
        }
        if (all_built && !cold_rows) two_images = false;
    }
    // (the edge-valued LDS-resident walk keeps its per-call slot values behind the image, once its stream exists: tcgnn_lds_val.inc)
    const size_t vals = (round_up(D, 16) % 64 == 0 && round_up(D, 16) <= 2 * kMaxChunkDims) ? val_stream_bytes(plan) : (size_t)0;
    return image + std::max({agnn_partial_bytes(plan) + agnn_slice_bytes(plan, D), two_images ? image : (size_t)0, vals});
}

int tcgnn_spmm(const tcgnn_plan* plan, const float* d_X, float* d_Y, int32_t D, void* ws, size_t ws_bytes, void* stream) {
    return run_spmm(plan, d_X, nullptr, d_Y, D, ws, ws_bytes, stream);
}

int tcgnn_spmm_fused(const tcgnn_plan* plan, const float* d_X, const float* d_gate, float* d_Y, int32_t D, int32_t flags,
                     void* ws, size_t ws_bytes, void* stream) {
    if (flags & ~TCGNN_FUSE_RELU) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_spmm_fused: unknown flag bits 0x%x", flags & ~TCGNN_FUSE_RELU);
    return run_spmm(plan, d_X, nullptr, d_Y, D, ws, ws_bytes, stream, (flags & TCGNN_FUSE_RELU) ? 1 : 0, d_gate);
}

int tcgnn_spmm_gemm(const tcgnn_plan* plan, const float* d_X, const float* d_W, float* d_Y, int32_t D_in, int32_t D_out, int32_t flags,
                    void* ws, size_t ws_bytes, void* stream) {
    if (flags & ~TCGNN_FUSE_RELU) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_spmm_gemm: unknown flag bits 0x%x", flags & ~TCGNN_FUSE_RELU);
    if (!d_W || D_out < 1) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_spmm_gemm: null W or D_out < 1");
    if (D_in > kMaxChunkDims || D_out > kMaxChunkDims)
        return fail(TCGNN_ERR_UNSUPPORTED, "tcgnn_spmm_gemm: the fused dense update covers D_in, D_out <= %d (got %d -> %d)", kMaxChunkDims, D_in, D_out);
    return run_spmm(plan, d_X, nullptr, d_Y, D_in, ws, ws_bytes, stream, (flags & TCGNN_FUSE_RELU) ? 1 : 0, nullptr, nullptr, 0, false, d_W, D_out);
}

int tcgnn_x16_pitch(int32_t D) { return D < 1 ? 0 : x16_pitch(round_up(D, 16)); }

int tcgnn_stage_absmax(const float* d_X, int64_t n, uint32_t* d_word, void* stream_v) {
    if (n < 0 || (n > 0 && !d_X) || !d_word) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_stage_absmax: null argument");
    if (n == 0) return TCGNN_OK;
    const int grid = absmax_grid(n);
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(kAbsmaxThreads), 0, static_cast<hipStream_t>(stream_v), d_X, n, d_word, (uint32_t*)nullptr, 0u, 0u);
    HIP_TRY(hipGetLastError());
    return TCGNN_OK;
}

int tcgnn_stage_rows(const float* d_X, int32_t rows, int32_t D, const uint32_t* d_absmax_word, void* d_dst, void* stream_v) {
    if (rows < 0 || D < 1 || (rows > 0 && !d_X) || !d_absmax_word || !d_dst) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_stage_rows: bad argument");
    if ((reinterpret_cast<uintptr_t>(d_dst) & 15) != 0) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_stage_rows: destination must be 16-byte aligned");
    const int dpad = round_up(D, 16), pitch = x16_pitch(dpad);
    const int64_t chunks = ((int64_t)rows + 1) * (dpad / 8);
    const unsigned cgrid = (unsigned)((chunks + 255) / 256);
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    const bool vec = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_X) & 15) == 0);
    _Float16* dst = static_cast<_Float16*>(d_dst);
    if (vec) hipLaunchKernelGGL((convert_kernel<true>), dim3(cgrid), dim3(256), 0, stream, d_X, rows, D, dpad, pitch, dst, d_absmax_word, (const float*)nullptr);
    else     hipLaunchKernelGGL((convert_kernel<false>), dim3(cgrid), dim3(256), 0, stream, d_X, rows, D, dpad, pitch, dst, d_absmax_word, (const float*)nullptr);
    HIP_TRY(hipGetLastError());
    return TCGNN_OK;
}

int tcgnn_spmm_staged(const tcgnn_plan* plan, const void* d_image, float* d_Y, int32_t D, void* stream) {
    if (!d_image || (reinterpret_cast<uintptr_t>(d_image) & 255)) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_spmm_staged: the image must be 256-byte aligned");
    // Every MFMA kernel opens with the range guard's test on header words 2, 4, 6, 7; a caller-staged image carries no range words
    // (never "wide": there is no fp32 X to fall back to).  The reserved words 1 .. 7 are cleared here, on the caller's stream, so a
    // header a caller left uninitialised beyond word 0 cannot make the kernels return early with Y unwritten.
    if (hipMemsetAsync(static_cast<char*>(const_cast<void*>(d_image)) + 4, 0, 28, static_cast<hipStream_t>(stream)) != hipSuccess)
        return fail(TCGNN_ERR_HIP, "tcgnn_spmm_staged: clearing the reserved header words failed");
    return run_spmm(plan, nullptr, nullptr, d_Y, D, nullptr, 0, stream, 0, nullptr, d_image);
}

int tcgnn_spmm_val(const tcgnn_plan* plan, const float* d_X, const float* d_edge_val, float* d_Y, int32_t D,
                   void* ws, size_t ws_bytes, void* stream) {
    if (plan && plan->E > 0 && !d_edge_val) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_spmm_val: edge values are null");
    return run_spmm(plan, d_X, plan && plan->E > 0 ? d_edge_val : nullptr, d_Y, D, ws, ws_bytes, stream);
}

int tcgnn_sddmm(const tcgnn_plan* plan, const float* d_X, float* d_ef, int32_t D, void* ws, size_t ws_bytes, void* stream_v) {
    if (!plan || D < 1 || (plan->N > 0 && !d_X) || (plan->E > 0 && !d_ef)) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_sddmm: null argument or D < 1");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    if (plan->E == 0 || plan->N == 0) return TCGNN_OK;
    if (!plan->canonical) {
        hipLaunchKernelGGL(sddmm_csr_kernel, dim3((unsigned)((plan->N + 3) / 4)), dim3(256), 0, stream, plan->rowptr, plan->col, d_X, d_ef, plan->N, D, plan->row_off);
        HIP_TRY(hipGetLastError());
        return TCGNN_OK;
    }
    if ((int64_t)plan->nw_eff * kWinRows < plan->N) HIP_TRY(hipMemsetAsync(d_ef, 0, (size_t)plan->E * sizeof(float), stream));
    const uint32_t* hdr; const _Float16* x16; int dpad, pitch;
    const Guard gsd = guard_sddmm(D);
    int rc = stage_features(plan, d_X, nullptr, D, ws, ws_bytes, stream, &hdr, &x16, &dpad, &pitch, false, nullptr, 0, false, nullptr, &gsd);
    if (rc) return rc;
    SddmmArgs a{plan->d_wb_ptr, plan->d_order, plan->d_cols, plan->d_mask, plan->d_ebase, x16, hdr, d_ef, plan->N, plan->Nc, plan->row_off, dpad, pitch, plan->rowptr, plan->d_bptr, plan->nbuckets, 0, 0, plan->nw_eff, image_is_big(plan->Nc, pitch), 0};
    const int ks = (dpad + 31) / 32;
    KernelTimer timer(plan, stream, ks <= 4 ? "sddmm_kernel" : "sddmm_wide_kernel");
    const size_t x16_bytes = ((size_t)plan->Nc + 1) * pitch * sizeof(_Float16);
    // Range-major walk (bit-identical results).  With the outputs staged per row the loop is bound by the gather again,
    // and keeping it inside ~4 MB column ranges wins on the Reddit shape: D=16 1.14 -> 1.07 ms, D=32 1.38 -> 1.14,
    // D=64 1.74 -> 1.66, D=128 3.37 -> 3.26.  No accumulators live across ranges, so ranges are 4x the SpMM's.
    const bool blocked = ks <= 4 && plan->nbuckets > 0 && g_spmm_mode != 1 && (g_spmm_mode == 2 || (x16_bytes > kBlockedMinBytes && windows_balanced(plan) && ranges_fit_l2(plan, x16_bytes) && !has_locality(plan)));
    hipError_t e;
    if (blocked) {
        // (r03, whole-line gathers: D = 64 1.26 / 1.24 ms at 4 / 8 MB ranges, 1.36 at 2 MB; D = 128 - an image of 60 MB - 2.33 at 2 MB,
        //  2.58 at 4 MB, 3.5 per-window; with XCD affinity 2.01 at 2 or 4 MB)
        size_t range_bytes = x16_bytes > ((size_t)32 << 20) ? 2 * kRangeTargetBytes : 4 * kRangeTargetBytes;
        if (const char* env = getenv("TCGNN_RANGE_KB")) range_bytes = (size_t)atol(env) << 10;
        int nranges = 1;
        while (nranges < plan->nbuckets && x16_bytes / nranges > range_bytes) nranges <<= 1;
        a.nranges = nranges;
        a.gsel = plan->nbuckets / nranges;
        const int lds_wg = 4 * sddmm_wave_lds(ks);
        const int per_cu = std::max(1, std::min(4, (160 * 1024) / lds_wg));
        const int64_t items = (int64_t)nranges * plan->nw_eff;
        int nwg = (int)std::min<int64_t>((items + 3) / 4, (int64_t)plan->num_cus * per_cu);
        // XCD affinity (sddmm_kernel; TCGNN_SDDMM_XCD=0 switches it off, read per call: tests compare the two).  Reddit shape:
        // D = 128 2.32 -> 2.01 ms, D = 64 1.36 -> 1.33, D = 16 / 32 -1 .. -2.5 %; before the whole-line gathers it returned nothing.
        const char* const xenv = getenv("TCGNN_SDDMM_XCD");
        // (like the fused kernel's sliced walk it wants every window's tiles spread evenly over the ranges: on the calibrated SBM graph -
        //  22.5 % of a window's edges inside its own community, near_frac 0.3 - the XCD that owns a window's community holds the others
        //  up, 1.43 -> 2.11 ms at D = 64, where an XCD has ONE range; with four ranges per XCD, spread over the graph, the load evens
        //  out again: D = 128 2.48 -> 2.25 ms there; TCGNN_SDDMM_XCD=2 forces it)
        const int xknob = xenv ? atoi(xenv) : 1;
        if (xknob && (xknob >= 2 || plan->near_frac <= 0.2 || nranges >= 4 * kXcdCount) && nranges % kXcdCount == 0 && nwg >= kXcdCount) { a.xcd = 1; nwg -= nwg % kXcdCount; }
        e = launch_sddmm_ks<4, true>(ks, a, nwg, stream);
    } else {
        e = plan->waves == 4 ? launch_sddmm_ks<4, false>(ks, a, plan->nw_eff, stream) : launch_sddmm_ks<1, false>(ks, a, plan->nw_eff, stream);
    }
    HIP_TRY(e);
    timer.stop();
    // (the range guard's fallback: returns at once unless X is "wide")
    if (g_range_guard >= 2) {   // a few dirty rows: the patch behind the MFMA kernel; many: the CSR fallback (each returns at once otherwise)
        const PatchArgs pa{hdr, plan->rowptr, plan->col, plan->e2r, d_X, x16, pitch, d_ef, nullptr, nullptr, nullptr, nullptr, plan->N, plan->Nc, D, plan->row_off, 0, plan->E};
        HIP_TRY(launch_wide_patch(pa, stream));
    }
    HIP_TRY(hipGetLastError());
    return TCGNN_OK;
}


} // extern "C"
