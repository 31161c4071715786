/*
 * tcgnn.h - C ABI of the MI355X-native TC-GNN aggregation engine (libtcgnn_hip.so).
 *
 * This is the drop-in boundary of the hot path.  Every entry point replaces one function of the
 * reference's `TCGNN` pybind11 module (file:line relative to the reference checkout) and takes
 * plain pointers and sizes only - no torch types.  A Python (ctypes), C++ or any FFI caller can
 * bind it; the binding the reference's maintainers would add is shown in INTEGRATION.md.
 *
 *   reference (TCGNN_conv/TCGNN.cpp)                     this ABI
 *   ---------------------------------------------------  --------------------------------------
 *   preprocess            :172-226 (+ :157-170)          tcgnn_preprocess
 *   preprocess_gpu        :229-256 (unfinished there)    tcgnn_preprocess_gpu
 *   forward  = spmm_forward       :63-86   -> TCGNN_kernel.cu:175-220, :336-454     tcgnn_spmm
 *   forward_AGNN = spmm_forward_AGNN :93-118 -> TCGNN_kernel.cu:227-279, :459-578   tcgnn_spmm_val
 *   forward_ef = sddmm_forward    :126-150 -> TCGNN_kernel.cu:286-327, :584-727     tcgnn_sddmm
 *   backward / backward_ef :270-271 (aliases of forward / forward_ef)               same two calls
 *
 * Conventions
 *   - Every function returns a tcgnn_status (0 = OK) and never aborts the process (the reference
 *     printf()s and exit(-1)s on a launch error, TCGNN_kernel.cu:211-217).  tcgnn_last_error()
 *     returns a thread-local human-readable message for the last non-OK status.
 *   - Device pointers are BORROWED for the duration of the call (the plan borrows the five legacy
 *     arrays for its lifetime, see tcgnn_plan_create); outputs are caller-allocated.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  All device
 *     work is enqueued asynchronously on it.  What synchronises the stream: tcgnn_plan_create (to size its
 *     outputs), tcgnn_preprocess_gpu_ws (ONCE, to read back 24 bytes; it allocates nothing), tcgnn_plan_prepare, and - ONLY for a feature width
 *     tcgnn_plan_prepare was not called for - the first tcgnn_spmm / tcgnn_spmm_fused / tcgnn_spmm_gemm
 *     call of that width on a plan whose time model picks the LDS-resident kernel (it builds the
 *     width's cell stream: allocations, a few count-and-place round trips).  Call tcgnn_plan_prepare
 *     for every width a model uses and no hot-path call synchronises, allocates or - inside a HIP
 *     graph capture - takes a different walk than it would outside one.
 *   - Operand range.  The MFMA kernels read a power-of-two-scaled fp16 image of X (one scale per matrix and call): every
 *     element within 2^28 of the largest is rounded bit-for-bit like the reference's TF32 operand (TCGNN_kernel.cu:438-444),
 *     smaller ones lose mantissa bits and, 2^39 below the maximum, are flushed - an absolute error of at most max|X| * 2^-39
 *     per element.  A matrix (or edge-value array) whose largest magnitude is LARGE and which holds nonzero elements more than
 *     2^28 below it is routed - by a test on the device, no read-back - to fp32 fallback kernels that keep fp32's exponent like
 *     the reference does (slow, correct for any magnitudes); everything else stays on the MFMA path.  "Large" is where the
 *     errors one result can collect - at most k = min(longest row of the graph, number of elements of the matrix that lose
 *     bits) of them for an SpMM, min(2 D, that number) for SDDMM and the fused AGNN pair - could leave the contract's
 *     1e-3 max(1, |ref|): max|X| >= 2^29 / k (binary SpMM), max|A| max|X| >= 2^28 / k (edge-valued), max|X|^2 >= 2^29 / k
 *     (SDDMM, fused AGNN).  A training epoch's activations (one stray 1e-5 among 1e4's) stay on the MFMA path.  Images the
 *     CALLER stages (tcgnn_spmm_staged) carry no range words and always take the MFMA path (the kernels
 *     do not look at the rest of such a header).
 *   - Index arrays are int32 (the reference API's dtype); all address arithmetic inside the
 *     kernels is 64-bit, so N*D may exceed 2^32 (the reference overflows there,
 *     TCGNN_kernel.cu:420).
 *   - Tile unit: blockPartition counts 16x8 TC blocks per 16-row window (config.h:4-5); the
 *     kernels consume them four at a time as 16x32 MFMA operand tiles.
 */
#ifndef TCGNN_H
#define TCGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TCGNN_ABI_VERSION 1
#define TCGNN_BLK_H 16 /* rows per row window          (config.h:4) */
#define TCGNN_BLK_W 8  /* condensed columns per TC block (config.h:5) */

typedef enum tcgnn_status {
    TCGNN_OK = 0,
    TCGNN_ERR_INVALID_ARG = 1, /* null pointer, negative size, unsupported tile shape ... */
    TCGNN_ERR_HIP = 2,         /* a HIP runtime call or kernel launch failed             */
    TCGNN_ERR_OOM = 3,         /* host or device allocation failed                       */
    TCGNN_ERR_BAD_GRAPH = 4,   /* metadata inconsistent with the CSR (out-of-range ids)  */
    TCGNN_ERR_WORKSPACE = 5,   /* workspace pointer null or smaller than required        */
    TCGNN_ERR_UNSUPPORTED = 6  /* the fused entry point does not cover this plan / width  */
} tcgnn_status;

/* Opaque device-resident translation of one graph: the condensed 16x32 tile stream the kernels
 * read (per tile: 32 source-row ids, a 16x32 adjacency bitmask, 16 edge offsets). */
typedef struct tcgnn_plan tcgnn_plan;

typedef struct tcgnn_plan_info {
    int32_t num_nodes;       /* N */
    int32_t num_windows;     /* ceil(N/16) as given by the caller's blockPartition length */
    int64_t num_edges;       /* E = nnz of the CSR */
    int64_t tc_blocks;       /* sum(blockPartition): 16x8 TC blocks (the reference's "TC_Blocks") */
    int64_t wide_blocks;     /* 16x32 operand tiles actually streamed by the kernels */
    int64_t plan_bytes;      /* device bytes owned by the plan */
    int32_t canonical;       /* 1 if every CSR row is strictly increasing (scipy canonical form) */
    int32_t waves_per_window;/* workgroup shape the launcher picked (1 or 4 wavefronts) */
    int32_t column_buckets;  /* > 0: the plan carries the bucket table of the range-blocked SpMM walk */
    int32_t lds_ranges;      /* > 0: the plan carries a cell stream of the LDS-resident column-range SpMM (column ranges of the
                              * finest stream built so far; streams are per pass width and built on first use) */
} tcgnn_plan_info;

int tcgnn_abi_version(void);
const char* tcgnn_status_string(int status);
const char* tcgnn_last_error(void);

/* ---- sparse-graph translation (SGT) ------------------------------------------------------- */

/* Host SGT.  Replaces TCGNN.preprocess (TCGNN.cpp:172-226): fills, in place,
 *   edgeToRow[e]      = row of CSR edge e,
 *   edgeToColumn[e]   = rank of edgeList[e] among the sorted unique column ids of e's row window,
 *   blockPartition[w] = ceil(#unique / blockSize_w), and 1 for a window without edges (what the
 *                       reference's zero-length read yields, TCGNN.cpp:160),
 * for w < bp_len only (the reference writes one slot past the end when N % blockSize_h == 0).
 * *tc_blocks receives the count the reference prints as "TC_Blocks" (including that phantom
 * window).  All pointers are HOST memory.  num_threads <= 0 means "all hardware threads".
 * Rows need not be sorted and may hold duplicate columns. */
int tcgnn_preprocess(const int32_t* edgeList, const int32_t* nodePointer, int32_t num_nodes,
                     int32_t blockSize_h, int32_t blockSize_w, int32_t* blockPartition,
                     int64_t bp_len, int32_t* edgeToColumn, int32_t* edgeToRow,
                     int64_t* tc_blocks, int32_t num_threads);

/* Device SGT.  Same outputs as tcgnn_preprocess, all pointers DEVICE memory.  Finishes what the
 * reference's preprocess_gpu / fill_window only sketch (TCGNN.cpp:229-256,
 * TCGNN_kernel.cu:42-80).
 *
 * tcgnn_preprocess_gpu_ws (r06) runs on CALLER scratch: d_workspace of at least
 * tcgnn_preprocess_gpu_workspace_bytes(num_nodes, num_edges, blockSize_h) bytes, 256-byte aligned (sort keys / positions / flags /
 * ranks of the num_edges edge slots plus rocPRIM's scratch: ~2.3 GB at Reddit size).  It allocates and frees NOTHING and synchronises
 * `stream` ONCE, to read back 24 bytes - nodePointer[0], nodePointer[num_nodes], the largest column id and *tc_blocks.  (A graph with
 * column ids beyond num_nodes - legal, the host path takes them too - is sorted a second time with every key bit: the only way to a
 * second synchronisation.)  Malformed row pointers are reported (TCGNN_ERR_BAD_GRAPH) after that read-back; until then every index
 * derived from them is clamped to the arrays.
 * tcgnn_preprocess_gpu keeps the reference's shape (no scratch argument): it allocates the workspace with hipMalloc, calls the _ws
 * form and frees it - what r01-r05 measured as 5 ms or 95 ms by the allocator's mood. */
int tcgnn_preprocess_gpu_workspace_bytes(int32_t num_nodes, int64_t num_edges, int32_t blockSize_h, size_t* bytes);
int tcgnn_preprocess_gpu_ws(const int32_t* d_edgeList, const int32_t* d_nodePointer,
                            int32_t num_nodes, int64_t num_edges, int32_t blockSize_h,
                            int32_t blockSize_w, int32_t* d_blockPartition, int64_t bp_len,
                            int32_t* d_edgeToColumn, int32_t* d_edgeToRow, void* d_workspace,
                            size_t workspace_bytes, int64_t* tc_blocks, void* stream);
int tcgnn_preprocess_gpu(const int32_t* d_edgeList, const int32_t* d_nodePointer,
                         int32_t num_nodes, int64_t num_edges, int32_t blockSize_h,
                         int32_t blockSize_w, int32_t* d_blockPartition, int64_t bp_len,
                         int32_t* d_edgeToColumn, int32_t* d_edgeToRow, int64_t* tc_blocks,
                         void* stream);

/* Tile statistics of the translation - what the reference's counting scripts report
 * (3_cnt_TC_blk_SpMM.py:38-94 with 16x8 tiles, 3_cnt_TC_blk_SDDMM.py with 16x16; logs/16x8_reduction.csv,
 * logs/reduce_blocks.csv): per window of tile_h rows, U = sorted unique neighbour ids;
 * condensed tiles = ceil(|U|/tile_w), sliding tiles = greedy cover of U by intervals of tile_w ids.
 * HOST pointers; rows need not be sorted.  num_threads <= 0 means "all hardware threads". */
typedef struct tcgnn_tile_stats_t {
    int64_t windows;                 /* ceil(N / tile_h) */
    int64_t nonempty_windows;
    int64_t edges;                   /* nnz of the CSR (duplicates included) */
    int64_t unique_columns;          /* sum over windows of |U| */
    int64_t sliding_tiles;           /* the scripts' "origin" column */
    int64_t condensed_tiles;         /* the scripts' "reduced" column */
    int64_t max_sliding_per_window;
    int64_t max_condensed_per_window;
} tcgnn_tile_stats_t;
int tcgnn_tile_stats(const int32_t* edgeList, const int32_t* nodePointer, int32_t num_nodes,
                     int32_t tile_h, int32_t tile_w, tcgnn_tile_stats_t* out, int32_t num_threads);

/* ---- plan: legacy metadata -> packed tile stream (device) --------------------------------- */

/* Builds the packed tile stream on the device from the five legacy arrays every reference entry
 * point receives (TCGNN.cpp:63-70).  The arrays are DEVICE pointers and stay borrowed by the
 * plan until tcgnn_plan_destroy (the fallback kernels for non-canonical CSRs read them).
 * Synchronises `stream` once. */
int tcgnn_plan_create(const int32_t* d_nodePointer, const int32_t* d_edgeList,
                      const int32_t* d_blockPartition, const int32_t* d_edgeToColumn,
                      const int32_t* d_edgeToRow, int32_t num_nodes, int64_t num_edges,
                      int32_t num_windows, void* stream, tcgnn_plan** plan_out);
/* Row-sharded variant for multi-GPU runs (no counterpart in the reference, which is single-GPU):
 * A holds `num_rows` rows (this rank's row windows; nodePointer has num_rows + 1 entries) whose
 * column ids index a feature matrix of `num_cols` rows (the all-gathered X); A's row r is X's row
 * row_offset + r (used by SDDMM).  X passed to the kernels is [num_cols, D], Y is [num_rows, D]. */
int tcgnn_plan_create_sharded(const int32_t* d_nodePointer, const int32_t* d_edgeList,
                              const int32_t* d_blockPartition, const int32_t* d_edgeToColumn,
                              const int32_t* d_edgeToRow, int32_t num_rows, int32_t num_cols,
                              int32_t row_offset, int64_t num_edges, int32_t num_windows,
                              void* stream, tcgnn_plan** plan_out);
/* Builds, now, whatever the hot path would otherwise build on its first call at feature width D (the cell stream(s) of the
 * LDS-resident kernel where the plan's time model picks it for that width; nothing for the gather walks).  Synchronises
 * `stream`.  Idempotent; widths are independent.  No counterpart in the reference (its kernels re-derive their tiles per launch). */
int tcgnn_plan_prepare(tcgnn_plan* plan, int32_t D, void* stream);
/* The same for the EDGE-VALUED SpMM (tcgnn_spmm_val = TCGNN.forward_AGNN, gnn_conv.py:132,143): builds the single-edge cell stream of the
 * LDS-resident edge-valued walk where the plan takes that walk at width D (whole 64-column chunks, alone or followed by a remainder of
 * 33 .. 48 columns - Reddit's 41 classes -, on graphs the time model sends to the LDS-resident kernel; nothing otherwise).  Call it BEFORE tcgnn_workspace_bytes: the answer then includes the per-call slot
 * values, and the first tcgnn_spmm_val of that width neither allocates nor synchronises nor takes the gather walk.  Synchronises
 * `stream`.  Idempotent. */
int tcgnn_plan_prepare_val(tcgnn_plan* plan, int32_t D, void* stream);
int tcgnn_plan_destroy(tcgnn_plan* plan);
int tcgnn_plan_get_info(const tcgnn_plan* plan, tcgnn_plan_info* info);

/* Tuning / test aid: which SpMM walk tcgnn_spmm and tcgnn_spmm_val use.  0 = automatic (default: per plan
 * and feature width, the LDS-resident kernel or the gather walks by their time models - DESIGN.md "Which walk
 * runs"; among the gather walks range-blocked when the fp16 image of X exceeds the L2 and the windows are long;
 * TCGNN_LDS_AUTO=0 in the environment keeps the automatic mode off the LDS-resident kernel), 1 = always the
 * plain per-window kernel, 2 = range-blocked whenever the plan has a bucket table, 3 = the
 * LDS-resident column-range kernel (binary SpMM only; builds its cell stream on first use if the
 * plan was created without one), 4 = the single-launch fp32-MFMA kernel small graphs take automatically
 * (no staging pass; binary SpMM only), 5 = the slice-synchronised range walk (r06: graphs whose communities exceed an XCD's L2 -
 * taken automatically there - from 64 columns, sparse windows, by operator and hot span: DESIGN.md 4.4; forced, it runs wherever the plan built its tables and falls back to the
 * gather walks elsewhere; SDDMM and the fused AGNN pair follow the same switch).  Process-wide; the environment variable TCGNN_SPMM_MODE sets the initial value. */
int tcgnn_set_spmm_mode(int32_t mode);
/* The same switch for ONE plan: mode 0 .. 5 as above, -1 = follow the process-wide value (the default).  Two plans of one process may
 * walk differently. */
int tcgnn_plan_set_spmm_mode(tcgnn_plan* plan, int32_t mode);

/* The range guard (see "Operand range" above), process-wide level (environment TCGNN_RANGE_GUARD sets the initial one):
 *   0  off - every call stays on the MFMA path;
 *   1  the aggregation operators are guarded: tcgnn_spmm / _fused / _gemm and tcgnn_spmm_val, whose error bound is LINEAR in
 *      max|X| - k max 2^-39 - and which ordinary training tensors never reach;
 *   2  (default since r04) also tcgnn_sddmm and the fused AGNN pair, whose bound is QUADRATIC - min(2 D, lost elements)
 *      max|X|^2 2^-39 - for what training produces: with the reference's unscaled weights an AGNN epoch's activations (max ~3e4,
 *      ONE element 2^28 below) cross that bound now and then.  Such a matrix has "dirty" rows - rows of X holding elements that lose
 *      bits in the image; the conversion pass marks them in a bitmap behind the image - and the MFMA kernels run as usual while one
 *      more launch scans the edges and recomputes, in fp32 with the reference's operand rounding, exactly the edges that touch a
 *      dirty row (scores, their share of the aggregate and of d_w): ~0.2 ms of scan plus a wavefront's work per such edge when it
 *      happens, a launch that returns at once when it does not.  Any number of dirty rows (r04 patched at most 48 and left a matrix
 *      with more - hub rows of a power-law graph under unscaled weights - to the bound above: closed in r05).  On a structurally
 *      symmetric graph (checked once, at plan creation) up to 256 dirty rows are patched without the scan: a dirty row's own edges
 *      and their mirrors, ~10 us - what a training epoch usually meets;
 *   3  strict: a wide matrix is computed in plain fp32, CSR order, by SDDMM and the fused AGNN pair as well (correct for any
 *      magnitudes, ~50x slower than the MFMA path on a Reddit-sized graph; level 2 gives the same guarantee at the cost of the dirty edges).
 * tcgnn_range_mode reports which way the LAST staged call on this workspace went: *wide_x = 1 if its feature matrix took the fp32
 * fallback as a binary SpMM / SDDMM / fused AGNN operand, 2 if it stayed on the MFMA path with its dirty rows patched (SDDMM /
 * fused AGNN), *wide_val (optional) = 1 if it took the fallback as an edge-valued SpMM.  Reads 40 bytes of the workspace header
 * back: synchronises `stream` (a test / diagnosis aid, like tcgnn_plan_last_kernel - the hot path never reads anything back). */
int tcgnn_set_range_guard(int32_t level);
/* The guard level of ONE plan: 0 .. 3 as above, -1 = follow the process-wide level (the default). */
int tcgnn_plan_set_range_guard(tcgnn_plan* plan, int32_t level);
int tcgnn_range_mode(const void* d_workspace, void* stream, int32_t* wide_x, int32_t* wide_val);

/* Measurement aid: reserve HIP event pairs for up to `max_calls` kernel calls (0 = off).  While
 * enabled, tcgnn_spmm / tcgnn_spmm_val / tcgnn_sddmm bracket their main kernel (not the fp16
 * staging pass) with events recorded on the caller's stream, without synchronising.
 * tcgnn_plan_read_timing waits for the recorded calls, returns their durations in milliseconds
 * (call order) and rearms the pairs. */
int tcgnn_plan_set_timing(tcgnn_plan* plan, int32_t max_calls);
/* Measurement aid: name of the main kernel the most recent hot-path call on this plan launched ("spmm_lds_kernel",
 * "spmm_blocked_kernel", "spmm_kernel", "spmm_small_kernel", "sddmm_kernel", "agnn_kernel", ...; "" before the first call) -
 * so a benchmark line names the kernel that ran instead of re-deriving the launcher's choice.  Static storage. */
const char* tcgnn_plan_last_kernel(const tcgnn_plan* plan);
int tcgnn_plan_read_timing(tcgnn_plan* plan, float* ms_out, int32_t capacity, int32_t* count);

/* Scratch bytes the three kernels need for feature width D (fp16 staging copy of X with one
 * zero sentinel row, plus the scale words).  The caller owns the scratch; it must be 256-byte
 * aligned device memory and may be reused across calls on the same stream. */
size_t tcgnn_workspace_bytes(const tcgnn_plan* plan, int32_t D);

/* ---- the hot path ------------------------------------------------------------------------- */

/* Y[N,D] = A_bin * X.  Replaces TCGNN.forward / TCGNN.backward (TCGNN.cpp:63-86).
 * X, Y: fp32 row-major device arrays [N, D]; Y is fully overwritten (rows without edges = 0).
 * Operands are rounded to a 10-bit mantissa (fp16, per-call power-of-two scaled) exactly as the
 * reference rounds to TF32; products accumulate in fp32 on MFMA. Any D >= 1. */
int tcgnn_spmm(const tcgnn_plan* plan, const float* d_X, float* d_Y, int32_t D,
               void* d_workspace, size_t workspace_bytes, void* stream);

/* f3 - the dense update's element-wise steps fused around the aggregation (the reference runs them as separate torch
 * kernels: F.relu after the layer, main_tcgnn.py:100-139; its backward mask before gnn_conv.py:80):
 *   flags & TCGNN_FUSE_RELU : Y = max(A_bin * X, 0)                      (ReLU epilogue in the kernels' stores)
 *   d_gate != NULL          : X'[r,c] = d_gate[r,c] > 0 ? X[r,c] : 0     (applied while X is staged to fp16: with
 *                             d_gate = the forward output this is the ReLU backward mask on dY).  [N, D] like X.
 * Results are bit-identical to max(tcgnn_spmm(X), 0) and tcgnn_spmm(X * (gate > 0)). */
#define TCGNN_FUSE_RELU 1
int tcgnn_spmm_fused(const tcgnn_plan* plan, const float* d_X, const float* d_gate, float* d_Y, int32_t D,
                     int32_t flags, void* d_workspace, size_t workspace_bytes, void* stream);

/* f3, the dense update itself: Y[N, D_out] = (A_bin * X) * W in one launch - the GIN order of the reference
 * (gnn_conv.py:92-97: X' = TCGNN.forward(X, ...)[0]; X' = torch.mm(X', weights)), which the harness also uses for a GCN layer that
 * narrows (A (H W) = (A H) W).  X [N, D_in] and W [D_in, D_out] fp32 row-major, D_in, D_out <= 128.  The aggregated rows never
 * leave the chip: each window's 16 x D_in fp32 sums go through LDS into the fp32 matrix pipe (v_mfma_f32_16x16x4_f32 - the exact
 * fp32 products and fp32 accumulation of the torch.mm it replaces) against W, which is read through the caches.  On the
 * LDS-resident kernel a 64-column input is two 32-column passes whose products are added into a zeroed Y (two addends: the
 * result does not depend on their order).  flags: TCGNN_FUSE_RELU (not when the product is accumulated over passes: the call
 * then returns TCGNN_ERR_UNSUPPORTED, as it does for wider matrices, and the caller composes tcgnn_spmm with its own GEMM). */
int tcgnn_spmm_gemm(const tcgnn_plan* plan, const float* d_X, const float* d_W, float* d_Y, int32_t D_in, int32_t D_out, int32_t flags,
                    void* d_workspace, size_t workspace_bytes, void* stream);

/* Pre-staged operand (no counterpart in the reference, which is single-GPU).  The kernels read a scaled fp16 image of X that
 * tcgnn_spmm builds per call inside the workspace.  A caller may build it itself - in a row-sharded run every rank converts
 * only ITS rows and the fp16 image, not fp32 X, crosses the fabric (half the bytes of the all-gather, no staging of the
 * gathered matrix on every rank):
 *   image  = 256-byte header (word 0: bit pattern of max|X| over the WHOLE matrix, e.g. after an all-reduce(MAX) of the
 *            per-rank words) followed by (plan's num_cols + 1) rows of tcgnn_x16_pitch(D) halves, the last row all zero;
 *   tcgnn_stage_absmax : atomicMax of the |X| bit patterns of n elements into *d_word (zero it first);
 *   tcgnn_stage_rows   : `rows` rows of X -> rows + 1 image rows at d_dst (the extra row is zero), scaled by *d_absmax_word
 *                        exactly as tcgnn_spmm would;
 *   tcgnn_spmm_staged  : Y = A_bin * X from such an image (gather walks; results identical to tcgnn_spmm on the same walk).
 *                        Only word 0 of the header is read (a staged image is never "wide": bytes 4 .. 255 are ignored and may hold
 *                        anything).  The image is READ-ONLY to the call: one image may feed several streams or ranks at once. */
int tcgnn_x16_pitch(int32_t D);
/* r06 - the same for the LDS-RESIDENT kernel, so that a row-sharded call keeps the fast walk.  That kernel reads a PLANAR image:
 *   256-byte header (word 0 as above; words 1 .. 63 ZERO) followed by ceil(D / 16) planes of (num_cols + 1) records of 16 halves
 *   (32 bytes): plane p, record r = columns 16 p .. 16 p + 15 of row r; record num_cols of every plane and all padding zero.
 *   tcgnn_spmm_staged_layout : 1 if tcgnn_spmm would run the LDS-resident kernel on this plan at this width and can take a planar
 *                              staged image (builds the width's cell streams if missing: may synchronise `stream` once), else 0 -
 *                              stage row-major;
 *   tcgnn_stage_rows_planar  : `rows` rows of X -> their records in every plane: d_dst = the slice's first record in plane 0,
 *                              plane p lies plane_rows (= num_cols + 1) records further on per plane.  Writes nothing else;
 *   tcgnn_spmm_staged_planar : Y = A_bin * X from such an image on the LDS-resident kernel (results identical to tcgnn_spmm in
 *                              mode 3 on the same matrix); TCGNN_ERR_UNSUPPORTED where tcgnn_spmm_staged_layout says 0.
 * A rank's slice of a plane is contiguous, so the exchange is one all-gather per plane. */
int tcgnn_spmm_staged_layout(const tcgnn_plan* plan, int32_t D, void* stream);
int tcgnn_stage_rows_planar(const float* d_X, int32_t rows, int32_t D, const uint32_t* d_absmax_word, void* d_dst, int64_t plane_rows, void* stream);
int tcgnn_spmm_staged_planar(const tcgnn_plan* plan, const void* d_image, float* d_Y, int32_t D, void* stream);
int tcgnn_stage_absmax(const float* d_X, int64_t n, uint32_t* d_word, void* stream);
int tcgnn_stage_rows(const float* d_X, int32_t rows, int32_t D, const uint32_t* d_absmax_word, void* d_dst, void* stream);
int tcgnn_spmm_staged(const tcgnn_plan* plan, const void* d_image, float* d_Y, int32_t D, void* stream);

/* Y[N,D] = A_val * X with A_val[r,c] = d_edge_val[e] for CSR edge e = (r,c).
 * Replaces TCGNN.forward_AGNN (TCGNN.cpp:93-118); d_edge_val is row 0 of edgeAttention[H,E]. */
int tcgnn_spmm_val(const tcgnn_plan* plan, const float* d_X, const float* d_edge_val, float* d_Y,
                   int32_t D, void* d_workspace, size_t workspace_bytes, void* stream);

/* ef[e] = <X[row(e),:], X[col(e),:]> for every CSR edge.  Replaces TCGNN.forward_ef /
 * TCGNN.backward_ef (TCGNN.cpp:126-150).  d_ef: fp32 [E], fully overwritten. */
int tcgnn_sddmm(const tcgnn_plan* plan, const float* d_X, float* d_ef, int32_t D,
                void* d_workspace, size_t workspace_bytes, void* stream);

/* ---- fused AGNN layer products (one gather of the neighbour rows feeds both) ----------------
 *
 * The reference's AGNN layer (gnn_conv.py:115-158) calls forward_ef and forward_AGNN back to back on
 * the same matrix, and again in backward; each call gathers every neighbour row once.  These two
 * entry points compute the same results with ONE pass over the tile stream.
 *
 * forward:   ef[e] = <X[row e], X[col e]>                         (= tcgnn_sddmm)
 *            Y     = A_att * X,  att[e] = fl32(w * ef[e])          (= tcgnn_spmm_val on w * ef)
 *            d_ef_absmax: 1 + N device words the caller allocates - word 0 = bit pattern of max |ef|, words 1 .. N = the power-of-two
 *            exponent each row's edge weights were scaled by (r05: one fp16 scale per ROW of A, so that small rows of a matrix whose
 *            weights spread over 2^30 and more keep their mantissas) - consumed by the backward call
 * backward:  G     = A_att * dY, att[e] = fl32(w * ef[e]) with the saved ef   (gnn_conv.py:143)
 *            *d_dw = sum_e <dY[row e], dY[col e]> * (float)col(e)             (gnn_conv.py:150-153:
 *                    the reference's d_attention_w, mm(backward_ef(dY)[None,:], column_index[:,None].float()))
 * d_w is the attention weight as a DEVICE scalar (no host read-back).  Rounding is the same as in
 * the separate calls (10-bit mantissa operands, fp32 accumulate); the power-of-two scale of row r's att comes
 * from a bound (|w| D max|x_r| max|X|) instead of a pass over E, so results equal the separate calls bit for bit unless an
 * edge weight is more than 2^-20 below its row's bound (a row whose neighbours' magnitudes spread that far).  d_dw is a fixed-order reduction (deterministic).
 * Supported for canonical plans (sorted, duplicate-free rows), D <= 128, E >= 8:
 * tcgnn_agnn_supported() tells; otherwise the calls return TCGNN_ERR_UNSUPPORTED and the caller
 * uses the three separate entry points. */
int tcgnn_agnn_supported(const tcgnn_plan* plan, int32_t D);
/* (r06, ADVICE r05: these two replace r04 / r05's tcgnn_agnn_forward / tcgnn_agnn_backward, whose d_ef_absmax grew from 1 word to 1 + N
 *  words in r05 under an unchanged signature - a caller built against the older header would have been written 4 N bytes past its word.
 *  The names are gone, so such a caller fails to resolve them; the new ones take the element count of d_ef_absmax and return
 *  TCGNN_ERR_INVALID_ARG when it is below 1 + N.) */
int tcgnn_agnn_pair_forward(const tcgnn_plan* plan, const float* d_X, const float* d_w, float* d_ef,
                            uint32_t* d_ef_absmax, int64_t ef_absmax_words, float* d_Y, int32_t D, void* d_workspace,
                            size_t workspace_bytes, void* stream);
int tcgnn_agnn_pair_backward(const tcgnn_plan* plan, const float* d_dY, const float* d_w, const float* d_ef,
                             const uint32_t* d_ef_absmax, int64_t ef_absmax_words, float* d_G, float* d_dw, int32_t D,
                             void* d_workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TCGNN_H */
