// tcgnn_host.cpp - host side of libtcgnn_hip.so: status strings and the host sparse-graph
// translation (SGT).
//
// tcgnn_preprocess replaces the reference's `preprocess` (TCGNN_conv/TCGNN.cpp:172-226, helper
// inplace_deduplication :157-170).  Same outputs, different method: the reference handles the
// windows serially (its two `#pragma omp` lines are inert because setup.py passes no -fopenmp;
// see TCGNN_conv/debug.log), mallocs (and leaks) a buffer per window, thrust::sort()s it and
// builds a std::map whose lookup it pays once per edge.  Here windows are handed out to a pool of
// host threads; a window whose rows are already sorted (scipy canonical CSR, the only producer in
// the reference: dataset.py:94-104) is merged run by run instead of sorted, and every edge's rank
// comes from one monotone walk per row.  No allocation per window, nothing leaked.
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "tcgnn.h"
#include "tcgnn_internal.h"

namespace tcgnn {

static thread_local std::string g_last_error;

int fail(int status, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}

namespace {

struct WindowScratch {
    std::vector<uint32_t> a, b;
};

// Sorted unique column ids of rows [n0, n1) into s.a[0..uniq); returns uniq.
// *rows_sorted tells the caller whether every row was strictly non-decreasing (merge path taken).
inline uint32_t window_unique(const int32_t* col, const int32_t* rp, int64_t n0, int64_t n1,
                              WindowScratch& s, bool* rows_sorted) {
    const int64_t e0 = rp[n0], e1 = rp[n1];
    const size_t n = (size_t)(e1 - e0);
    if (n == 0) { *rows_sorted = true; return 0; }
    if (s.a.size() < n) { s.a.resize(n); s.b.resize(n); }
    uint32_t* a = s.a.data();
    std::memcpy(a, col + e0, n * sizeof(uint32_t));
    bool sorted = true;
    for (int64_t r = n0; r < n1 && sorted; ++r) {
        const uint32_t* p = a + (rp[r] - e0);
        const size_t len = (size_t)(rp[r + 1] - rp[r]);
        for (size_t i = 1; i < len; ++i)
            if (p[i - 1] > p[i]) { sorted = false; break; }
    }
    *rows_sorted = sorted;
    if (!sorted) {
        std::sort(a, a + n);
    } else {
        // bottom-up merge of the (up to 16) sorted row runs, ping-ponging between a and b
        size_t off[TCGNN_BLK_H * 4 + 2];
        int runs = 0;
        std::vector<size_t> big; // only for exotic blockSize_h
        size_t* bounds = off;
        const int64_t nrows = n1 - n0;
        if (nrows + 1 > (int64_t)(sizeof off / sizeof off[0])) { big.resize((size_t)nrows + 1); bounds = big.data(); }
        for (int64_t r = n0; r <= n1; ++r) bounds[runs++] = (size_t)(rp[r] - e0);
        runs -= 1; // number of runs
        uint32_t* src = a;
        uint32_t* dst = s.b.data();
        while (runs > 1) {
            int out = 0;
            for (int i = 0; i + 1 < runs; i += 2) {
                std::merge(src + bounds[i], src + bounds[i + 1], src + bounds[i + 1], src + bounds[i + 2], dst + bounds[i]);
                bounds[out++] = bounds[i];
            }
            if (runs & 1) {
                std::memcpy(dst + bounds[runs - 1], src + bounds[runs - 1], (bounds[runs] - bounds[runs - 1]) * sizeof(uint32_t));
                bounds[out++] = bounds[runs - 1];
            }
            bounds[out] = n;
            runs = out;
            std::swap(src, dst);
        }
        if (src != a) std::memcpy(a, src, n * sizeof(uint32_t));
    }
    return (uint32_t)(std::unique(a, a + n) - a);
}

} // namespace
} // namespace tcgnn

using namespace tcgnn;

extern "C" {

int tcgnn_abi_version(void) { return TCGNN_ABI_VERSION; }

const char* tcgnn_status_string(int status) {
    switch (status) {
        case TCGNN_OK: return "ok";
        case TCGNN_ERR_INVALID_ARG: return "invalid argument";
        case TCGNN_ERR_HIP: return "HIP runtime error";
        case TCGNN_ERR_OOM: return "out of memory";
        case TCGNN_ERR_BAD_GRAPH: return "graph metadata inconsistent";
        case TCGNN_ERR_WORKSPACE: return "workspace missing or too small";
        case TCGNN_ERR_UNSUPPORTED: return "not supported by the fused entry point";
        default: return "unknown status";
    }
}

const char* tcgnn_last_error(void) { return g_last_error.c_str(); }

int tcgnn_preprocess(const int32_t* edgeList, const int32_t* nodePointer, int32_t num_nodes,
                     int32_t blockSize_h, int32_t blockSize_w, int32_t* blockPartition,
                     int64_t bp_len, int32_t* edgeToColumn, int32_t* edgeToRow,
                     int64_t* tc_blocks, int32_t num_threads) {
    if (!nodePointer || num_nodes < 0 || blockSize_h <= 0 || blockSize_w <= 0 || bp_len < 0)
        return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_preprocess: bad sizes (N=%d, bh=%d, bw=%d)", num_nodes, blockSize_h, blockSize_w);
    if (nodePointer[0] < 0) return fail(TCGNN_ERR_BAD_GRAPH, "tcgnn_preprocess: nodePointer[0] = %d is negative", nodePointer[0]);
    for (int32_t r = 0; r < num_nodes; ++r)
        if (nodePointer[r + 1] < nodePointer[r])
            return fail(TCGNN_ERR_BAD_GRAPH, "tcgnn_preprocess: nodePointer decreases at row %d (%d -> %d)", r, nodePointer[r], nodePointer[r + 1]);
    const int64_t E = nodePointer[num_nodes];
    if (E < 0 || (E > 0 && (!edgeList || !edgeToColumn || !edgeToRow)) || (bp_len > 0 && !blockPartition))
        return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_preprocess: null array");
    // windows the reference's loop visits: iter = 0, bh, 2bh, ... <= N   (TCGNN.cpp:200)
    const int64_t visited = (int64_t)num_nodes / blockSize_h + 1;
    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = num_threads > 0 ? num_threads : (hw ? (int)hw : 1);
    if ((int64_t)nthreads > visited) nthreads = (int)visited;
    if (E < (1 << 15)) nthreads = 1; // thread start-up would dominate

    std::atomic<int64_t> next{0};
    std::atomic<int64_t> total{0};
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(64, visited / (nthreads * 8) + 1));

    auto worker = [&]() {
        WindowScratch s;
        int64_t local = 0;
        for (;;) {
            const int64_t w0 = next.fetch_add(chunk);
            if (w0 >= visited) break;
            const int64_t w1 = std::min(visited, w0 + chunk);
            for (int64_t w = w0; w < w1; ++w) {
                const int64_t n0 = w * blockSize_h;
                const int64_t n1 = std::min<int64_t>(n0 + blockSize_h, num_nodes);
                for (int64_t r = n0; r < n1; ++r)                       // TCGNN.cpp:194-197
                    for (int64_t e = nodePointer[r]; e < nodePointer[r + 1]; ++e) edgeToRow[e] = (int32_t)r;
                bool rows_sorted = true;
                uint32_t uniq = 0;
                if (n0 < n1) uniq = window_unique(edgeList, nodePointer, n0, n1, s, &rows_sorted);
                const uint32_t* U = s.a.data();
                if (uniq) {
                    if (rows_sorted) {
                        for (int64_t r = n0; r < n1; ++r) {
                            uint32_t p = 0;
                            for (int64_t e = nodePointer[r]; e < nodePointer[r + 1]; ++e) {
                                const uint32_t key = (uint32_t)edgeList[e];
                                while (U[p] < key) ++p;
                                edgeToColumn[e] = (int32_t)p;
                            }
                        }
                    } else {
                        for (int64_t e = nodePointer[n0]; e < nodePointer[n1]; ++e)
                            edgeToColumn[e] = (int32_t)(std::lower_bound(U, U + uniq, (uint32_t)edgeList[e]) - U);
                    }
                }
                // an edgeless window reports one unique id in the reference (TCGNN.cpp:160 reads
                // array[0] of a zero-byte buffer) -> blockPartition = 1
                const uint32_t eff = uniq ? uniq : 1;
                const int32_t bp = (int32_t)((eff + (uint32_t)blockSize_w - 1) / (uint32_t)blockSize_w);
                if (w < bp_len) blockPartition[w] = bp;                 // never past the end
                local += bp;
            }
        }
        total.fetch_add(local);
    };

    if (nthreads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        pool.reserve((size_t)nthreads);
        for (int t = 0; t < nthreads; ++t) pool.emplace_back(worker);
        for (auto& t : pool) t.join();
    }
    if (tc_blocks) *tc_blocks = total.load();
    return TCGNN_OK;
}

// Tile statistics of the sparse-graph translation (what 3_cnt_TC_blk_SpMM.py:38-94 / 3_cnt_TC_blk_SDDMM.py count
// with Python sets): per window of tile_h rows take the sorted unique neighbour ids U;
//   condensed tiles = ceil(|U| / tile_w)                                  (3_cnt_TC_blk_SpMM.py:66)
//   sliding tiles   = greedy cover of U by intervals [u, u + tile_w)      (3_cnt_TC_blk_SpMM.py:72-81)
int tcgnn_tile_stats(const int32_t* edgeList, const int32_t* nodePointer, int32_t num_nodes, int32_t tile_h, int32_t tile_w,
                     tcgnn_tile_stats_t* out, int32_t num_threads) {
    if (!nodePointer || !out || num_nodes < 0 || tile_h <= 0 || tile_w <= 0)
        return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_tile_stats: bad argument (N=%d, tile %dx%d)", num_nodes, tile_h, tile_w);
    if (nodePointer[0] < 0) return fail(TCGNN_ERR_BAD_GRAPH, "tcgnn_tile_stats: nodePointer[0] = %d is negative", nodePointer[0]);
    for (int32_t r = 0; r < num_nodes; ++r)
        if (nodePointer[r + 1] < nodePointer[r])
            return fail(TCGNN_ERR_BAD_GRAPH, "tcgnn_tile_stats: nodePointer decreases at row %d", r);
    const int64_t E = nodePointer[num_nodes];
    if (E > 0 && !edgeList) return fail(TCGNN_ERR_INVALID_ARG, "tcgnn_tile_stats: null edge list");
    const int64_t windows = ((int64_t)num_nodes + tile_h - 1) / tile_h;
    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = num_threads > 0 ? num_threads : (hw ? (int)hw : 1);
    if ((int64_t)nthreads > windows) nthreads = (int)std::max<int64_t>(1, windows);
    if (E < (1 << 15)) nthreads = 1;
    std::atomic<int64_t> next{0};
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(64, windows / (nthreads * 8) + 1));
    struct Acc { int64_t sliding = 0, condensed = 0, uniq = 0, nonempty = 0, max_sliding = 0, max_condensed = 0; };
    std::vector<Acc> accs((size_t)nthreads);
    auto worker = [&](int tid) {
        WindowScratch s;
        Acc a;
        for (;;) {
            const int64_t w0 = next.fetch_add(chunk);
            if (w0 >= windows) break;
            const int64_t w1 = std::min(windows, w0 + chunk);
            for (int64_t w = w0; w < w1; ++w) {
                const int64_t n0 = w * tile_h, n1 = std::min<int64_t>(n0 + tile_h, num_nodes);
                bool rows_sorted;
                const uint32_t uniq = window_unique(edgeList, nodePointer, n0, n1, s, &rows_sorted);
                if (!uniq) continue;
                const uint32_t* U = s.a.data();
                int64_t slide = 0;
                for (uint32_t i = 0; i < uniq; ++slide) {
                    const uint64_t end = (uint64_t)U[i] + (uint64_t)tile_w;
                    while (i < uniq && U[i] < end) ++i;
                }
                const int64_t cond = ((int64_t)uniq + tile_w - 1) / tile_w;
                a.sliding += slide; a.condensed += cond; a.uniq += uniq; a.nonempty += 1;
                a.max_sliding = std::max(a.max_sliding, slide); a.max_condensed = std::max(a.max_condensed, cond);
            }
        }
        accs[(size_t)tid] = a;
    };
    if (nthreads <= 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t) pool.emplace_back(worker, t);
        for (auto& t : pool) t.join();
    }
    Acc a;
    for (const Acc& b : accs) {
        a.sliding += b.sliding; a.condensed += b.condensed; a.uniq += b.uniq; a.nonempty += b.nonempty;
        a.max_sliding = std::max(a.max_sliding, b.max_sliding); a.max_condensed = std::max(a.max_condensed, b.max_condensed);
    }
    out->windows = windows; out->nonempty_windows = a.nonempty; out->edges = E; out->unique_columns = a.uniq;
    out->sliding_tiles = a.sliding; out->condensed_tiles = a.condensed;
    out->max_sliding_per_window = a.max_sliding; out->max_condensed_per_window = a.max_condensed;
    return TCGNN_OK;
}

} // extern "C"
