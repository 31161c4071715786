/* cpu_baseline.c - the CPU side of bench.py's `cpu_baseline` leg: Y = A X as a row-parallel CSR gather-add, the aggregation
 * step of the DGL CPU GCN path the north star names as the baseline (dgl_baseline/gcn.py:26-31 -> GraphConv copy_u / sum).
 * TEST / MEASUREMENT INFRASTRUCTURE, like everything under oracle/: never imported by the product.
 *
 * Unlike the oracle proper (tcgnn_oracle.c: portable x86-64-v2 build, bit-reproducible, contraction off) this file is built
 * ON THE MACHINE THAT RUNS IT with -O3 -march=native (oracle/cpu_baseline.py), because a baseline compiled for SSE and
 * first-touched by one thread is a strawman (r1 VERDICT):
 *   - rows are cut into one contiguous block per thread with EQUAL EDGE COUNTS (static, no scheduler traffic);
 *   - X and Y are first-touched in parallel by the owners of those row blocks (pages spread over every NUMA node);
 *   - the inner loop is width-specialised (D = 16 / 32 / 64 / 128: accumulators stay in vector registers) and prefetches
 *     the feature rows of the edges a few iterations ahead (the gather is latency-bound otherwise).
 */
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PF_DIST 12

/* bounds[t] .. bounds[t+1]: rows of thread t, equal nnz */
static void partition(const int32_t* rp, int32_t n, int threads, int32_t* bounds) {
    const int64_t total = rp[n];
    bounds[0] = 0;
    int32_t r = 0;
    for (int t = 1; t < threads; t++) {
        const int64_t target = total * t / threads;
        while (r < n && rp[r] < target) r++;
        bounds[t] = r;
    }
    bounds[threads] = n;
}

#define ROW_KERNEL(NAME, DD)                                                                                   \
    static void NAME(const int32_t* rp, const int32_t* col, const float* X, float* Y, int32_t r0, int32_t r1) { \
        for (int32_t r = r0; r < r1; r++) {                                                                    \
            float acc[DD];                                                                                     \
            for (int d = 0; d < DD; d++) acc[d] = 0.0f;                                                        \
            const int64_t e0 = rp[r], e1 = rp[r + 1];                                                          \
            for (int64_t e = e0; e < e1; e++) {                                                                \
                if (e + PF_DIST < e1) {                                                                        \
                    const char* p = (const char*)(X + (size_t)col[e + PF_DIST] * DD);                          \
                    for (int b = 0; b < DD * 4; b += 64) __builtin_prefetch(p + b, 0, 0);                      \
                }                                                                                              \
                const float* x = X + (size_t)col[e] * DD;                                                      \
                _Pragma("omp simd") for (int d = 0; d < DD; d++) acc[d] += x[d];                               \
            }                                                                                                  \
            float* y = Y + (size_t)r * DD;                                                                     \
            for (int d = 0; d < DD; d++) y[d] = acc[d];                                                        \
        }                                                                                                      \
    }
ROW_KERNEL(rows16, 16)
ROW_KERNEL(rows32, 32)
ROW_KERNEL(rows64, 64)
ROW_KERNEL(rows128, 128)

static void rows_any(const int32_t* rp, const int32_t* col, const float* X, float* Y, int32_t D, int32_t r0, int32_t r1) {
    for (int32_t r = r0; r < r1; r++) {
        float* y = Y + (size_t)r * D;
        for (int d = 0; d < D; d++) y[d] = 0.0f;
        for (int64_t e = rp[r]; e < rp[r + 1]; e++) {
            const float* x = X + (size_t)col[e] * D;
#pragma omp simd
            for (int d = 0; d < D; d++) y[d] += x[d];
        }
    }
}

/* Y = A X.  first_touch != 0: X is (re)initialised from Xsrc and Y zeroed INSIDE the parallel region by each block's owner before
 * the product (call once with freshly allocated, untouched X / Y to place their pages; later calls pass 0). */
int cpu_csr_spmm(const int32_t* rp, const int32_t* col, int32_t n, int32_t D, float* X, const float* Xsrc, float* Y, int32_t threads,
                 int32_t first_touch) {
    if (threads <= 0) threads = omp_get_max_threads();
    int32_t* bounds = (int32_t*)malloc(sizeof(int32_t) * (size_t)(threads + 1));
    if (!bounds) return 1;
    partition(rp, n, threads, bounds);
#pragma omp parallel num_threads(threads)
    {
        const int t = omp_get_thread_num();
        const int32_t r0 = bounds[t], r1 = bounds[t + 1];
        if (first_touch) {
            memcpy(X + (size_t)r0 * D, Xsrc + (size_t)r0 * D, sizeof(float) * (size_t)(r1 - r0) * D);
            memset(Y + (size_t)r0 * D, 0, sizeof(float) * (size_t)(r1 - r0) * D);
#pragma omp barrier
        }
        switch (D) {
            case 16: rows16(rp, col, X, Y, r0, r1); break;
            case 32: rows32(rp, col, X, Y, r0, r1); break;
            case 64: rows64(rp, col, X, Y, r0, r1); break;
            case 128: rows128(rp, col, X, Y, r0, r1); break;
            default: rows_any(rp, col, X, Y, D, r0, r1);
        }
    }
    free(bounds);
    return 0;
}

int cpu_max_threads(void) { return omp_get_max_threads(); }
