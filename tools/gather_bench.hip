// gather_bench.hip - what can the memory system deliver for the SpMM access pattern?
// Each wavefront streams "tiles": 32 row ids (coalesced 128-B read) -> 32 random rows of ROWB bytes
// gathered straight into LDS with global_load_lds_dwordx4, DEPTH tiles kept in flight with a
// counted s_waitcnt.  No MFMA, no LDS reads: this is the ceiling the SpMM kernel's gather can
// approach.  Sweeps the table size (L2-resident / MALL-resident / HBM) and DEPTH.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o tools/bin/gather_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
// A load hipcc does not know about: with LDS-DMA in flight it would otherwise wait vmcnt(0) at the
// first use of ANY ordinary load (cdna_hip_programming.md 5, trap (b)); we count the queue by hand.
__device__ __forceinline__ int hidden_load(const int* p) { int v; asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void settle(int& v) { asm volatile("" : "+v"(v)); }

// ROWB = bytes per gathered row (32..256), INSTR = DMA instructions per tile = 32*ROWB/1024
template <int ROWB, int DEPTH, int WAVES, bool HALF = false>
__global__ __launch_bounds__(WAVES * 64) void gather_kernel(const int* __restrict__ ids, const char* __restrict__ table, long tiles_per_wave, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int INSTR = 32 * ROWB / 1024;
    constexpr int TILE = 32 * ROWB;
    constexpr int LPR = ROWB / 16; // lanes per row
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* ring = smem + wave * DEPTH * TILE;
    const long gw = (long)blockIdx.x * WAVES + wave;
    const int* myids = ids + gw * tiles_per_wave * 32;
    int cid[DEPTH][INSTR];
    auto load_ids = [&](long t, int slot) {
#pragma unroll
        for (int k = 0; k < INSTR; ++k) cid[slot][k] = hidden_load(myids + t * 32 + (HALF ? 16 * (k >> 1) + (lane & 15) : (k * 64 + lane) / LPR));
    };
    auto issue = [&](int slot) {
#pragma unroll
        for (int k = 0; k < INSTR; ++k) {
            // HALF (ROWB = 128; r03): the fused AGNN kernel's mapping - an instruction takes 64 bytes of each of SIXTEEN rows (lane (g, i):
            // row i of half k / 2, chunk 4 (k % 2) + g), so every 128-byte line is asked for by two instructions
            const char* src = table + (long)cid[slot][k] * ROWB + (HALF ? (4 * (k & 1) + (lane >> 4)) * 16 : ((k * 64 + lane) % LPR) * 16);
            __builtin_amdgcn_global_load_lds((GLB_AS const void*)src, (LDS_AS void*)(ring + slot * TILE + k * 1024), 16, 0, 0);
        }
    };
    // Pipeline: per ring slot d the VMEM queue holds [DMA tile, id load for the tile after next] pairs,
    // oldest first.  vmcnt retires in order, so ids must be requested a full round before they are
    // used or waiting for them would drain every younger DMA.
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_ids(d, d);
    wait_vm<0>();
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { for (int k = 0; k < INSTR; ++k) settle(cid[d][k]); issue(d); load_ids(DEPTH + d, d); }
    long t = DEPTH;
    for (; t + 2 * DEPTH <= tiles_per_wave; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            wait_vm<(2 * DEPTH - 2) * INSTR>();   // oldest two groups retired: this slot's previous tile + its next ids
#pragma unroll
            for (int k = 0; k < INSTR; ++k) settle(cid[d][k]);
            issue(d);                              // ids were requested one round ago
            load_ids(t + DEPTH + d, d);
        }
    }
    wait_vm<0>();
    if (sink && ring[lane] == 123 && t == -1) sink[0] = 1; // keep LDS alive
}

template <int ROWB, int DEPTH, int WAVES, bool HALF = false>
double run(const int* d_ids, const char* d_table, long total_tiles, int reps, int* sink) {
    const long waves_total = 256L * 8 * 4 / 1; // plenty of waves
    long blocks = waves_total / WAVES;
    long tpw = total_tiles / (blocks * WAVES);
    if (tpw < 3 * DEPTH) tpw = 3 * DEPTH;
    const size_t lds = (size_t)WAVES * DEPTH * 32 * ROWB;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather_kernel<ROWB, DEPTH, WAVES, HALF>), dim3(blocks), dim3(WAVES * 64), lds, 0, d_ids, d_table, tpw, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((gather_kernel<ROWB, DEPTH, WAVES, HALF>), dim3(blocks), dim3(WAVES * 64), lds, 0, d_ids, d_table, tpw, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)blocks * WAVES * tpw * 32 * ROWB * reps;
    return bytes / (ms * 1e-3) / 1e12;
}

int main() {
    const long total_tiles = 3500000; // ~ the Reddit-shaped graph
    std::vector<int> ids((size_t)total_tiles * 32 + 4096);
    int* d_ids; CK(hipMalloc(&d_ids, ids.size() * 4));
    int* sink; CK(hipMalloc(&sink, 4));
    printf("%-28s %8s %8s %8s %8s   (TB/s gathered into LDS)\n", "table", "D1", "D2", "D3", "D4");
    {   // r03: whole lines per instruction against half lines of twice as many rows (what agnn_kernel / sddmm_kernel issue), Reddit image
        const long rows = 232965; const int rowb = 128;
        const size_t tbytes = (size_t)rows * rowb;
        char* d_table; CK(hipMalloc(&d_table, tbytes)); CK(hipMemset(d_table, 1, tbytes));
        unsigned long long s = 88172645463325252ULL;
        for (auto& v : ids) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int)(s % (unsigned long long)rows); }
        CK(hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
        printf("%-28s %8.2f %8.2f %8.2f %8.2f\n", "232965 x 128 B, whole lines", run<128, 1, 4>(d_ids, d_table, total_tiles, 3, sink), run<128, 2, 4>(d_ids, d_table, total_tiles, 3, sink),
               run<128, 3, 4>(d_ids, d_table, total_tiles, 3, sink), run<128, 4, 4>(d_ids, d_table, total_tiles, 3, sink));
        printf("%-28s %8.2f %8.2f %8.2f %8.2f\n", "232965 x 128 B, half lines", run<128, 1, 4, true>(d_ids, d_table, total_tiles, 3, sink), run<128, 2, 4, true>(d_ids, d_table, total_tiles, 3, sink),
               run<128, 3, 4, true>(d_ids, d_table, total_tiles, 3, sink), run<128, 4, 4, true>(d_ids, d_table, total_tiles, 3, sink));
        // range-major order: every wavefront walks the table's S slices in step (tile t of a wavefront draws its rows from slice
        // t S / tiles_per_wave), so what is gathered at any moment is 1/S of the table - L2-resident from S = 8 or 16 on
        const long waves = 256L * 8 * 4, tpw = total_tiles / waves;
        for (int S : {8, 16, 32}) {
            for (long gw = 0; gw < waves; ++gw)
                for (long t = 0; t < tpw; ++t) {
                    const long slice = t * S / tpw, lo = slice * rows / S, span = rows / S;
                    for (int k = 0; k < 32; ++k) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; ids[(size_t)(gw * tpw + t) * 32 + k] = (int)(lo + (long)(s % (unsigned long long)span)); }
                }
            CK(hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
            char name[64]; snprintf(name, sizeof name, "  in step over %d slices", S);
            printf("%-28s %8.2f %8.2f %8.2f %8.2f\n", name, run<128, 1, 4>(d_ids, d_table, total_tiles, 3, sink), run<128, 2, 4>(d_ids, d_table, total_tiles, 3, sink),
                   run<128, 3, 4>(d_ids, d_table, total_tiles, 3, sink), run<128, 4, 4>(d_ids, d_table, total_tiles, 3, sink));
            snprintf(name, sizeof name, "    the same, half lines");
            printf("%-28s %8.2f %8.2f %8.2f %8.2f\n", name, run<128, 1, 4, true>(d_ids, d_table, total_tiles, 3, sink), run<128, 2, 4, true>(d_ids, d_table, total_tiles, 3, sink),
                   run<128, 3, 4, true>(d_ids, d_table, total_tiles, 3, sink), run<128, 4, 4, true>(d_ids, d_table, total_tiles, 3, sink));
        }
        CK(hipFree(d_table));
        if (getenv("GATHER_BENCH_R03_ONLY")) return 0;
    }
    for (int rowb : {32, 128, 256}) {
        for (long rows : {29000L, 232965L, 2449029L}) {
            const size_t tbytes = (size_t)rows * rowb;
            char* d_table; CK(hipMalloc(&d_table, tbytes)); CK(hipMemset(d_table, 1, tbytes));
            unsigned long long s = 88172645463325252ULL;
            for (auto& v : ids) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int)(s % (unsigned long long)rows); }
            CK(hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
            char name[64]; snprintf(name, sizeof name, "%ld rows x %d B = %.1f MB", rows, rowb, tbytes / 1e6);
            double r1, r2, r3, r4;
            if (rowb == 32)       { r1 = run<32, 1, 4>(d_ids, d_table, total_tiles, 3, sink); r2 = run<32, 2, 4>(d_ids, d_table, total_tiles, 3, sink); r3 = run<32, 3, 4>(d_ids, d_table, total_tiles, 3, sink); r4 = run<32, 4, 4>(d_ids, d_table, total_tiles, 3, sink); }
            else if (rowb == 128) { r1 = run<128, 1, 4>(d_ids, d_table, total_tiles, 3, sink); r2 = run<128, 2, 4>(d_ids, d_table, total_tiles, 3, sink); r3 = run<128, 3, 4>(d_ids, d_table, total_tiles, 3, sink); r4 = run<128, 4, 4>(d_ids, d_table, total_tiles, 3, sink); }
            else                  { r1 = run<256, 1, 4>(d_ids, d_table, total_tiles, 3, sink); r2 = run<256, 2, 4>(d_ids, d_table, total_tiles, 3, sink); r3 = run<256, 3, 4>(d_ids, d_table, total_tiles, 3, sink); r4 = run<256, 4, 4>(d_ids, d_table, total_tiles, 3, sink); }
            printf("%-28s %8.2f %8.2f %8.2f %8.2f\n", name, r1, r2, r3, r4);
            CK(hipFree(d_table));
        }
    }
    return 0;
}
