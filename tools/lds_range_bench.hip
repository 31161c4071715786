// lds_range_bench.hip - feasibility probe for an LDS-resident column-range SpMM on gfx950.
//
// Question: instead of gathering one 128-byte fp16 row per edge from L2 (the range-blocked SpMM:
// 1.15e11 line requests/s, 1.04 ms at Reddit D = 64), let every CU STREAM each column range of X16
// into its LDS once (coalesced LDS-DMA) and gather the MFMA B operand from LDS with
// ds_read_b64_tr_b16 at per-lane row addresses.  Three per-CU resources then bound the kernel:
//   fill   : N * 128 B through the CU's L1/TA path (all CUs read the same range at the same time),
//   gather : slots * 128 B of LDS reads, times the bank-conflict factor of random row addresses,
//   mfma   : slots / 32 * NT v_mfma_f32_16x16x32_f16.
// This probe measures each alone and together.   hipcc --offload-arch=gfx950 -O3 -o lds_range_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ROWB = 128;    // bytes per fp16 row (D = 64)
constexpr int WAVES = 16;

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// MODE bit 0: fill (LDS-DMA of the next range), bit 1: compute (tr reads + MFMA), bit 2: random row ids
// (bank conflicts as they come) instead of ids whose low 3 bits are distinct inside every 32-lane pass,
// bit 3: per-tile 256-byte metadata DMA from a global stream, bit 4: skip the MFMAs (reads only)
template <int MODE, int RROWS>
__global__ __launch_bounds__(WAVES * 64) void probe(const char* __restrict__ x16, int nranges, int tiles, const char* __restrict__ meta,
                                                    float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUFB = RROWS * ROWB;
    constexpr int NDMA = BUFB / 1024 / WAVES;   // 1 KB DMA instructions per wave per range
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)((LDS_AS char*)smem);
    const uint32_t pad = lds0 + 2 * BUFB + wave * 256;
    floatx4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[j][s] = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 af;
#pragma unroll
    for (int j = 0; j < 8; ++j) af[j] = (_Float16)((lane + j) & 1);

    auto fill = [&](int r, int buf) {
#pragma unroll
        for (int q = 0; q < NDMA; ++q) {
            const int j = wave * NDMA + q;
            const int slot = j * 64 + lane;
            const int row = slot >> 3, cp = slot & 7;
            const int c = cp ^ (((row >> 1) & 3) << 1);
            const char* src = x16 + ((size_t)r * RROWS + row) * ROWB + c * 16;
            __builtin_amdgcn_global_load_lds((GLB_AS const void*)src, (LDS_AS void*)(uintptr_t)(lds0 + buf * BUFB + j * 1024), 16, 0, 0);
        }
    };
    if (MODE & 1) fill(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* mp = meta + ((size_t)blockIdx.x * WAVES + wave) * 256 + lane * 4;
    for (int r = 0; r < nranges; ++r) {
        const int buf = r & 1;
        if ((MODE & 1) && r + 1 < nranges) fill(r + 1, buf ^ 1);
        if (MODE & 2) {
            for (int t4 = 0; t4 < tiles; t4 += 4) {
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                const int t = t4 + tj;
                if (t >= tiles) break;
                if (MODE & 8) {
                    __builtin_amdgcn_global_load_lds((GLB_AS const void*)(mp + ((size_t)(r * tiles + t) * gridDim.x * WAVES) * 256),
                                                     (LDS_AS void*)(uintptr_t)pad, 4, 0, 0);
                }
                uint32_t ad[4][2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int K = 8 * g + 4 * h + (i >> 2);
                    uint32_t id = hash32((uint32_t)(r * 131 + t * 7 + wave * 1031 + blockIdx.x * 7919) * 32u + K);
                    if (MODE & 4) id &= (RROWS - 1);
                    else id = (id & (RROWS - 8)) | (uint32_t)((g & 1) * 4 + (i >> 2));
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        ad[s][h] = lds0 + buf * BUFB + id * ROWB + (((2 * s + ((i >> 1) & 1)) ^ (((id >> 1) & 3) << 1)) << 4) + (i & 1) * 8;
                }
                half4 lo[4], hi[4];
                asm volatile("ds_read_b64_tr_b16 %0, %8\n\t"
                             "ds_read_b64_tr_b16 %1, %9\n\t"
                             "ds_read_b64_tr_b16 %2, %10\n\t"
                             "ds_read_b64_tr_b16 %3, %11\n\t"
                             "ds_read_b64_tr_b16 %4, %12\n\t"
                             "ds_read_b64_tr_b16 %5, %13\n\t"
                             "ds_read_b64_tr_b16 %6, %14\n\t"
                             "ds_read_b64_tr_b16 %7, %15\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1]), "=&v"(lo[2]), "=&v"(hi[2]), "=&v"(lo[3]), "=&v"(hi[3])
                             : "v"(ad[0][0]), "v"(ad[0][1]), "v"(ad[1][0]), "v"(ad[1][1]), "v"(ad[2][0]), "v"(ad[2][1]), "v"(ad[3][0]), "v"(ad[3][1])
                             : "memory");
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const half8 bf = __builtin_shufflevector(lo[s], hi[s], 0, 1, 2, 3, 4, 5, 6, 7);
                    if (MODE & 16) { acc[tj][s][0] += (float)bf[0] + (float)bf[7]; }
                    else acc[tj][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[tj][s], 0, 0, 0);
                }
            }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) v += acc[j][s][0] + acc[j][s][1] + acc[j][s][2] + acc[j][s][3];
    if (!(MODE & 2)) v = *(const float*)(smem + lane * 4);
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = v;
}

template <int MODE, int RROWS>
static float run(const char* x16, int n_rows, int nwg, int tiles, const char* meta, float* out, const char* what) {
    const int nranges = n_rows / RROWS;
    const size_t lds = 2 * (size_t)RROWS * ROWB + WAVES * 256;
    CK(hipFuncSetAttribute((const void*)probe<MODE, RROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<MODE, RROWS>), dim3(nwg), dim3(WAVES * 64), lds, 0, x16, nranges, tiles, meta, out);
    CK(hipDeviceSynchronize());
    const int reps = 5;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe<MODE, RROWS>), dim3(nwg), dim3(WAVES * 64), lds, 0, x16, nranges, tiles, meta, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double fillB = (double)nwg * nranges * RROWS * ROWB, slots = (double)nwg * WAVES * nranges * tiles * 32.0;
    printf("%-46s R=%4d tiles/wave/range=%2d : %7.3f ms | fill %6.2f TB/s  lds-gather %6.1f TB/s  slots %.1f M  mfma %.0f TF\n", what, RROWS, tiles, ms,
           (MODE & 1) ? fillB / ms / 1e9 : 0.0, (MODE & 2) ? slots * 128 / ms / 1e9 : 0.0, slots / 1e6, (MODE & 2) ? slots * 16 * 64 * 2 / ms / 1e9 : 0.0);
    return ms;
}

int main(int argc, char** argv) {
    const int n_rows = 232960;   // Reddit, rounded down to a multiple of 1024
    const int nwg = argc > 1 ? atoi(argv[1]) : 256;
    char* x16; float* out; char* meta;
    CK(hipMalloc(&x16, (size_t)n_rows * ROWB));
    CK(hipMalloc(&out, (size_t)nwg * WAVES * 64 * 4));
    const size_t metaB = (size_t)nwg * WAVES * 256 * (n_rows / 512) * 8;
    CK(hipMalloc(&meta, metaB));
    CK(hipMemset(meta, 0, metaB));
    std::vector<uint16_t> h((size_t)n_rows * 64);
    for (size_t k = 0; k < h.size(); ++k) h[k] = 0x3c00 + (uint16_t)(k * 2654435761u >> 24);
    CK(hipMemcpy(x16, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    // Reddit D=64: 114.6 M edges over 14 561 windows; 64 windows per workgroup -> 228 workgroups; a (window, 512-row range)
    // cell holds 17.3 edges -> 1 tile; 16 waves x 4 windows -> 4 tiles per wave per range.  1024-row ranges: ~1.65 tiles per cell.
    run<1, 512>(x16, n_rows, nwg, 4, meta, out, "fill only");
    run<2, 512>(x16, n_rows, nwg, 4, meta, out, "compute only, conflict-free ids");
    run<2 | 4, 512>(x16, n_rows, nwg, 4, meta, out, "compute only, random ids");
    run<2 | 16, 512>(x16, n_rows, nwg, 4, meta, out, "reads only (no mfma), conflict-free ids");
    run<2 | 4 | 16, 512>(x16, n_rows, nwg, 4, meta, out, "reads only (no mfma), random ids");
    run<3, 512>(x16, n_rows, nwg, 4, meta, out, "fill + compute, conflict-free ids");
    run<3 | 4, 512>(x16, n_rows, nwg, 4, meta, out, "fill + compute, random ids");
    run<3 | 8, 512>(x16, n_rows, nwg, 4, meta, out, "fill + compute + metadata, conflict-free ids");
    run<3 | 4 | 8, 512>(x16, n_rows, nwg, 4, meta, out, "fill + compute + metadata, random ids");
    return 0;
}
