// Micro-benchmark: how fast can one CU (16 wavefronts) take 48 KB ranges of an fp16 image into LDS -
// (a) by LDS-DMA (global_load_lds_dwordx4), (b) through registers (global_load_dwordx4 + ds_write_b128) - with every CU at it at once.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ingest_bench tools/ingest_bench.hip ; run: tools/bin/ingest_bench [MB of source]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define LDS_AS __attribute__((address_space(3)))
constexpr int kRange = 48 * 1024;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024, 1) void ingest_kernel(const char* src, size_t bytes, int nranges, int barrier, uint32_t* sink, int spread) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int tid = threadIdx.x;
    const size_t nr_src = bytes / kRange;
    uint32_t acc = 0;
    for (int r = 0; r < nranges; ++r) {
        const char* base = src + ((size_t)(spread ? r + (int)((blockIdx.x / 8) % spread) : r + blockIdx.x * 7) % nr_src) * kRange;   // spread 0: every workgroup elsewhere; k: the workgroups of an XCD k ranges apart at most
        char* dst = smem + (r & 1) * kRange;
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int off = (q * 1024 + tid) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                                 (LDS_AS void*)(dst + (q * 1024 + (tid & ~63)) * 16), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (MODE == 2) {
            // the flat kernel's filler: a struct buffer load (stride 32 bytes = one plane row, index = row, offset = half row) -> LDS
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)32, kRange / 32, 0x00020000);
            const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int blk = q * 16 + wave;
                __builtin_amdgcn_struct_ptr_buffer_load_lds(rs, (LDS_AS void*)(dst + blk * 1024), 16, blk * 32 + (lane >> 1), (lane & 1) * 16, 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u32x4 v[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) v[q] = *(const u32x4*)(base + (q * 1024 + tid) * 16);
#pragma unroll
            for (int q = 0; q < 3; ++q) *(u32x4*)(dst + (q * 1024 + tid) * 16) = v[q];
        }
        if (barrier) __syncthreads();
        if (DEPTH) acc += *(const uint32_t*)(dst + tid * 4);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atol(argv[1]) : 15;
    const size_t bytes = (mb << 20) / kRange * kRange;
    char* src; uint32_t* sink;
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&sink, 4);
    const int nranges = 300, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int spreads[] = {0, 1, 4};
    for (int mode = 0; mode < 3; ++mode) for (int barrier = 1; barrier < 2; ++barrier) for (int spread : spreads) {
        auto k = mode == 0 ? ingest_kernel<0, 1> : (mode == 1 ? ingest_kernel<1, 1> : ingest_kernel<2, 1>);
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kRange);
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(1024), 2 * kRange, 0, src, bytes, nranges, barrier, sink, spread);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double per_cu = (double)nranges * kRange / (best * 1e-3) / 2.4e9;
        printf("src %zu MB spread %d mode %s barrier %d: %.3f ms, %.1f B/clk/CU (2.4 GHz), aggregate %.2f TB/s  [%s]\n", mb, spread, mode == 2 ? "lds-dma struct-buffer" : (mode ? "regs+ds_write" : "lds-dma"), barrier, best,
               per_cu, (double)grid * nranges * kRange / (best * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
