// probe_gfx950.hip - checks, on a real MI355X, the three hardware contracts the kernels rely on:
//   1. v_mfma_f32_16x16x32_f16 fragment layout (A row = lane&15, B col = lane&15, K slot
//      (lane>>4, j) shared by A and B; C/D: col = lane&15, row = 4*(lane>>4)+reg)
//   2. ds_read_b64_tr_b16: inside a 16-lane group, lane i receives element (i&3) of the 8-byte
//      pieces addressed by lanes 4j + (i>>2), j = 0..3
//   3. global_load_lds_dwordx4: lane l's 16 bytes land at LDS base + 16*l
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_gfx950.hip -o tools/bin/probe_gfx950
// Exit code 0 = all three hold; otherwise the observed mapping is printed.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 fp16x4_raw;
#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void mfma_probe(const _Float16* A /*16x32 row-major*/, const _Float16* B /*32x16 row-major*/, float* C /*16x16*/) {
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[i * 32 + 8 * g + j]; b[j] = B[(8 * g + j) * 16 + i]; }
    floatx4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + i] = c[r];
}

__global__ void tr_probe(float* out /*64x4*/) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[64 * 4];
    const int lane = threadIdx.x;
    for (int j = 0; j < 4; ++j) lds[lane * 4 + j] = (_Float16)(float)(lane * 4 + j);
    __syncthreads();
    fp16x4_raw v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4_raw*)(&lds[lane * 4]));
    half4 h = __builtin_bit_cast(half4, v);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (float)h[j];
}

__global__ void dma_probe(const _Float16* src /*64*8 halves*/, float* out /*512*/) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[64 * 8];
    const int lane = threadIdx.x;
    const int srclane = (lane * 7 + 3) & 63; // a permutation: lane l fetches piece srclane
    __builtin_amdgcn_global_load_lds((GLB_AS const void*)(src + srclane * 8), (LDS_AS void*)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < 8; ++j) out[lane * 8 + j] = (float)lds[lane * 8 + j];
}

int main() {
    int fails = 0;
    // ---------------- 1. MFMA layout (asymmetric operands)
    {
        std::vector<_Float16> A(16 * 32), B(32 * 16);
        std::vector<float> C(256), R(256, 0.f);
        for (int r = 0; r < 16; ++r) for (int k = 0; k < 32; ++k) A[r * 32 + k] = (_Float16)(float)((r * 3 + k * 5) % 7 - 3);
        for (int k = 0; k < 32; ++k) for (int c = 0; c < 16; ++c) B[k * 16 + c] = (_Float16)(float)((k * 2 + c * 7) % 9 - 4);
        for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) { float s = 0; for (int k = 0; k < 32; ++k) s += (float)A[r * 32 + k] * (float)B[k * 16 + c]; R[r * 16 + c] = s; }
        _Float16 *dA, *dB; float* dC;
        CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, 1024));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
        int bad = 0; for (int k = 0; k < 256; ++k) bad += C[k] != R[k];
        printf("[probe] mfma_f32_16x16x32_f16 layout: %s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);
        if (bad) { fails++; int bt = 0; for (int k = 0; k < 256; ++k) bt += C[(k % 16) * 16 + k / 16] != R[k]; printf("        transposed-output hypothesis: %d mismatches\n", bt); }
    }
    // ---------------- 2. transpose read
    {
        float* d; CK(hipMalloc(&d, 1024)); std::vector<float> o(256);
        hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d);
        CK(hipMemcpy(o.data(), d, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
            const int G = l >> 4, i = l & 15;
            const int srclane = 16 * G + 4 * j + (i >> 2), elem = i & 3;
            bad += o[l * 4 + j] != (float)(srclane * 4 + elem);
        }
        printf("[probe] ds_read_b64_tr_b16 semantics: %s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);
        if (bad) { fails++; for (int l = 0; l < 64; ++l) { printf("        lane %2d:", l); for (int j = 0; j < 4; ++j) { int v = (int)o[l * 4 + j]; printf(" (lane %2d,e%d)", v >> 2, v & 3); } printf("\n"); } }
    }
    // ---------------- 3. LDS DMA destination order
    {
        std::vector<_Float16> s(512); for (int k = 0; k < 512; ++k) s[k] = (_Float16)(float)k;
        _Float16* ds; float* d; CK(hipMalloc(&ds, 1024)); CK(hipMalloc(&d, 2048)); std::vector<float> o(512);
        CK(hipMemcpy(ds, s.data(), 1024, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(dma_probe, dim3(1), dim3(64), 0, 0, ds, d);
        CK(hipMemcpy(o.data(), d, 2048, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) bad += o[l * 8 + j] != (float)((((l * 7 + 3) & 63)) * 8 + j);
        printf("[probe] global_load_lds_dwordx4 lane-linear destination: %s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);
        if (bad) { fails++; for (int l = 0; l < 64; l += 8) printf("        slot %2d holds source piece %d (expected %d)\n", l, (int)o[l * 8] / 8, (l * 7 + 3) & 63); }
    }
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("[probe] device: %s, %d CUs, %.0f MHz, LDS/block %zu, arch %s\n", p.name, p.multiProcessorCount, p.clockRate / 1000.0, p.sharedMemPerBlock, p.gcnArchName);
    return fails ? 1 : 0;
}
