// Microbenchmark: how fast can one CU stream GEMM-shaped operand tiles (8 rows x 128 B pieces) through
// the vector-memory path, with no MFMA work at all?  Same access pattern as gemm.hip's staging.
//   mode 0: global_load_lds_dwordx4 (LDS-DMA)      mode 1: global_load_dwordx4 into VGPRs
// Build: hipcc --offload-arch=gfx950 -O3 -o vmem_rate vmem_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int MODE, int INFLIGHT>
__global__ __launch_bounds__(512, 1) void stream_kernel(const char* A, const char* W, int K, int nn, int nkt, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-contiguous tile order like the GEMM
    const int total = gridDim.x;
    const int q = total >> 3, r = total & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    int s = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int width = 8 * nn;
    const int pm = (s / width) * 8 + (s % width) % 8, pn = (s % width) / 8;
    const char* src[8];
    for (int i = 0; i < 8; ++i) {
        const int p = ((i & 3) * 8 + wave) * 64 + lane;
        const int row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
        const char* base = i < 4 ? A + (size_t)(pm * 256 + row) * K * 2 : W + (size_t)(pn * 256 + row) * K * 2;
        src[i] = base + c * 16;
    }
    u32x4 acc = {0, 0, 0, 0};
    for (int kt = 0; kt < nkt; ++kt) {
        char* dst = smem + (kt & 1) * 65536 + wave * 1024;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)kt * 128),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 8192), 16, 0, 0);
            } else {
                u32x4 v = *(const u32x4*)(src[i] + (size_t)kt * 128);
                acc ^= v;
            }
        }
        if (MODE == 0) {
            if (INFLIGHT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (INFLIGHT == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            if (INFLIGHT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 1 && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

int main(int argc, char** argv) {
    const int M = 8192, N = 8192, K = argc > 1 ? atoi(argv[1]) : 8192;
    char *A, *W;
    uint32_t* sink;
    hipMalloc(&A, (size_t)M * K * 2);
    hipMalloc(&W, (size_t)N * K * 2);
    hipMalloc(&sink, 4);
    hipMemset(A, 1, (size_t)M * K * 2);
    hipMemset(W, 2, (size_t)N * K * 2);
    const int nm = M / 256, nn = N / 256, nkt = K / 64, tiles = nm * nn;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), 131072, 0, A, W, K, nn, nkt, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 5;
            const double bytes = (double)tiles * nkt * 65536.0;
            // B/clk/CU quoted at 2.1 GHz nominal memory-side clock for orientation only
            printf("%-28s %8.3f ms  %7.2f TB/s chip  %6.1f GB/s/CU  (equiv GEMM rate if fetch-bound: %6.0f TF)\n", name, ms,
                   bytes / ms * 1e-9, bytes / ms * 1e-6 / 256, 2.0 * M * N * K / ms * 1e-9);
        }
    };
    run("lds-dma inflight 8", stream_kernel<0, 8>);
    run("lds-dma inflight 16", stream_kernel<0, 16>);
    run("lds-dma inflight 0", stream_kernel<0, 0>);
    run("vgpr loads", stream_kernel<1, 8>);
    return 0;
}
