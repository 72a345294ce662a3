// What does a workgroup COST when only one fits a CU?  10368 workgroups of 512 threads (the 96->96 slab launch) that do
// nothing / spin N barriers / touch LDS, with the LDS allocation of the slab kernel (155 KB -> 1 WG per CU) or a small one.
//   hipcc --offload-arch=gfx950 -O3 -o wg_launch wg_launch.hip && ./wg_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NBAR, int VGPRS>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    extern __shared__ char smem[];
    float acc[VGPRS];
#pragma unroll
    for (int i = 0; i < VGPRS; ++i) acc[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
        if (NBAR) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < VGPRS; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < VGPRS; ++i) s += acc[i];
    if (s == 123.456f) out[0] = s + smem[threadIdx.x];
}

template <int NBAR, int VGPRS>
void run(const char* tag, int grid, int lds, int iters, float* d) {
    hipFuncSetAttribute((const void*)k<NBAR, VGPRS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NBAR, VGPRS>), dim3(grid), dim3(512), lds, 0, d, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<NBAR, VGPRS>), dim3(grid), dim3(512), lds, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s grid %6d lds %6d iters %3d: %8.3f ms/launch = %7.2f us per WG-round (256 CUs)\n", tag, grid, lds, iters, ms / 10,
           ms / 10 * 1e3 / (grid / 256.0));
}

int main() {
    float* d;
    hipMalloc(&d, 1024);
    const int G = 10368;
    run<0, 8>("empty, small LDS", G, 1024, 0, d);
    run<0, 8>("empty, 155 KB LDS (1 WG/CU)", G, 158720, 0, d);
    run<0, 8>("empty, 64 KB LDS (2 WG/CU)", G, 65536, 0, d);
    run<1, 8>("54 barriers, 155 KB", G, 158720, 54, d);
    run<1, 8>("54 barriers, 64 KB", G, 65536, 54, d);
    run<0, 96>("96 live regs, 54 iters, 155 KB", G, 158720, 54, d);
    run<1, 96>("96 live regs + 54 barriers, 155 KB", G, 158720, 54, d);
    run<0, 8>("empty, 155 KB, 1296 WGs", 1296, 158720, 0, d);
    run<0, 8>("empty, 155 KB, 256 WGs", 256, 158720, 0, d);
    return 0;
}
