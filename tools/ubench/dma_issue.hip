// What does ISSUING an LDS-DMA piece (buffer_load_dwordx4 ... lds, 64 lanes x 16 B = 1 KiB) cost the issuing wave?  One workgroup
// per CU (512 threads), every wave issues P pieces back to back, R rounds; s_memtime around the issue sequence only (the wait for
// the data is outside the stamps).  Variants:
//   A  every piece its own LDS base (a new M0 per piece)              B  one M0, pieces 1 KiB apart through the immediate offset
//   C  plain buffer_load_dwordx4 into VGPRs (no LDS)                  D  as A but dword pieces (256 B each)
//   E  as A with 8 independent v_fma between the pieces (is it a fixed wave-side stall or queue back-pressure?)
// All loads hit L2 (a 1 MiB window read over and over).
//   hipcc --offload-arch=gfx950 -O3 -o bin/dma_issue dma_issue.hip && bin/dma_issue   (bin/ is git-ignored; it ships to the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;

template <int VAR, int P>
__global__ __launch_bounds__(512, 2) void k(const char* src, unsigned long long* out, float* sink, int rounds, int active_waves) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 20, 0x00020000);
    unsigned long long tot = 0;
    float f = (float)lane;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 keep = {0, 0, 0, 0};
    if (wave < active_waves) {
        for (int r = 0; r < rounds; ++r) {
            const int voff = ((r * 8 + wave) * P * 1024 + lane * 16) & ((1 << 20) - 16384);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (VAR == 0 || VAR == 4)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(smem + (wave * P + p) * 1024 + (r & 1) * 65536), 16, voff + p * 1024, 0, 0, 0);
                else if (VAR == 1) {
                    lds_ptr L = (lds_ptr)(smem + (wave * P) * 1024 + (r & 1) * 65536);
                    switch (p) {
                        case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, L, 16, voff, 0, 0, 0); break;
                        case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, L, 16, voff, 0, 1024, 0); break;
                        case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, L, 16, voff, 0, 2048, 0); break;
                        default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, L, 16, voff, 0, 3072, 0); break;
                    }
                }
                else if (VAR == 2) {
                    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + p * 1024, 0, 0);
                    keep ^= v;
                } else if (VAR == 3)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(smem + (wave * P + p) * 1024 + (r & 1) * 65536), 4, voff + p * 1024, 0, 0, 0);
                if (VAR == 4) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) f = f * 1.0001f + 0.5f;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tot += t1 - t0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (lane == 0 && blockIdx.x == 17) out[wave] = tot;
    if (f == 123.25f || keep[0] == 0x12345u) sink[0] = f + smem[lane];
}

template <int VAR, int P>
void run(const char* tag, const char* src, unsigned long long* d, float* sink, int active) {
    const int lds = 131072, rounds = 64;
    hipFuncSetAttribute((const void*)k<VAR, P>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipMemset(d, 0, 64);
    hipLaunchKernelGGL((k<VAR, P>), dim3(256), dim3(512), lds, 0, src, d, sink, rounds, active);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    double s = 0;
    for (int w = 0; w < active; ++w) s += (double)h[w];
    printf("%-52s P=%d waves=%d: %7.1f cycles per piece (incl. ~%s stamp overhead / P)\n", tag, P, active, s / active / rounds / P, "2 s_memtime");
}

int main() {
    char* src;
    unsigned long long* d;
    float* sink;
    hipMalloc(&src, 1 << 20);
    hipMemset(src, 1, 1 << 20);
    hipMalloc(&d, 64);
    hipMalloc(&sink, 64);
    for (int active : {8, 1}) {
        run<0, 1>("A: new M0 per piece", src, d, sink, active);
        run<0, 2>("A: new M0 per piece", src, d, sink, active);
        run<0, 4>("A: new M0 per piece", src, d, sink, active);
        run<0, 8>("A: new M0 per piece", src, d, sink, active);
        run<1, 2>("B: one M0, immediate offsets", src, d, sink, active);
        run<1, 4>("B: one M0, immediate offsets", src, d, sink, active);
        run<2, 2>("C: plain buffer_load_dwordx4 -> VGPR", src, d, sink, active);
        run<2, 4>("C: plain buffer_load_dwordx4 -> VGPR", src, d, sink, active);
        run<2, 8>("C: plain buffer_load_dwordx4 -> VGPR", src, d, sink, active);
        run<3, 4>("D: dword LDS-DMA pieces (256 B)", src, d, sink, active);
        run<3, 8>("D: dword LDS-DMA pieces (256 B)", src, d, sink, active);
        run<4, 4>("E: A + 8 v_fma between pieces", src, d, sink, active);
    }
    return 0;
}
