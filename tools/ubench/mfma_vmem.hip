// Microbenchmark: what does one LDS-DMA piece (global_load_lds_dwordx4) cost the matrix pipe?
// Every wave runs a stream of independent v_mfma_f32_32x32x16_bf16 with one piece inserted every
// `GAP` MFMAs (0 = none) and reports shader cycles per MFMA (s_memtime), for 1 or 2 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_vmem mfma_vmem.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// MODE 0: LDS-DMA, MODE 1: plain VGPR load, MODE 2: ds_read_b128 instead of a memory load, MODE 3: buffer_load .. lds,
// MODE 4: register-staged path — buffer_load_dwordx4 into a 4-deep VGPR ring, ds_write_b128 of the piece loaded 4 pieces ago
template <int WAVES, int GAP, int MODE, int STAGGER, int ACCA = 0>
__global__ __launch_bounds__(WAVES * 64, 1) void k(const char* src, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(0.001f * ((lane * 7 + i) % 13));
        b[i] = (__bf16)(0.002f * ((lane * 5 + i) % 11));
    }
    const char* p = src + ((size_t)blockIdx.x * WAVES + wave) * (8 * 16384) + (lane >> 3) * 16384 + (lane & 7) * 16;
    char* dst = smem + wave * 8192;
    const char* wbase = src + ((size_t)blockIdx.x * WAVES + wave) * (8 * 16384);
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, 8 * 16384, 0x00020000);
    const int voff = (lane >> 3) * 16384 + (lane & 7) * 16;
    u32x4 sink = {0, 0, 0, 0};
    u32x4 ring[4] = {};
    int npiece = 0;
    if (STAGGER) {
        for (int i = 0; i < wave * STAGGER; ++i) asm volatile("s_nop 15");
    }
    __builtin_amdgcn_s_barrier();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int piece = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            if (ACCA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m & 7]) : "v"(a), "v"(b));
            else acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 7], 0, 0, 0);
            if (GAP > 0 && (m % GAP) == GAP - 1) {
                if (MODE == 0) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)((it * 32 + m) & 127) * 128),
                                                     (__attribute__((address_space(3))) void*)(dst + ((m / GAP) & 7) * 1024), 16, 0, 0);
                } else if (MODE == 3) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + ((m / GAP) & 7) * 1024), 16, voff,
                                                         ((it * 32 + m) & 127) * 128, 0, 0);
                } else if (MODE == 4) {
                    const int slot = (m / GAP) & 3;      // compile-time after unrolling (32 % (4 GAP) == 0 for GAP in {2, 4, 8})
                    if (npiece >= 4) {
                        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                        asm volatile("ds_write_b128 %0, %1" ::"v"((uint32_t)(uintptr_t)(dst + ((m / GAP) & 7) * 1024 + lane * 16)),
                                     "v"(ring[slot])
                                     : "memory");
                    }
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ring[slot]) : "v"(voff), "s"(rsrc),
                                 "s"(((it * 32 + m) & 127) * 128) : "memory");
                    ++npiece;
                } else if (MODE == 1) {
                    sink ^= *(const u32x4*)(p + (size_t)((it * 32 + m) & 127) * 128);
                } else {
                    sink ^= *(const u32x4*)(dst + ((m / GAP) & 7) * 1024 + lane * 16);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 0 || MODE == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f || sink[0] == 0x1234567u) out[0] = s;
    if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
}

template <int WAVES, int GAP, int MODE, int STAGGER, int ACCA = 0>
void run(const char* name, const char* src, float* out, unsigned long long* cyc, int blocks) {
    const int iters = 200;
    auto kern = k<WAVES, GAP, MODE, STAGGER, ACCA>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), 65536, 0, src, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), 65536, 0, src, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * WAVES);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= h.size();
    const double per_mfma_simd = avg / (iters * 32.0) / (WAVES / 4);  // pipe cycles per MFMA on one SIMD
    const double tf = 2.0 * 32 * 32 * 16 * 32.0 * iters * WAVES * blocks / (ms * 1e-3) * 1e-12;
    printf("%-44s waves=%d gap=%2d  %6.1f cyc/MFMA/SIMD (ideal 32)  %7.1f TF  clk~%.2f GHz\n", name, WAVES, GAP, per_mfma_simd, tf,
           avg / (ms * 1e-3) * 1e-9);
}

int main() {
    char* src;
    float* out;
    unsigned long long* cyc;
    const int blocks = 256;
    hipMalloc(&src, (size_t)blocks * 8 * 8 * 16384 + (1 << 20));
    hipMemset(src, 1, (size_t)blocks * 8 * 8 * 16384 + (1 << 20));
    hipMalloc(&out, 4);
    hipMalloc(&cyc, blocks * 8 * 8);
    run<4, 0, 0, 0>("4 waves, MFMA only", src, out, cyc, blocks);
    run<8, 0, 0, 0>("8 waves, MFMA only", src, out, cyc, blocks);
    run<4, 8, 0, 0>("4 waves, glds every 8 MFMA", src, out, cyc, blocks);
    run<4, 4, 0, 0>("4 waves, glds every 4 MFMA (GEMM 4-wave rate)", src, out, cyc, blocks);
    run<4, 2, 0, 0>("4 waves, glds every 2 MFMA", src, out, cyc, blocks);
    run<8, 8, 0, 0>("8 waves, glds every 8 MFMA", src, out, cyc, blocks);
    run<8, 4, 0, 0>("8 waves, glds every 4 MFMA (GEMM 8-wave rate)", src, out, cyc, blocks);
    run<8, 2, 0, 0>("8 waves, glds every 2 MFMA", src, out, cyc, blocks);
    run<8, 4, 0, 1>("8 waves, glds every 4, stagger 16 cyc/wave", src, out, cyc, blocks);
    run<8, 4, 0, 2>("8 waves, glds every 4, stagger 32 cyc/wave", src, out, cyc, blocks);
    run<8, 4, 3, 0>("8 waves, buffer_load lds every 4", src, out, cyc, blocks);
    run<4, 4, 3, 0>("4 waves, buffer_load lds every 4", src, out, cyc, blocks);
    run<8, 8, 3, 0>("8 waves, buffer_load lds every 8", src, out, cyc, blocks);
    run<8, 0, 0, 0, 1>("8 waves, MFMA only, AGPR acc", src, out, cyc, blocks);
    run<8, 4, 0, 0, 1>("8 waves, glds every 4, AGPR acc", src, out, cyc, blocks);
    run<8, 4, 3, 0, 1>("8 waves, buffer_load lds every 4, AGPR acc", src, out, cyc, blocks);
    run<4, 4, 0, 0, 1>("4 waves, glds every 4, AGPR acc", src, out, cyc, blocks);
    run<8, 4, 4, 0>("8 waves, VGPR-staged (load + ds_write) every 4", src, out, cyc, blocks);
    run<4, 4, 4, 0>("4 waves, VGPR-staged (load + ds_write) every 4", src, out, cyc, blocks);
    run<8, 8, 4, 0>("8 waves, VGPR-staged (load + ds_write) every 8", src, out, cyc, blocks);
    run<8, 4, 4, 0, 1>("8 waves, VGPR-staged every 4, AGPR acc", src, out, cyc, blocks);
    run<8, 1, 2, 0>("8 waves, ds_read_b128 every MFMA", src, out, cyc, blocks);
    run<4, 1, 2, 0>("4 waves, ds_read_b128 every MFMA", src, out, cyc, blocks);
    return 0;
}
