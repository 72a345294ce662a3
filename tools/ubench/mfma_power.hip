// Microbenchmark: sustained MFMA rate at the chip's power limit, 32x32x16 vs 16x16x32 bf16, with operand
// registers that change from one MFMA to the next (random data) — register-resident, no memory traffic.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ inline unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

template <int SHAPE, int WAVES, int RANDOM>
__global__ __launch_bounds__(WAVES * 64, 1) void k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
            // random: uniform [-1, 1) values, different in every lane / register; otherwise all equal
            float va = RANDOM ? ((int)(lcg(seed) >> 8) - (1 << 23)) * (1.0f / (1 << 23)) : 0.5f;
            float vb = RANDOM ? ((int)(lcg(seed) >> 8) - (1 << 23)) * (1.0f / (1 << 23)) : 0.25f;
            a[i][j] = (__bf16)va;
            b[i][j] = (__bf16)vb;
        }
    float s = 0;
    if (SHAPE == 32) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 32; ++m)
                acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(m >> 2) & 7], b[m & 7], acc[m & 7], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i)
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 64; ++m)
                acc[m & 31] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(m >> 3) & 7], b[m & 7], acc[m & 31], 0, 0, 0);
        }
        for (int i = 0; i < 32; ++i)
            for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    if (s == 123.456f) out[0] = s;
}

template <int SHAPE, int WAVES, int RANDOM>
void run(const char* name, float* out) {
    const int iters = 4000 * (WAVES == 4 ? 2 : 1), blocks = 256 * 2;
    auto kern = k<SHAPE, WAVES, RANDOM>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), 131072, 0, out, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), 131072, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * 32 * 32 * 16 * 32.0 * iters * WAVES * blocks;  // both shapes: 32768*32 flops per iteration per wave
        printf("%-40s %7.1f TF  (%.2f ms)\n", name, flops / (ms * 1e-3) * 1e-12, ms);
    }
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    run<32, 4, 0>("32x32x16, 4 waves/CU-slot, constant", out);
    run<32, 4, 1>("32x32x16, 4 waves, random", out);
    run<16, 4, 0>("16x16x32, 4 waves, constant", out);
    run<16, 4, 1>("16x16x32, 4 waves, random", out);
    run<32, 8, 1>("32x32x16, 8 waves, random", out);
    run<16, 8, 1>("16x16x32, 8 waves, random", out);
    return 0;
}
