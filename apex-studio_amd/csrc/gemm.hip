// bf16 MFMA GEMM with fused epilogues for the denoise path's Linear layers.
//
//   C[M,N] = epi(A[M,K] * W[N,K]^T + bias[N])        (nn.Linear layout: W is [out, in])
//
// Replaces torch.nn.Linear at: to_q/to_k/to_v/add_*_proj/to_out/to_add_out
// (reference transformer/flux/base/model.py:106-129), ff.net.0.proj / ff.net.2 (:258-263),
// proj_mlp / proj_out of the single block (:180-182), x_embedder / context_embedder (:439-440),
// and the gate * y + residual tails of FluxTransformerBlock.forward (:296-297, :305-307).
//
// Tiling (gfx950): 128x128x64 block tile, 256 threads = 4 waves (2 x 2), each wave a 64x64
// sub-tile as 2x2 v_mfma_f32_32x32x16_bf16 accumulators.  Both operands are K-contiguous, so
// the A and W tiles are staged identically with 16-byte global_load_lds into a double-buffered
// 64 KiB LDS image; the XOR swizzle (chunk ^= row & 7) is applied on the per-lane SOURCE address
// and again on the ds_read_b128 address (the LDS destination of an LDS-DMA is lane-linear).
// Operands are fed swapped (MFMA "A" = W rows, "B" = activation rows) so a lane holds four
// consecutive output columns of one row -> 8-byte epilogue accesses.
// Block ids are remapped so each XCD owns a contiguous run of tiles, grouped 8 tiles tall.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + W
constexpr int GROUP_M = 8;

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw,
    const bf16_t* __restrict__ bias, bf16_t* C, int64_t ldc, int M, int N, int K,
    const float* __restrict__ gate, const bf16_t* R, int64_t ldr, int nm, int nn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- tile id -> (pm, pn): XCD-contiguous, grouped GROUP_M tall ----
    const int total = nm * nn;
    const int s = xcd_remap(blockIdx.x, total);
    const int width = GROUP_M * nn;
    const int group = s / width;
    const int first_m = group * GROUP_M;
    const int gsz = min(nm - first_m, GROUP_M);
    const int pm = first_m + (s % width) % gsz;
    const int pn = (s % width) / gsz;
    const int m0 = pm * BM, n0 = pn * BN;

    // ---- per-lane staging sources (row clamped at the edges; OOB rows are never stored) ----
    const char* a_src[4];
    const char* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = (wave * 4 + i) * 64 + lane;  // 16-byte chunk index inside the tile image
        const int row = p >> 3, pc = p & 7;
        const int c = pc ^ (row & 7);
        const int ar = min(m0 + row, M - 1);
        const int wr = min(n0 + row, N - 1);
        a_src[i] = (const char*)(A + (int64_t)ar * lda + c * 8);
        w_src[i] = (const char*)(W + (int64_t)wr * ldw + c * 8);
    }

    f32x16 acc[2][2];  // [nt][mt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nkt = K / BK;

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES + wave * 4096;
        const int64_t koff = (int64_t)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(a_src[i] + koff, base + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(w_src[i] + koff, base + TILE_BYTES + i * 1024);
    };

    // fragment read offsets (bytes) inside a tile image, for k-step 0; k-step ks adds a chunk xor
    int a_off[2], w_off[2], a_sw[2], w_sw[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ra = wm * 64 + t * 32 + l31;
        const int rw = wn * 64 + t * 32 + l31;
        a_off[t] = ra * 128;
        a_sw[t] = ra & 7;
        w_off[t] = rw * 128;
        w_sw[t] = rw & 7;
    }

    stage(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();  // tile kt landed (vmcnt(0) by the compiler) and the other buffer is free
        if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
        const char* As = smem + (kt & 1) * STAGE_BYTES;
        const char* Ws = As + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = ks * 2 + hi;
            bf16x8 af[2], wf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = *(const bf16x8*)(As + a_off[t] + ((c ^ a_sw[t]) << 4));
                wf[t] = *(const bf16x8*)(Ws + w_off[t] + ((c ^ w_sw[t]) << 4));
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] =
                        __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
    }

    // ---- epilogue: lane holds, per (nt, mt, g), C[m][n .. n+3] ----
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + wm * 64 + mt * 32 + l31;
        if (m >= M) continue;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + nt * 32 + 8 * g + 4 * hi;
                if (n >= N) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[nt][mt][4 * g + j];
                if (bias != nullptr) {
                    const u32x2 b = *(const u32x2*)(bias + n);
                    v[0] += bf16_lo(b[0]);
                    v[1] += bf16_hi(b[0]);
                    v[2] += bf16_lo(b[1]);
                    v[3] += bf16_hi(b[1]);
                }
                if (EPI == APEXMI_EPI_BIAS_GELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = gelu_tanh_f(v[j]);
                }
                if (EPI == APEXMI_EPI_BIAS_GATE_RES) {
                    const f32x4 gt = *(const f32x4*)(gate + n);
                    const u32x2 rr = *(const u32x2*)(R + (int64_t)m * ldr + n);
                    v[0] = bf16_lo(rr[0]) + gt[0] * v[0];
                    v[1] = bf16_hi(rr[0]) + gt[1] * v[1];
                    v[2] = bf16_lo(rr[1]) + gt[2] * v[2];
                    v[3] = bf16_hi(rr[1]) + gt[3] * v[3];
                }
                u32x2 o;
                o[0] = pack_bf16(v[0], v[1]);
                o[1] = pack_bf16(v[2], v[3]);
                *(u32x2*)(C + (int64_t)m * ldc + n) = o;
            }
        }
    }
}

template <int EPI>
int launch_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, void* C,
                int64_t ldc, int M, int N, int K, const float* gate, const void* R, int64_t ldr,
                hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
        attr_set = true;
    }
    const int nm = (M + BM - 1) / BM, nn = (N + BN - 1) / BN;
    hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(nm * nn), dim3(256), 2 * STAGE_BYTES, stream,
                       (const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias,
                       (bf16_t*)C, ldc, M, N, K, gate, (const bf16_t*)R, ldr, nm, nn);
    return apexmi_check_launch("gemm_bf16");
}

}  // namespace

extern "C" int apexmi_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw,
                                const void* bias, void* C, int64_t ldc, int M, int N, int K,
                                int epilogue, const float* gate, const void* R, int64_t ldr,
                                apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(A && W && C, "gemm_bf16: null operand");
    APEXMI_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_bf16: empty problem M=%d N=%d K=%d", M, N, K);
    APEXMI_REQUIRE(K % BK == 0, "gemm_bf16: K=%d must be a multiple of %d", K, BK);
    APEXMI_REQUIRE(N % 8 == 0, "gemm_bf16: N=%d must be a multiple of 8", N);
    APEXMI_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0,
                   "gemm_bf16: leading dimensions must keep rows 16-byte aligned");
    APEXMI_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)C % 8) == 0,
                   "gemm_bf16: operands must be 16-byte aligned");
    if (epilogue == APEXMI_EPI_BIAS_GATE_RES) {
        APEXMI_REQUIRE(gate && R, "gemm_bf16: gate/residual epilogue needs gate and R");
        APEXMI_REQUIRE(ldr % 4 == 0 && ((uintptr_t)gate % 16) == 0 && ((uintptr_t)R % 8) == 0,
                       "gemm_bf16: gate/R alignment");
    }
    ApexmiProfScope prof(0, stream, 2.0 * M * N * (double)K,
                         2.0 * ((double)M * K + (double)N * K + (double)M * N));
    switch (epilogue) {
        case APEXMI_EPI_BIAS:
            return launch_gemm<APEXMI_EPI_BIAS>(A, lda, W, ldw, bias, C, ldc, M, N, K, gate, R, ldr,
                                                stream);
        case APEXMI_EPI_BIAS_GELU:
            return launch_gemm<APEXMI_EPI_BIAS_GELU>(A, lda, W, ldw, bias, C, ldc, M, N, K, gate, R,
                                                     ldr, stream);
        case APEXMI_EPI_BIAS_GATE_RES:
            return launch_gemm<APEXMI_EPI_BIAS_GATE_RES>(A, lda, W, ldw, bias, C, ldc, M, N, K, gate,
                                                         R, ldr, stream);
        default:
            apexmi_set_error("gemm_bf16: unknown epilogue %d", epilogue);
            return 1;
    }
}
