// bf16 MFMA GEMM with fused epilogues for the denoise path's Linear layers.
//
//   C[M,N] = epi(A[M,K] * W[N,K]^T + bias[N])        (nn.Linear layout: W is [out, in])
//
// Replaces torch.nn.Linear at: to_q/to_k/to_v/add_*_proj/to_out/to_add_out
// (reference transformer/flux/base/model.py:106-129), ff.net.0.proj / ff.net.2 (:258-263),
// proj_mlp / proj_out of the single block (:180-182), x_embedder / context_embedder (:439-440),
// and the gate * y + residual tails of FluxTransformerBlock.forward (:296-297, :305-307).
//
// One kernel template, five tilings (gfx950, BK = 64); `apexmi_tune_set("gemm.config", n)` forces one:
//   1 CFG_128    : 128x128 block, 4 waves (2x2) of 64x64, 2 blocks/CU — small / ragged problems
//   2 CFG_256    : 256x256 block, 8 waves (2x4) of 128x64, plain double buffer (baseline for A/B)
//   3 CFG_256P   : ping-pong schedule on v_mfma_f32_32x32x16_bf16: the K-tile is cut into 4 quadrant
//                  phases of {ds_read sub-tile | barrier | 8 MFMA | barrier}; the M-halves of the block
//                  (waves w and w+4 share a SIMD) run one barrier apart, so on every SIMD one wave is in
//                  its MFMA segment while its partner is in its LDS/DMA segment, and the next K-tile's
//                  LDS-DMA stays in flight for three phases behind a counted wait.
//   7 CFG_256P16 : the same schedule on v_mfma_f32_16x16x32_bf16 — SHIPPED for the large GEMMs.  The
//                  denoise step runs at the chip's power limit, and at equal matrix-pipe occupancy the
//                  16x16x32 form sustains 2.03 GHz against 1.79 GHz for 32x32x16 (tools/ubench/
//                  mfma_power.hip); in the Flux step it is 8 % faster per GEMM, 6.5 % per step.
//   9 CFG_256R   : free-running ring on v_mfma_f32_16x16x32_bf16 (round 4): 32-deep sub-tiles in a four-slot LDS ring, fragments
//                  read one step ahead into a second register set, two barriers per K-tile — see the SCHED 7 branch.
//   6 CFG_256W   : 4 waves (2x2) of 128x128, one wave per SIMD, accumulators in AGPRs, LDS-DMA through
//                  buffer_load with scalar piece offsets.  17 % fewer cycles than CFG_256P and faster in
//                  an isolated loop, but not in the step: the chip answers the denser instruction
//                  stream with a lower clock (1.30 vs 1.57 GHz under the profiler).  Kept for the record
//                  and for shapes that are not power-bound.
// Both operands are K-contiguous, so A and W tiles are staged identically with 16-byte
// global_load_lds into a double-buffered LDS image; the XOR swizzle (chunk ^= (row >> 1) & 7: two 128-byte rows share a 256-byte bank
// row, and a 32-row MFMA fragment read must spread each 16-lane group over all 16 slots) is applied on
// the per-lane SOURCE address and again on the ds_read_b128 address (an LDS-DMA destination is
// lane-linear).  Operands are fed swapped (MFMA "A" = W rows, "B" = activation rows) so a lane holds
// four consecutive output columns of one row -> 8-byte epilogue accesses.
// Up to 4 problems sharing (N, K, epilogue) run in ONE launch (the img / txt streams of a double
// block), tile ids are remapped so each XCD owns a contiguous run of tiles, grouped GROUP_M tiles tall.
#include "common.h"

#include <cstring>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <vector>

namespace {

// Ablation build of the ring schedule (tools/gemm_ring_ablate.sh; WRONG results, timing only): bit 1 no barriers, 2 no vmcnt waits,
// 4 no LDS-DMA pieces, 8 no fragment reads.  0 in every shipped library.
#ifndef APEXMI_GEMM_ABLATE
#define APEXMI_GEMM_ABLATE 0
#endif
// Stream-K launches (round 4 experiment, profiles/r04_gemm_streamk_ab.log): compiled in only with -DAPEXMI_GEMM_STREAMK=1
// (APEXMI_GEMM_STREAMK=1 python -m apex_studio_amd.build).  Correct and deterministic, but 6 .. 40 % SLOWER than the tile launch:
// the persistent workgroups lose the dispatcher's dynamic load balancing (+8 us per tile even without a split tile) and a split
// tile's hand-over costs ~15 us.  The shipped library does not contain the persistent path (its kernel allocates scratch).
// Those figures are from the commit that built it (git log -S APEXMI_GEMM_STREAMK); since the epilogue rewrite later in round 4
// (one instantiation per activation mode, both residual slabs pre-loaded) the persistent item loop spills 64-191 VGPRs and runs at
// half the tile launch's rate (profiles/r04_gemm_persistent_probe_b.log) — the build flag is kept for the record, not maintained.
// 2 (shipped, round 6): the pieces of the 256x256 ping-pong kernel go out in the scalar-base form of global_load_lds; 1: the builtin's
// 64-bit vector addresses (round 5's form; the A/B arm of profiles/r06_gemm_peel_ab.log)
#ifndef APEXMI_GEMM_PEEL
#define APEXMI_GEMM_PEEL 2
#endif
#ifndef APEXMI_GEMM_STREAMK
#define APEXMI_GEMM_STREAMK 0
#endif
// Per-workgroup timeline of the shipped schedule (tools/gemm_tile_trace.py, profiles/r04_gemm_tile_trace.log): compiled in only with
// -DAPEXMI_GEMM_TRACE=1 into a side library; tune keys "gemm.trace_lo" / "gemm.trace_hi" = halves of a device pointer to 8 u64 per
// workgroup: {HW_ID, XCC_ID, t_entry, t_loop_begin, t_loop_end, t_stores_issued, t_stores_acknowledged} in s_memrealtime ticks (10 ns).
#ifndef APEXMI_GEMM_TRACE
#define APEXMI_GEMM_TRACE 0
#endif
constexpr int BK = 64;
constexpr int GROUP_M = 6;   // tiles tall per group: an XCD's 32 concurrent tiles as ~6 x 5.3 (squarer than 8 x 4: fewer panel fetches per
                             // tile; interleaved A/B, profiles/r03_ab_gemm_group_m.log: Flux -0.4 %, Qwen -1.1 %, Wan -0.6 % vs 8)
// internal epilogue class of the f32-storage verification mode (APEXMI_EPI_F32_IO): gate * y + residual with float C / R.
// Together with APEXMI_EPI_BIAS_F32 (which also carries the activation flag) it covers every epilogue with float I/O.
constexpr int EPI_GATE_RES_F32 = 7;
constexpr int MAX_GROUPS = 4;

struct GemmProblem {
    const bf16_t* A;
    const bf16_t* W;
    const bf16_t* bias;
    bf16_t* C;
    const float* gate;
    const bf16_t* R;
    int64_t lda, ldw, ldc, ldr;
    int M, N, nm, nn, tile0, gelu;
    // fused q/k/v preparation (apexmi_gemm_bf16_grouped_qkv): this problem is a fused QKV projection, N = 3 H 128; its rows
    // are rows [row0, row0 + M) of the joint sequence; nq / nk = the per-head RMSNorm weights of ITS stream.  C is not written.
    int qkv, row0;
    const bf16_t* nq;
    const bf16_t* nk;
};
struct QkvShared {           // outputs of the fused preparation, shared by the problems of a launch
    bf16_t* qo;              // [H, S_out, 128]
    bf16_t* ko;              // [H, S_out, 128]
    bf16_t* vt;              // [H, 128, Skp]
    const float* rope;       // [2, S_out, 128] f32 (cos | sin), interleaved pairs
    int S_out, Skp, inner;   // inner = H * 128
    float eps;
    const float* rope_pairs; // optional [2, S_out, 64]: every second entry of `rope` when its entries come in equal pairs (half the bytes)
};
struct GemmGroup {
    GemmProblem p[MAX_GROUPS];
    int count, K, total, group_m;
    // batched mode (count == 1): blockIdx.y selects the batch element; strides in elements of A / W, bytes of C
    int batch;
    int64_t bsA, bsW, bsC_bytes;
    QkvShared qs;
    // live clock probe (apexmi_clk_enable): when non-null, every workgroup adds the shader cycles (s_memtime) and the 100 MHz
    // reference ticks (s_memrealtime) its K-loop took to clk[0] / clk[1]: sum(cycles) / sum(ticks) x 100 MHz = the effective
    // shader clock WHILE this kernel runs inside the real step (the number the "power-bound" argument of DESIGN.md rests on)
    unsigned long long* clk;
#if APEXMI_GEMM_TRACE
    unsigned long long* trace;
#endif
    // EXPERIMENT (tune key gemm.wpacked, tools/gemm_wpacked_ab.py): every W operand of the launch is TILE-MAJOR packed —
    // [N / 256][K / 64][256 rows][64] — so that an LDS-DMA piece (8 rows x 128 B) is 1 KiB contiguous; SCHED 5 only
    int wpacked;
    // stream-K (round 4; SCHED 5 launches whose last round of 256 tiles is part-filled, `gemm.streamk`): the launch is 256
    // PERSISTENT workgroups; the sk_r tiles past the last full round are cut along K into 256 equal unit ranges (see
    // gemm_bf16_kernel), partial sums travel through sk_slab (256 KiB of f32 per workgroup) guarded by sk_flag
    int sk_r, sk_tfull;
    float* sk_slab;
    unsigned* sk_flag;
};
// what a call of gemm_tile() does with its accumulators
enum { TILE_FULL = 0, TILE_WRITER = 1, TILE_OWNER = 2 };
struct SkCtx {
    int mode, kt0, kt1;          // K-tile range [kt0, kt1) of this call
    int cu;                      // this workgroup's slot: WRITER stores slab[cu] and raises flag[cu]
    int nparts;                  // OWNER: adds slab[cu - 1] .. slab[cu - nparts] (the workgroups holding the tile's earlier K-tiles)
};

template <int BM_, int BN_, int WM_, int WN_, int SCHED_>
struct Cfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    // 0 plain double buffer | 1 ping-pong phases, 32x32x16 | 4 one wave per SIMD, rotated pipeline | 5 ping-pong, 16x16x32 |
    // 7 free-running ring, 16x16x32
    static constexpr int SCHED = SCHED_;
    static constexpr bool PP = SCHED_ == 1;
    static constexpr int NW = WM * WN, NT = NW * 64;
    static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;  // 32x32 MFMA tiles per wave
    static constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2;
    static constexpr int STAGE = A_BYTES + W_BYTES;
    static constexpr int LDS = SCHED_ == 8 ? 5 * 32768 : 2 * STAGE;   // SCHED 8: five-slot ring (160 KiB, the whole LDS of a CU)
    static constexpr int A_LD = BM * 8 / NT, W_LD = BN * 8 / NT;  // glds per thread per K-tile
    static constexpr int OCC = (LDS <= 80 * 1024 && NT == 256) ? 2 : (NT == 512 ? 2 : 1);  // waves per EU for launch_bounds
};
using CFG_128 = Cfg<128, 128, 2, 2, 0>;
using CFG_256 = Cfg<256, 256, 2, 4, 0>;
using CFG_128E = Cfg<128, 128, 2, 4, 0>;   // 128x128 block on EIGHT waves (64x32 each): half the LDS-DMA pieces per wave and K-tile
using CFG_256P = Cfg<256, 256, 2, 4, 1>;
using CFG_256W = Cfg<256, 256, 2, 2, 4>;
using CFG_256P16 = Cfg<256, 256, 2, 4, 5>;
using CFG_256R = Cfg<256, 256, 2, 4, 7>;     // free-running ring schedule (round 4), `gemm.large = 9`: four slots, pieces 3 sub-tiles ahead
using CFG_256R5 = Cfg<256, 256, 2, 4, 8>;    // the same with five slots, pieces 4 sub-tiles ahead (`gemm.large = 10`)

// activation of the bias epilogue: 1 gelu(tanh), 2 gelu(erf, torch nn.GELU() default), 3 silu
APEXMI_DEVICE float act_f(float x, int mode) {
    if (mode == 1) return gelu_tanh_f(x);
    if (mode == 2) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    if (mode == 4) return x_sigmoid_log2(x, (1.702f * 1.4426950408889634f) * x);   // quick_gelu (CLIP text MLP)
    return silu_f(x);
}
// the same with the mode known at compile time.  The epilogues dispatch on the (block-uniform) mode ONCE, outside their loops
// (APEXMI_ACT_DISPATCH): with act_f(x, P.gelu) per element hipcc emitted a four-way scalar branch chain around every one of a lane's
// 128 values (1577 branches in the 256 x 256 kernel), so no two values' exp / rcp chains ever overlapped.
template <int ACT>
APEXMI_DEVICE float act_c(float x) {
    if constexpr (ACT == 0) return x;
    else return act_f(x, ACT);
}
#define APEXMI_ACT_DISPATCH(mode, ...)                                    \
    switch (mode) {                                                       \
        case 1: { constexpr int ACT = 1; __VA_ARGS__; } break;            \
        case 2: { constexpr int ACT = 2; __VA_ARGS__; } break;            \
        case 3: { constexpr int ACT = 3; __VA_ARGS__; } break;            \
        case 4: { constexpr int ACT = 4; __VA_ARGS__; } break;            \
        default: { constexpr int ACT = 0; __VA_ARGS__; } break;           \
    }
inline int act_mode(int epilogue) {
    return epilogue == APEXMI_EPI_BIAS_GELU ? 1 : epilogue == APEXMI_EPI_BIAS_GELU_ERF ? 2 : epilogue == APEXMI_EPI_BIAS_SILU ? 3 : epilogue == APEXMI_EPI_BIAS_QUICK_GELU ? 4 : 0;
}

// exchange so that (a, b) = this lane's two 4-column groups (8g.., 8(g+1)..) become 8 CONSECUTIVE
// columns: low half-wave gets [a_lo | a_hi] = cols 8g..8g+7, high half-wave [b_lo | b_hi] = cols
// 8(g+1)..8(g+1)+7.  The exchange is an involution, so it also maps a 16-byte residual load back to
// the accumulator layout.
APEXMI_DEVICE void swap_pair(u32x2& a, u32x2& b) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        auto r = __builtin_amdgcn_permlane32_swap(a[i], b[i], false, false);
        a[i] = r[0];
        b[i] = r[1];
    }
}

// Epilogue for one n-tile (32 columns) of a wave: ALL loads (bias, gate, residual rows of every
// m-tile) are issued before the first store, so the wave pays one memory round trip per n-tile instead
// of one per 8-byte group (C may alias R, so the compiler cannot hoist loads above stores itself).
// Accumulator layout: lane holds C[m][nbase + 8 g + 4 hi + (0..3)], g = 0..3; pairs of groups are
// exchanged with the other half-wave into 8 consecutive columns -> 16-byte accesses.
template <int EPI, int TM, int ACT = 0>
APEXMI_DEVICE void store_ntile(const f32x16 (&acc)[TM], const GemmProblem& P, int N, const int (&m)[TM],
                               int nbase, int hi) {
    float bs[4][4];
    f32x4 gt[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        // loads are unconditional on clamped addresses (a predicated load makes hipcc branch around
        // it and drain vmcnt(0) per element); out-of-range lanes compute garbage and store nothing
        const int n = min(nbase + 8 * g + 4 * hi, N - 4);
        u32x2 b = {0u, 0u};
        if (P.bias != nullptr) b = *(const u32x2*)(P.bias + n);  // uniform condition
        bs[g][0] = bf16_lo(b[0]);
        bs[g][1] = bf16_hi(b[0]);
        bs[g][2] = bf16_lo(b[1]);
        bs[g][3] = bf16_hi(b[1]);
        if (EPI == APEXMI_EPI_BIAS_GATE_RES || EPI == EPI_GATE_RES_F32) gt[g] = *(const f32x4*)(P.gate + n);
    }
    if (EPI == APEXMI_EPI_BIAS_F32 || EPI == EPI_GATE_RES_F32) {
        // C (and R) are float[M][ld]: the lane's 4 consecutive columns of each group, no exchange.  Same formulas as the
        // bf16 path below, minus the rounding at the store (attention scores; the f32-storage verification mode).
        float* Cf = (float*)P.C;
        const float* Rf = (const float*)P.R;
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + 8 * g + 4 * hi;
                if (m[mt] >= 0 && n < N) {
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = acc[mt][4 * g + j] + bs[g][j];
                        o[j] = act_c<ACT>(o[j]);
                    }
                    if (EPI == EPI_GATE_RES_F32) {
                        const f32x4 r = *(const f32x4*)(Rf + (int64_t)m[mt] * P.ldr + n);
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = r[j] + gt[g][j] * o[j];
                    }
                    *(f32x4*)(Cf + (int64_t)m[mt] * P.ldc + n) = f32x4{o[0], o[1], o[2], o[3]};
                }
            }
        return;
    }
    u32x4 rr[TM][2];
    if (EPI == APEXMI_EPI_BIAS_GATE_RES) {
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int nst = min(nbase + 8 * (2 * pr + hi), N - 8);
                rr[mt][pr] = *(const u32x4*)(P.R + (int64_t)max(m[mt], 0) * P.ldr + nst);
            }
    }
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int g0 = 2 * pr;
            const int nst = nbase + 8 * (g0 + hi);  // first of the 8 columns this lane stores
            float v[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[q][j] = acc[mt][4 * (g0 + q) + j] + bs[g0 + q][j];
                    v[q][j] = act_c<ACT>(v[q][j]);
                }
            if (EPI == APEXMI_EPI_BIAS_GATE_RES) {
                u32x2 ra = {rr[mt][pr][0], rr[mt][pr][1]}, rb = {rr[mt][pr][2], rr[mt][pr][3]};
                swap_pair(ra, rb);  // 16-byte row segment -> accumulator layout
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const u32x2 r2 = q ? rb : ra;
                    v[q][0] = bf16_lo(r2[0]) + gt[g0 + q][0] * v[q][0];
                    v[q][1] = bf16_hi(r2[0]) + gt[g0 + q][1] * v[q][1];
                    v[q][2] = bf16_lo(r2[1]) + gt[g0 + q][2] * v[q][2];
                    v[q][3] = bf16_hi(r2[1]) + gt[g0 + q][3] * v[q][3];
                }
            }
            u32x2 oa = {pack_bf16(v[0][0], v[0][1]), pack_bf16(v[0][2], v[0][3])};
            u32x2 ob = {pack_bf16(v[1][0], v[1][1]), pack_bf16(v[1][2], v[1][3])};
            swap_pair(oa, ob);
            if (m[mt] >= 0 && nst < N) {
                const u32x4 o = {oa[0], oa[1], ob[0], ob[1]};
                *(u32x4*)(P.C + (int64_t)m[mt] * P.ldc + nst) = o;
            }
        }
}


typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// x[lane] + x[lane ^ 16]: with both operands a copy of x the swap leaves {r0, r0, r2, r2} and {r1, r1, r3, r3} (rows of 16 lanes)
APEXMI_DEVICE float sum_xor16(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// v_permlane16_swap on a register pair: rows of 16 lanes, odd rows of `a` <-> even rows of `b`
APEXMI_DEVICE void swap16(uint32_t& a, uint32_t& b) {
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

// Epilogue for one 32-column slab (two 16x16 n-tiles x, y) of a wave that accumulated with
// v_mfma_f32_16x16x32_bf16 (operands swapped: D rows = output columns).  Accumulator layout: lane (g = lane >> 4,
// c = lane & 15) holds C[m = mtile*16 + c][n = nbase + 16 t + 4 g + (0..3)] for t = 0 (x), 1 (y).  One
// v_permlane16_swap per dword pair turns that into 8 consecutive columns per lane, starting at
// nbase + 16 (g & 1) + 8 (g >> 1), so stores and residual loads are 16 bytes wide.
template <int EPI, int MT, int ACT = 0>
APEXMI_DEVICE void store_slab16(const f32x4_t (&x)[MT], const f32x4_t (&y)[MT], const GemmProblem& P, int N,
                                const int (&m)[MT], int nbase, int g, const u32x4* rr_pre = nullptr) {
    float bs[2][4];
    f32x4 gt[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = min(nbase + 16 * t + 4 * g, N - 4);
        u32x2 b = {0u, 0u};
        if (P.bias != nullptr) b = *(const u32x2*)(P.bias + n);
        bs[t][0] = bf16_lo(b[0]);
        bs[t][1] = bf16_hi(b[0]);
        bs[t][2] = bf16_lo(b[1]);
        bs[t][3] = bf16_hi(b[1]);
        if (EPI == APEXMI_EPI_BIAS_GATE_RES || EPI == EPI_GATE_RES_F32) gt[t] = *(const f32x4*)(P.gate + n);
    }
    if (EPI == APEXMI_EPI_BIAS_F32 || EPI == EPI_GATE_RES_F32) {  // float C / R: 4 consecutive columns per tile, no exchange
        float* Cf = (float*)P.C;
        const float* Rf = (const float*)P.R;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n = nbase + 16 * t + 4 * g;
                const f32x4_t& a = t ? y[mt] : x[mt];
                if (m[mt] >= 0 && n < N) {
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = a[j] + bs[t][j];
                        o[j] = act_c<ACT>(o[j]);
                    }
                    if (EPI == EPI_GATE_RES_F32) {
                        const f32x4 r = *(const f32x4*)(Rf + (int64_t)m[mt] * P.ldr + n);
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = r[j] + gt[t][j] * o[j];
                    }
                    *(f32x4*)(Cf + (int64_t)m[mt] * P.ldc + n) = f32x4{o[0], o[1], o[2], o[3]};
                }
            }
        return;
    }
    const int nst = nbase + 16 * (g & 1) + 8 * (g >> 1);  // first of the 8 columns this lane stores
    u32x4 rr[MT];
    if (EPI == APEXMI_EPI_BIAS_GATE_RES) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            rr[mt] = rr_pre ? rr_pre[mt] : *(const u32x4*)(P.R + (int64_t)max(m[mt], 0) * P.ldr + min(nst, N - 8));
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float v[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[0][j] = x[mt][j] + bs[0][j];
            v[1][j] = y[mt][j] + bs[1][j];
            v[0][j] = act_c<ACT>(v[0][j]);
            v[1][j] = act_c<ACT>(v[1][j]);
        }
        if (EPI == APEXMI_EPI_BIAS_GATE_RES) {
            uint32_t r0 = rr[mt][0], r1 = rr[mt][1], r2 = rr[mt][2], r3 = rr[mt][3];
            swap16(r0, r2);  // 16-byte row segment -> accumulator layout (the exchange is an involution)
            swap16(r1, r3);
            const uint32_t ra[2][2] = {{r0, r1}, {r2, r3}};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                v[t][0] = bf16_lo(ra[t][0]) + gt[t][0] * v[t][0];
                v[t][1] = bf16_hi(ra[t][0]) + gt[t][1] * v[t][1];
                v[t][2] = bf16_lo(ra[t][1]) + gt[t][2] * v[t][2];
                v[t][3] = bf16_hi(ra[t][1]) + gt[t][3] * v[t][3];
            }
        }
        uint32_t x0 = pack_bf16(v[0][0], v[0][1]), x1 = pack_bf16(v[0][2], v[0][3]);
        uint32_t y0 = pack_bf16(v[1][0], v[1][1]), y1 = pack_bf16(v[1][2], v[1][3]);
        swap16(x0, y0);
        swap16(x1, y1);
        if (m[mt] >= 0 && nst < N) {
            const u32x4 o = {x0, x1, y0, y1};
            *(u32x4*)(P.C + (int64_t)m[mt] * P.ldc + nst) = o;     // (non-temporal stores: no difference, profiles/r04_gemm_tile_trace_*)
        }
    }
}

// Fused q/k/v preparation in the epilogue of the 256x256 tile on v_mfma_f32_16x16x32_bf16 (what qkv_prepare_fused_kernel,
// elementwise.hip, does in a separate pass over the [S, 3 H 128] projection: reference transformer/flux/base/attention.py:62-94
// — unflatten, norm_q / norm_k, apply_rotary_emb, the [B, H, S, D] layout, and V^T for the attention kernel's PV product).
// A 256-column tile is two whole heads of q, of k or of v (256 | H 128); a head's 128 columns sit in two waves (wn, wn ^ 1).
//   q / k: y = bf16(acc + bias) — the value the unfused path stores — then per (row, head) sum of squares IN THE SAME ORDER as
//          qk_norm_rope4_body (8 columns in the lane, then the butterfly over the 16 eight-column chunks: chunk ^ 8 is the
//          partner wave (through LDS), chunk ^ 4 the lane's other 32-column slab, chunk ^ 2 / ^ 1 lanes ^ 16 / ^ 32), the same
//          rsqrt, weight and interleaved rotation: bit-identical outputs, written 16 bytes per lane into [H, S_out, 128].
//   v:     the bf16 tile goes through the (now free) staging LDS, rotated per row, and leaves transposed: 8 consecutive
//          sequence positions per lane, 128-byte runs per d-row of [H, 128, Skp].
// Rows >= M are not stored; the zero padding of V^T beyond S_out is the caller's (the workspace is allocated zeroed).  A stream
// whose row range is not 8-aligned takes an element-wise (still coalesced) V^T store; q / k rows have no alignment to keep.
// MT = 16-row m-tiles per wave, WROWS = rows per M-half of the block: <8, 128> on the 256 x 256 tiling, <12, 192> on the 384 x 256 one
// (round 5), whose V^T tile (384 x 256 bf16 = 192 KiB) does not fit the LDS and goes out in two passes, one per M-half.
// timing-only ablations of the q / k epilogue (side builds -DAPEXMI_QK_ABL=mask, results WRONG by construction; tools/gemm_qk_ablate.sh):
// 1 no rotary-table loads | 2 no output store | 4 no sum-of-squares pass | 8 no norm factor | 16 no sched_barrier | 32 no main loop
#ifndef APEXMI_QK_ABL
#define APEXMI_QK_ABL 0
#endif
template <int MT = 8, int WROWS = 128>
APEXMI_DEVICE void qkv_epilogue16(f32x4_t (&acc16)[4][MT], const GemmProblem& P, const QkvShared& Q, int M, int m0, int n0,
                                  int wave, int wm, int wn, int lane, char* smem) {
    const int g = lane >> 4, c = lane & 15;
    const int which = n0 / Q.inner;                    // 0 q, 1 k, 2 v: block-uniform
    const int ncol0 = n0 - which * Q.inner;            // first column of the tile inside q / k / v
    // bf16(acc + bias) of (slab p, m-tile mt): 8 consecutive columns per lane.  Recomputed where it is needed instead of kept
    // (64 registers on top of the 128 accumulators would spill)
    float bs[2][4];                                    // bias of the slab being processed (reloaded per slab: registers)
    auto load_bias = [&](int p) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            u32x2 b = {0u, 0u};
            if (P.bias != nullptr) b = *(const u32x2*)(P.bias + n0 + wn * 64 + p * 32 + 16 * t + 4 * g);
            bs[t][0] = bf16_lo(b[0]);
            bs[t][1] = bf16_hi(b[0]);
            bs[t][2] = bf16_lo(b[1]);
            bs[t][3] = bf16_hi(b[1]);
        }
    };
    auto rounded = [&](int p, int mt) -> u32x4 {
        const f32x4_t& x = acc16[2 * p][mt];
        const f32x4_t& y = acc16[2 * p + 1][mt];
        uint32_t x0 = pack_bf16(x[0] + bs[0][0], x[1] + bs[0][1]), x1 = pack_bf16(x[2] + bs[0][2], x[3] + bs[0][3]);
        uint32_t y0 = pack_bf16(y[0] + bs[1][0], y[1] + bs[1][1]), y1 = pack_bf16(y[2] + bs[1][2], y[3] + bs[1][3]);
        swap16(x0, y0);
        swap16(x1, y1);
        return u32x4{x0, x1, y0, y1};
    };
    const int cw = 16 * (g & 1) + 8 * (g >> 1);        // first of the lane's 8 columns inside a 32-column slab
    __syncthreads();                                    // every wave is out of the main loop's fragment reads
    if (which == 2) {
        // ---- V^T: tile[r][d] (ROWS x 256 bf16), row r rotated by ((r >> 3) + 4 (r & 7)) sixteen-byte chunks
        constexpr int NPASS = (2 * WROWS * 512 > 160 * 1024) ? 2 : 1;         // M-halves staged together, or one after the other
        constexpr int ROWS = NPASS == 1 ? 2 * WROWS : WROWS;
        bf16_t* tile = (bf16_t*)smem;
        const int tid = wave * 64 + lane;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            if (NPASS == 2 && pass == 1) __syncthreads();                      // the first half's readers are done with the tile
            if (NPASS == 1 || wm == pass) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    load_bias(p);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int r = (NPASS == 1 ? wm * WROWS : 0) + mt * 16 + c;
                        const int dch = (wn * 64 + p * 32 + cw) >> 3;
                        *(u32x4*)(tile + r * 256 + (((dch + (r >> 3) + 4 * (r & 7)) & 31) << 3)) = rounded(p, mt);
                    }
                }
            }
            __syncthreads();
            const int mp = m0 + (NPASS == 1 ? 0 : pass * WROWS);               // first output row of this pass
            if (((P.row0 | M) & 7) != 0) {
                // a stream that does not start (or end) on an 8-position boundary of the joint sequence (prompt lengths are what they
                // are): 16-byte stores would be misaligned, so one element per lane, a wave = 64 consecutive positions of one d-row
                // (a contiguous 128-byte run) — 8 x the store instructions of the aligned form, the same bytes
                for (int i = 0; i < ROWS / 2; ++i) {
                    const int idx = i * 512 + tid;
                    const int r = idx % ROWS, d = idx / ROWS;
                    const bf16_t e = tile[r * 256 + ((((d >> 3) + (r >> 3) + 4 * (r & 7)) & 31) << 3) + (d & 7)];
                    const int h = (ncol0 + d) >> 7, dd = (ncol0 + d) & 127;
                    if (mp + r < M) Q.vt[((int64_t)h * 128 + dd) * Q.Skp + P.row0 + mp + r] = e;
                }
                continue;
            }
#pragma unroll 4
            for (int i = 0; i < ROWS / 16; ++i) {
                const int idx = i * 512 + tid;
                const int sc = idx & 7, d = (idx >> 3) & 255, sc_hi = idx >> 11;     // 8 lanes = 8 consecutive position chunks of one d
                const int s8 = (sc_hi * 8 + sc) * 8;                                // first of the 8 positions (tile row)
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = s8 + j;
                    const uint32_t e = tile[r * 256 + ((((d >> 3) + (r >> 3) + 4 * (r & 7)) & 31) << 3) + (d & 7)];
                    if (j & 1) w[j >> 1] |= e << 16;
                    else w[j >> 1] = e;
                }
                const int h = (ncol0 + d) >> 7, dd = (ncol0 + d) & 127;
                if (mp + s8 < M)
                    *(u32x4*)(Q.vt + ((int64_t)h * 128 + dd) * Q.Skp + P.row0 + mp + s8) = u32x4{w[0], w[1], w[2], w[3]};
            }
        }
        return;
    }
    // ---- q / k: RMS norm over the head, rotation, [H, S_out, 128]
    float* red = (float*)smem;                          // [8 waves][2 MT = slab x m-tile][64 lanes]
#pragma unroll
    for (int p = 0; p < ((APEXMI_QK_ABL & 4) ? 0 : 2); ++p) {
        load_bias(p);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float x[8];
            unpack8(rounded(p, mt), x);
            float sq = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sq += x[j] * x[j];
            red[(wave * 2 * MT + p * MT + mt) * 64 + lane] = sq;
        }
    }
    __syncthreads();
    const bf16_t* nw = which ? P.nk : P.nq;
    const int h = (ncol0 >> 7) + (wn >> 1);
    bf16_t* dst_base = (which ? Q.ko : Q.qo) + (int64_t)h * Q.S_out * 128;
    auto rinv_of = [&](int mt) -> float {
        const float a0 = red[(wave * 2 * MT + mt) * 64 + lane] + red[((wave ^ 1) * 2 * MT + mt) * 64 + lane];          // chunk ^ 8: the partner wave
        const float a1 = red[(wave * 2 * MT + MT + mt) * 64 + lane] + red[((wave ^ 1) * 2 * MT + MT + mt) * 64 + lane];
        float sq = a0 + a1;                                                              // chunk ^ 4: the other slab
        // chunk ^ 2 and chunk ^ 1: lane ^ 16 and lane ^ 32 through v_permlane16_swap / v_permlane32_swap — one instruction each where
        // __shfl_xor is a ds_bpermute with its index arithmetic and an LDS round trip (round 6: the 384 x 256 form calls this 24 times
        // per tile and wave); x + partner is commutative, so the bits are those of `sq += __shfl_xor(sq, ...)`
        sq = sum_xor16(sq);
        sq = sum_xor32(sq);
        return rsqrtf(sq * (1.0f / 128) + Q.eps);
    };
    // 256 x 256 tiling: the 8 factors once, in registers; 384 x 256 (192 accumulators): recomputed per (slab, m-tile) from the LDS sums
    float rinv[MT > 8 ? 1 : MT];
    if constexpr (MT <= 8) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) rinv[mt] = rinv_of(mt);
    }
    if constexpr (!(APEXMI_QK_ABL & 64)) {
        if (Q.rope_pairs != nullptr) {
            // ---- compact table (round 6): the rows of the rotary table travel through a per-wave LDS ring of NS slots of 2 KiB
            // (cos pairs | sin pairs, 16 bytes per lane each), requested NS - 1 iterations ahead with register-free LDS-DMA.
            // Why: the loads of this epilogue queue behind the K-loop staging traffic of the other 255 CUs, which saturates the
            // L2 -> CU path; with one (or two) rows in flight every iteration of a wave waits that queue out (a q / k tile of
            // 384 x 256: 31 us against 9-10 us for the V^T and GELU tiles; timing-only ablation: without the table loads the fused
            // launch is as fast as the un-fused one).  Neither halving the bytes nor prefetching two iterations ahead moved it;
            // four ahead does: +45..57 us -> +13 us on the 470 us single-block launch of Flux (profiles/r06_gemm_x384_epilogue_trace.log).
            // The consumer waits on a counted vmcnt — vector-memory operations complete in issue order under one counter, so "at
            // most 2 x (later iterations' DMAs)" outstanding means this iteration's two pieces have landed; the output stores
            // issued in between only make the wait stronger — and reads the pieces back with ds_read_b128.  Same values into the
            // same arithmetic in the same order: bit-identical to the direct loads below. ----
            constexpr int NIT = 2 * MT, NS = 5;      // 384 x 256: 48 KiB sums + 8 x 10 KiB ring + 32 KiB weights = the 160 KiB of the CU
            constexpr int RED_BYTES = 8 * 2 * MT * 64 * 4;
            char* ring = smem + RED_BYTES + wave * (NS * 2048);
            const unsigned ring_lane = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring + (unsigned)lane * 16u;
            auto issue = [&](int it) {
                const int p = it / MT, mt = it % MT;
                const int d = (wn & 1) * 64 + p * 32 + cw;
                int m = m0 + wm * WROWS + mt * 16 + c;
                asm volatile("" : "+v"(m));        // opaque: the two slabs share their rows, and row pointers kept across them spill
                const int srow = P.row0 + min(m, M - 1);
                const float* cp = Q.rope_pairs + (int64_t)srow * 64 + (d >> 1);
                char* dst = ring + (it % NS) * 2048;
                glds16(cp, dst);
                glds16(cp + (int64_t)Q.S_out * 64, dst + 1024);
            };
#pragma unroll
            for (int it = 0; it < NS - 1; ++it) issue(it);
            // 384 x 256: the norm weights of the lane's two 8-column groups wait in the wave's own LDS corner (8 registers fewer
            // across the slab: with them live the compiler spills a value per iteration, and a scratch reload is a vmcnt(0) — which
            // would wait for every prefetched piece); 256 x 256: registers
            float* wtab = (float*)(smem + RED_BYTES + 8 * NS * 2048) + wave * 1024 + lane * 8;
            float wv[MT > 8 ? 1 : 8];
            if constexpr (MT > 8) {
                if (nw != nullptr) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        float w8[8];
                        unpack8(*(const u32x4*)(nw + (wn & 1) * 64 + p * 32 + cw), w8);
                        *(f32x4*)(wtab + p * 512) = f32x4{w8[0], w8[1], w8[2], w8[3]};
                        *(f32x4*)(wtab + p * 512 + 4) = f32x4{w8[4], w8[5], w8[6], w8[7]};
                    }
                }
            }
            auto body = [&](auto IT) {
                constexpr int it = decltype(IT)::value;
                if constexpr (it < NIT) {
                    constexpr int p = it / MT, mt = it % MT;
                    constexpr int later = (NIT - 1 - it) < (NS - 1) ? (NIT - 1 - it) : (NS - 1);     // DMA iterations issued behind this one
                    if constexpr (it + NS - 1 < NIT) issue(it + NS - 1);
                    const int d = (wn & 1) * 64 + p * 32 + cw;
                    if constexpr (mt == 0) {
                        load_bias(p);
                        if constexpr (MT <= 8) {
                            if (nw != nullptr) unpack8(*(const u32x4*)(nw + d), wv);
                        }
                    }
                    float x[8], y[8];
                    unpack8(rounded(p, mt), x);
                    if (nw != nullptr) {
                        if constexpr (MT > 8) {
                            const float ri = rinv_of(mt);
                            const f32x4 w0 = *(const f32x4*)(wtab + p * 512), w1 = *(const f32x4*)(wtab + p * 512 + 4);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                x[j] = x[j] * ri * w0[j];
                                x[j + 4] = x[j + 4] * ri * w1[j];
                            }
                        } else {
                            const float ri = rinv[MT > 8 ? 0 : mt];
#pragma unroll
                            for (int j = 0; j < 8; ++j) x[j] = x[j] * ri * wv[MT > 8 ? 0 : j];
                        }
                    }
                    f32x4 cq, sq4;
                    asm volatile("s_waitcnt vmcnt(%3)\n\t"
                                 "ds_read_b128 %0, %2 offset:%4\n\t"
                                 "ds_read_b128 %1, %2 offset:%5\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(cq), "=&v"(sq4)
                                 : "v"(ring_lane), "n"(2 * later), "n"((it % NS) * 2048), "n"((it % NS) * 2048 + 1024)
                                 : "memory");
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        y[2 * q] = fmaf(x[2 * q], cq[q], -(x[2 * q + 1] * sq4[q]));
                        y[2 * q + 1] = fmaf(x[2 * q + 1], cq[q], x[2 * q] * sq4[q]);
                    }
                    int m = m0 + wm * WROWS + mt * 16 + c;
                    asm volatile("" : "+v"(m));        // the store address is made HERE, behind the table read (hoisted above it, it spills)
                    const int srow = P.row0 + min(m, M - 1);
                    if (m < M)
                        *(u32x4*)(dst_base + (int64_t)srow * 128 + d) =
                            u32x4{pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]), pack_bf16(y[4], y[5]), pack_bf16(y[6], y[7])};
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
#define APEXMI_QK_IT(i) body(std::integral_constant<int, (i)>{});
            APEXMI_QK_IT(0) APEXMI_QK_IT(1) APEXMI_QK_IT(2) APEXMI_QK_IT(3) APEXMI_QK_IT(4) APEXMI_QK_IT(5)
            APEXMI_QK_IT(6) APEXMI_QK_IT(7) APEXMI_QK_IT(8) APEXMI_QK_IT(9) APEXMI_QK_IT(10) APEXMI_QK_IT(11)
            APEXMI_QK_IT(12) APEXMI_QK_IT(13) APEXMI_QK_IT(14) APEXMI_QK_IT(15) APEXMI_QK_IT(16) APEXMI_QK_IT(17)
            APEXMI_QK_IT(18) APEXMI_QK_IT(19) APEXMI_QK_IT(20) APEXMI_QK_IT(21) APEXMI_QK_IT(22) APEXMI_QK_IT(23)
#undef APEXMI_QK_IT
            static_assert(NIT <= 24, "the unrolled call list above covers 24 iterations");
            static_assert(MT > 8 ? RED_BYTES + 8 * NS * 2048 + 8 * 4096 <= 160 * 1024 : RED_BYTES + 8 * NS * 2048 <= 128 * 1024,
                          "sums + ring (+ weights) must fit the tile's LDS");
            return;
        }
    }
#pragma unroll
    for (int p = 0; p < ((APEXMI_QK_ABL & 32) ? 0 : 2); ++p) {
        const int d = (wn & 1) * 64 + p * 32 + cw;      // column inside the head
        load_bias(p);
        float wv[8];
        if (nw != nullptr) unpack8(*(const u32x4*)(nw + d), wv);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + wm * WROWS + mt * 16 + c;
            const int srow = P.row0 + min(m, M - 1);
            float x[8], y[8];
            unpack8(rounded(p, mt), x);
            if (nw != nullptr) {
                const float ri = (APEXMI_QK_ABL & 8) ? 1.0f : MT > 8 ? rinv_of(mt) : rinv[MT > 8 ? 0 : mt];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = x[j] * ri * wv[j];
            }
            const float* cp = Q.rope + (int64_t)srow * 128 + d;
            const float* sp = Q.rope + (int64_t)Q.S_out * 128 + (int64_t)srow * 128 + d;
            float cs[8], sn[8];
            if constexpr (APEXMI_QK_ABL & 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) cs[j] = 1.f, sn[j] = 0.5f;
            } else {
                const f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4);
                const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cs[j] = c0[j];
                    cs[j + 4] = c1[j];
                    sn[j] = s0[j];
                    sn[j + 4] = s1[j];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                y[2 * q] = fmaf(x[2 * q], cs[2 * q], -(x[2 * q + 1] * sn[2 * q]));
                y[2 * q + 1] = fmaf(x[2 * q + 1], cs[2 * q + 1], x[2 * q] * sn[2 * q + 1]);
            }
            if ((APEXMI_QK_ABL & 2) ? (y[0] + y[3] + y[5] == 12345.678f) : (m < M))
                *(u32x4*)(dst_base + (int64_t)srow * 128 + d) =
                    u32x4{pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]), pack_bf16(y[4], y[5]), pack_bf16(y[6], y[7])};
            // rows' table loads in flight at a time: two on the 256 x 256 tiling, ONE where 192 accumulators leave 64 registers
            if (!(APEXMI_QK_ABL & 16) && (MT > 8 || (mt & 1))) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

#if APEXMI_GEMM_TRACE
#define APEXMI_TRACE_END()                                                                                   \
    do {                                                                                                     \
        if (G.trace != nullptr) {                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                               \
            const unsigned long long t_st = __builtin_amdgcn_s_memrealtime();                                \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                 \
            const unsigned long long t_ack = __builtin_amdgcn_s_memrealtime();                               \
            if (tid == 0) {                                                                                  \
                unsigned long long* o = G.trace + (size_t)blockIdx.x * 8;                                    \
                o[0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  /* HW_REG_HW_ID */             \
                o[1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20); /* HW_REG_XCC_ID */            \
                o[2] = tr_in;                                                                                \
                o[3] = tr_l0;                                                                                \
                o[4] = tr_l1;                                                                                \
                o[5] = t_st;                                                                                 \
                o[6] = t_ack;                                                                                \
                o[7] = (unsigned long long)s;                                                                \
            }                                                                                                \
        }                                                                                                    \
    } while (0)
#else
#define APEXMI_TRACE_END() do { } while (0)
#endif

template <typename CFG, int EPI>
__device__ __forceinline__ void gemm_tile(const GemmGroup& G, char* smem, int s, const SkCtx& sk) {
    constexpr int BM = CFG::BM, BN = CFG::BN, TM = CFG::TM, TN = CFG::TN;

#if APEXMI_GEMM_TRACE
    unsigned long long tr_in = 0, tr_l0 = 0, tr_l1 = 0;
    if (G.trace != nullptr) tr_in = __builtin_amdgcn_s_memrealtime();
#endif
    int tid_ = threadIdx.x;
    // opaque per call: the persistent (stream-K) launch calls this in a loop, and everything derived from the lane id in the
    // epilogue would otherwise be hoisted out of that loop and held across the K-loop (+40 VGPRs: 160-220 spilled dwords)
    asm volatile("" : "+v"(tid_));
    const int tid = tid_;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / CFG::WN, wn = wave % CFG::WN;
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- tile id -> problem, (pm, pn): XCD-contiguous, grouped GROUP_M tall ----
    int gi = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUPS; ++i)
        if (i < G.count && s >= G.p[i].tile0) gi = i;
    GemmProblem P = G.p[gi];
    if (G.batch > 1) {   // uniform: one batch element per blockIdx.y (per-head attention GEMMs of the text encoders)
        const int64_t z = blockIdx.y;
        P.A += z * G.bsA;
        P.W += z * G.bsW;
        P.C = (bf16_t*)((char*)P.C + z * G.bsC_bytes);
    }
    s -= P.tile0;
    const int nn = P.nn, N = P.N, M = P.M;
    const int GM = G.group_m;
    const int width = GM * nn;
    const int first_m = (s / width) * GM;
    const int gsz = min(P.nm - first_m, GM);
    const int pm = first_m + (s % width) % gsz;
    const int pn = (s % width) / gsz;
    const int m0 = pm * BM, n0 = pn * BN;

    // ---- per-lane staging sources (rows clamped at the edges; OOB rows are never stored) ----
    const char* a_src[CFG::A_LD];
    const char* w_src[CFG::W_LD];
#pragma unroll
    for (int i = 0; i < CFG::A_LD; ++i) {
        const int p = (i * CFG::NW + wave) * 64 + lane;  // 16-byte chunk index inside the tile image
        const int row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
        a_src[i] = (const char*)(P.A + (int64_t)min(m0 + row, M - 1) * P.lda + c * 8);
    }
#pragma unroll
    for (int i = 0; i < CFG::W_LD; ++i) {
        const int p = (i * CFG::NW + wave) * 64 + lane;
        const int row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
        w_src[i] = (const char*)(P.W + (int64_t)min(n0 + row, N - 1) * P.ldw + c * 8);
    }

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x4_t acc16[4][8];  // SCHED 5 / 7 only: [16-column n-tile][16-row m-tile]
    if constexpr (CFG::SCHED == 5 || CFG::SCHED >= 7) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc16[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                // opaque: inside the persistent (stream-K) item loop the allocator otherwise fails to keep the accumulators in place
                // across the K-loop's back edge and copies all 128 registers once per K-tile (64 v_mov_b64: +50 % time)
                asm volatile("" : "+v"(acc16[i][j]));
            }
    }
    const int nkt_all = G.K / BK;
    const int nkt = (CFG::SCHED == 5) ? sk.kt1 - sk.kt0 : nkt_all;    // K-tiles of THIS call (stream-K segments: SCHED 5 only)
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (G.clk != nullptr) {          // kernel-uniform
        clk_c0 = __builtin_readcyclecounter();
        clk_r0 = __builtin_amdgcn_s_memrealtime();
    }
#if APEXMI_GEMM_TRACE
    if (G.trace != nullptr) tr_l0 = __builtin_amdgcn_s_memrealtime();
#endif

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * CFG::STAGE + wave * 1024;
        const int64_t koff = (int64_t)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < CFG::A_LD; ++i) glds16(a_src[i] + koff, base + i * (CFG::NW * 1024));
#pragma unroll
        for (int i = 0; i < CFG::W_LD; ++i)
            glds16(w_src[i] + koff, base + CFG::A_BYTES + i * (CFG::NW * 1024));
    };

    // fragment read offsets (bytes) inside a tile image; k-step ks selects chunk (2 ks + hi) ^ sw
    int a_off[TM], a_sw[TM], w_off[TN], w_sw[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int r = wm * (BM / CFG::WM) + t * 32 + l31;
        a_off[t] = r * 128;
        a_sw[t] = (r >> 1) & 7;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int r = wn * (BN / CFG::WN) + t * 32 + l31;
        w_off[t] = r * 128;
        w_sw[t] = (r >> 1) & 7;
    }

    if constexpr (CFG::SCHED == 0) {
#if APEXMI_GEMM_PEEL >= 2 && !defined(APEXMI_GEMM_SMALL_VECTOR)   // (-DAPEXMI_GEMM_SMALL_VECTOR: the A/B arm with the builtin's vector addresses)
        // the pieces in the scalar-base form of global_load_lds (as in the SCHED 5 loop below: uniform tile base in SGPRs + constant
        // 32-bit lane offsets): these launches put at most a workgroup or two on a CU, and their K-tile is a chain of piece issues
        const uint64_t a_u64 = (uint64_t)(P.A + (int64_t)min(m0, M - 1) * P.lda), w_u64 = (uint64_t)(P.W + (int64_t)min(n0, N - 1) * P.ldw);
        const char* const a_u = (const char*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a_u64 >> 32)) << 32) |
                                              (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a_u64));
        const char* const w_u = (const char*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(w_u64 >> 32)) << 32) |
                                              (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)w_u64));
        uint32_t a_o[CFG::A_LD], w_o[CFG::W_LD];
#pragma unroll
        for (int i = 0; i < CFG::A_LD; ++i) a_o[i] = (uint32_t)(a_src[i] - a_u);
#pragma unroll
        for (int i = 0; i < CFG::W_LD; ++i) w_o[i] = (uint32_t)(w_src[i] - w_u);
        const uint32_t smem_lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
        auto piece = [&](const char* sbase, uint32_t voff, uint32_t la) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"(voff), "s"(sbase), "s"(la) : "memory", "m0");
        };
#pragma clang diagnostic pop
        auto stage = [&](int buf, int kt) {
            const uint32_t base = smem_lds0 + buf * CFG::STAGE + wave * 1024;
            const char* ga = a_u + (int64_t)kt * (BK * 2);
            const char* gw = w_u + (int64_t)kt * (BK * 2);
#pragma unroll
            for (int i = 0; i < CFG::A_LD; ++i) piece(ga, a_o[i], base + i * (CFG::NW * 1024));
#pragma unroll
            for (int i = 0; i < CFG::W_LD; ++i) piece(gw, w_o[i], base + CFG::A_BYTES + i * (CFG::NW * 1024));
        };
#endif
        stage(0, 0);
        for (int kt = 0; kt < nkt; ++kt) {
            // tile kt's LDS-DMA landed (explicit: hipcc's __syncthreads() does not reliably wait for
            // LDS-DMA, see attention.hip) and the other buffer is free
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
            const char* As = smem + (kt & 1) * CFG::STAGE;
            const char* Ws = As + CFG::A_BYTES;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = ks * 2 + hi;
                bf16x8 af[TM], wf[TN];
#pragma unroll
                for (int t = 0; t < TM; ++t) af[t] = *(const bf16x8*)(As + a_off[t] + ((c ^ a_sw[t]) << 4));
#pragma unroll
                for (int t = 0; t < TN; ++t) wf[t] = *(const bf16x8*)(Ws + w_off[t] + ((c ^ w_sw[t]) << 4));
#pragma unroll
                for (int nt = 0; nt < TN; ++nt)
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
            }
        }
    } else if constexpr (CFG::SCHED == 4) {
        // ---- one wave per SIMD: 4 waves (2x2) of 128x128, accumulators in AGPRs ----
        // Measured on gfx950 (tools/ubench/mfma_vmem.hip): at the GEMM's rate of one LDS-DMA piece per
        // 4 MFMAs, two waves per SIMD lose 28-36 % of the matrix pipe to the piece issue, one wave per
        // SIMD loses 1 % (buffer_load .. lds) to 5 % (global_load_lds).  So: 256 threads, every wave
        // alone on its SIMD, the whole pipeline in one instruction stream.  Same rotated period as
        // SCHED 2/3 (barrier between k-steps 2 and 3), 64 MFMA slots per period; slot s carries MFMA s,
        // one of the 8 fragment reads of the next k-step on the first 8 slots of a k-step, and one of the
        // 16 LDS-DMA pieces of tile kt+2 on every third slot from ROT.  Pieces go through buffer_load
        // with the row/swizzle part in one VGPR per operand and the piece/K offset in an SGPR; rows past
        // M fall outside the descriptor's range and read as zero, so there is no clamping.
        static_assert(CFG::SCHED != 4 || (TM == 4 && TN == 4 && CFG::NW == 4), "SCHED 4 is written for 2x2 waves of 128x128");
        bf16x8 fa[2][4], fw[2][4];
        const int rows_a = min(M - m0, BM), rows_w = min(N - n0, BN);
        auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A + (int64_t)m0 * P.lda), 0,
                                                        (int)(((int64_t)(rows_a - 1) * P.lda + G.K) * 2), 0x00020000);
        auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)(P.W + (int64_t)n0 * P.ldw), 0,
                                                        (int)(((int64_t)(rows_w - 1) * P.ldw + G.K) * 2), 0x00020000);
        // piece i of a wave = rows (i*4 + wave)*8 .. +7; (row >> 1) & 7 = 4 (wave & 1) + (lane >> 4) for every i
        const int sw_src = ((lane & 7) ^ (4 * (wave & 1) + (lane >> 4))) * 16;
        const int voff_a = (lane >> 3) * (int)P.lda * 2 + sw_src;
        const int voff_w = (lane >> 3) * (int)P.ldw * 2 + sw_src;
        const int sa = (int)P.lda * 2 * 8, sw = (int)P.ldw * 2 * 8;  // bytes per 8-row piece step
        // When the row stride is a multiple of 8 KiB (K = 12288 or 8192) every row of every tile at one k
        // offset falls into the same cache set / memory channel; CUs marching through K in lockstep then
        // queue on it (945 vs 1220 TFLOP/s at 4096x3072x12288 against the same problem with the stride
        // padded by 128 B).  Starting each XCD's tiles an eighth of K apart spreads them (+4 %) and keeps
        // the operand sharing inside an XCD's L2; the fp32 sum is order-independent up to rounding.
        const int krot = (G.K * 2) % 8192 == 0 ? (int)(blockIdx.x & 7) * (nkt >> 3) : 0;
        auto dma1 = [&](int slot, int kt_, int i) {  // i in 0..15: 0..7 activation pieces, 8..15 weight pieces
            int kt = kt_ + krot;
            kt = kt >= nkt ? kt - nkt : kt;
            char* base = smem + slot * CFG::STAGE + wave * 1024;
            if (i < 8)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(base + i * 4096), 16, voff_a,
                                                         (i * 4 + wave) * sa + kt * (BK * 2), 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(base + CFG::A_BYTES + (i - 8) * 4096), 16,
                                                         voff_w, ((i - 8) * 4 + wave) * sw + kt * (BK * 2), 0, 0);
        };
        auto rd1 = [&](int slot, int ks, int b, int j) {  // j-th of the eight fragment reads of a k-step
            const char* As = smem + slot * CFG::STAGE;
            const char* Ws = As + CFG::A_BYTES;
            const int c = ks * 2 + hi;
            // order W0 A0 A1 A2 A3 W1 W2 W3: the first MFMAs of the next k-step need the fewest reads
            if (j == 0) fw[b][0] = *(const bf16x8*)(Ws + w_off[0] + ((c ^ w_sw[0]) << 4));
            else if (j <= 4) fa[b][j - 1] = *(const bf16x8*)(As + a_off[j - 1] + ((c ^ a_sw[j - 1]) << 4));
            else fw[b][j - 4] = *(const bf16x8*)(Ws + w_off[j - 4] + ((c ^ w_sw[j - 4]) << 4));
        };
        auto mma1 = [&](int b, int i) {  // i-th MFMA of a k-step: (nt, mt) = (i / 4, i % 4)
            acc[i >> 2][i & 3] =
                __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[b][i >> 2], fa[b][i & 3], acc[i >> 2][i & 3], 0, 0, 0);
        };
#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)
        constexpr int ROT = 0;
        constexpr int STEP = 2;  // MFMA slots between LDS-DMA pieces (3: -3 %, 1: -4 % on the Flux shapes)
        auto period = [&](int kt, auto has_dma, auto has_next) {
            constexpr bool DMA = decltype(has_dma)::value, NEXT = decltype(has_next)::value;
            const int cur = kt & 1, nxt = cur ^ 1;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            W4_FENCE();
            __builtin_amdgcn_s_barrier();
            W4_FENCE();
#pragma unroll
            for (int s = 0; s < (NEXT ? 64 : 16); ++s) {
                const int q = s >> 4, i = s & 15;
                mma1((q + 1) & 1, i);
                if (NEXT && i < 8) rd1(nxt, q, q & 1, i);
                if (DMA && s >= ROT && (s - ROT) % STEP == 0 && (s - ROT) / STEP < 16) dma1(cur, kt + 2, (s - ROT) / STEP);
                W4_FENCE();
            }
        };
#pragma unroll
        for (int i = 0; i < 16; ++i) dma1(0, 0, i);
        if (nkt > 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) dma1(1, 1, i);
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        W4_FENCE();
        __builtin_amdgcn_s_barrier();
        W4_FENCE();
#pragma unroll
        for (int j = 0; j < 8; ++j) rd1(0, 0, 0, j);
        W4_FENCE();
#pragma unroll
        for (int s = 0; s < 48; ++s) {  // k-steps 0..2 of tile 0
            const int q = s >> 4, i = s & 15;
            mma1(q & 1, i);
            if (i < 8) rd1(0, q + 1, (q + 1) & 1, i);
            W4_FENCE();
        }
        int kt = 0;
        for (; kt + 2 < nkt; ++kt) period(kt, std::true_type{}, std::true_type{});
        if (kt + 1 < nkt) {
            period(kt, std::false_type{}, std::true_type{});
            ++kt;
        }
        period(kt, std::false_type{}, std::false_type{});
#undef W4_FENCE
    } else if constexpr (CFG::SCHED >= 7) {
        // ---- free-running ring on v_mfma_f32_16x16x32_bf16 (round 4) ----
        // tools/ubench/gemm_roof.hip: the tile's own instruction mix as a stall-free stream (64 MFMA + 24 fragment reads + 8 LDS-DMA
        // pieces per wave and 64-deep K-tile, fragments read one phase ahead into a second register set, no barrier) keeps the
        // matrix pipe 93 % busy and is POWER-bound at 1.57 PF (clock 1.62 GHz), where the ping-pong schedule runs at 2.05 GHz with
        // the pipe 50 % busy: its eight barrier intervals per K-tile each end in an exposed fragment-read latency.  This schedule
        // is that stream plus the synchronisation a real tile needs and nothing else:
        //   * the K-tile is cut into 32-deep SUB-TILES (one MFMA k-step); LDS is a ring of four 32 KiB slots
        //     [A 256 rows x 64 B | W 256 rows x 64 B], chunk swizzle c ^ ((row >> 2) & 3) (64-byte pitch: four rows per bank row;
        //     a 16-row fragment read covers all 64 banks exactly once per 16-lane group);
        //   * step s: fragment reads of sub-tile s + 1 into the OTHER register set and the wave's four LDS-DMA pieces of sub-tile
        //     s + 3 (s + 4 with five slots) in one barrier interval, the 32 MFMAs of sub-tile s in the next: four barriers per 64-deep
        //     K-tile instead of eight, and nothing waits for a read it has just issued (RAW / WAR argument at RING_STEP below).
        // Same k order per accumulator as SCHED 5 (32-deep k-steps ascending): results are bit-identical to it.
        const int l15 = lane & 15, g4 = lane >> 4;
        const int swz = (g4 ^ (l15 >> 2)) << 4;
        const int rd_a = (wm * 128 + l15) * 64 + swz;          // + t * 1024 (16-row tile) + slot * 32768
        const int rd_w = 16384 + (wn * 64 + l15) * 64 + swz;   // + u * 1024
        // LDS-DMA sources: a wave-uniform base per operand (SGPR pair, advanced by the k offset) + a 32-bit per-lane offset, so
        // that a piece costs one VGPR, not a 64-bit pointer (the two fragment register sets leave no room for those)
        const char* base_a = (const char*)(P.A + (int64_t)m0 * P.lda);
        const char* base_w = (const char*)(P.W + (int64_t)n0 * P.ldw);
        unsigned off[4];
        int dst[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pc = wave + 8 * j;                        // 1 KiB piece = 16 rows x 64 B
            const int row = pc * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((row >> 2) & 3);
            off[j] = (unsigned)((min(m0 + row, M - 1) - m0) * (int)P.lda * 2 + c * 16);
            off[2 + j] = (unsigned)((min(n0 + row, N - 1) - n0) * (int)P.ldw * 2 + c * 16);
            dst[j] = pc * 1024;
            dst[2 + j] = 16384 + pc * 1024;
        }
        constexpr int RD_ = CFG::SCHED == 8 ? 4 : 3;           // pieces go out RD_ sub-tiles ahead into a ring of RD_ + 1 slots
        constexpr int NSLOT = RD_ + 1;
        auto dma = [&](int st, int slot) {
            char* base = smem + slot * 32768;
            const char* ka = base_a + (int64_t)st * 64;
            const char* kw = base_w + (int64_t)st * 64;
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(ka + off[j], base + dst[j]);
#pragma unroll
            for (int j = 2; j < 4; ++j) glds16(kw + off[j], base + dst[j]);
        };
        // Fragment reads as inline asm: beside an LDS-DMA in flight the compiler's wait-count pass guards every LDS consumer with
        // lgkmcnt(0) (a flat-class instruction with an LDS side is "pending flat" to it), which here would wait for the twelve reads
        // just issued for the NEXT step.  The asm reads are invisible to that pass; `frag_wait<N>` is the counted wait, and it names
        // every register of the set it releases as read-write, so no consumer can be scheduled above it (§5.7 form (ii)).
        bf16x8 fa[2][8], fw[2][4];
#define RING_DSR(dstreg, addr, offs) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dstreg) : "v"(addr), "i"(offs) : "memory")
        // (macros, not generic lambdas: clang rejects an asm operand that names a captured array inside a generic lambda)
#define RING_RD(slotidx, set)                              \
    do {                                                   \
        const int slot_ = (slotidx) * 32768;               \
        const int aw_ = slot_ + rd_w, aa_ = slot_ + rd_a;  \
        RING_DSR(fw[set][0], aw_, 0);                      \
        RING_DSR(fw[set][1], aw_, 1024);                   \
        RING_DSR(fw[set][2], aw_, 2048);                   \
        RING_DSR(fw[set][3], aw_, 3072);                   \
        RING_DSR(fa[set][0], aa_, 0);                      \
        RING_DSR(fa[set][1], aa_, 1024);                   \
        RING_DSR(fa[set][2], aa_, 2048);                   \
        RING_DSR(fa[set][3], aa_, 3072);                   \
        RING_DSR(fa[set][4], aa_, 4096);                   \
        RING_DSR(fa[set][5], aa_, 5120);                   \
        RING_DSR(fa[set][6], aa_, 6144);                   \
        RING_DSR(fa[set][7], aa_, 7168);                   \
    } while (0)
#define RING_WAIT(n, set)                                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                                 \
                 : "+v"(fw[set][0]), "+v"(fw[set][1]), "+v"(fw[set][2]), "+v"(fw[set][3]), "+v"(fa[set][0]), "+v"(fa[set][1]), \
                   "+v"(fa[set][2]), "+v"(fa[set][3]), "+v"(fa[set][4]), "+v"(fa[set][5]), "+v"(fa[set][6]), "+v"(fa[set][7]) \
                 :                                                                                                           \
                 : "memory")
#define RING_MMA(set)                                                                                                      \
    do {                                                                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                                     \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) _Pragma("unroll") for (int t = 0; t < 8; ++t) acc16[u][t] =          \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[set][u], fa[set][t], acc16[u][t], 0, 0, 0);                         \
        __builtin_amdgcn_s_setprio(0);                                                                                     \
    } while (0)
#define RING_FENCE() __builtin_amdgcn_sched_barrier(0)
        // One step: fragment reads of sub-tile st + 1 into the other set | the wave's pieces of sub-tile st + 3 | counted wait for
        // THIS step's fragments (read a step ago; the 12 newer reads fly on) | 32 MFMAs | vmcnt: own pieces of sub-tile st + 2
        // landed, the four of st + 3 stay in flight across the barrier | barrier.  RD / DMA are literals: the last steps have
        // nothing left to read / stage.
#define RING_VM(n)                                                                       \
    do {                                                                                 \
        if (!(APEXMI_GEMM_ABLATE & 2)) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); \
    } while (0)
#define RING_BAR()                                                      \
    do {                                                                \
        if (!(APEXMI_GEMM_ABLATE & 1)) __builtin_amdgcn_s_barrier();    \
    } while (0)
        // One step of a wave = two barrier intervals: ISSUE (fragment reads of sub-tile st + 1 into the other set, the wave's pieces of
        // sub-tile st + RD_, the counted vmcnt) | barrier | MMA (32 MFMAs on this step's set, then lgkmcnt(0): the reads issued in
        // ISSUE have had the whole MFMA segment to land) | barrier.  The M-halves run ONE INTERVAL APART (waves w and w + 4 share a
        // SIMD): while one wave of a SIMD is in its MFMA segment the other is in its issue segment, whose cost — an LDS-DMA piece
        // blocks the issuing wave for 60-180 cycles, measured — is then covered by the partner's MFMAs; the lock-step form of this
        // ring (profiles/r04_gemm_ring_lockstep_*.log) had both waves of a SIMD issuing at the same time and was 8-13 % slower
        // than the ping-pong schedule.  Nothing in ISSUE waits for what it has just issued.
        //   RAW: sub-tile st + 2 is first read in interval 2 st + 2 (M-half 0's ISSUE of step st + 1); every wave has waited for
        //        its pieces of it at the end of its own ISSUE of step st (intervals 2 st / 2 st + 1), before a barrier.
        //   WAR: pieces of sub-tile st + RD_ overwrite sub-tile st - 1, whose last reads (M-half 1, ISSUE of step st - 2, interval
        //        2 st - 3) were retired by the lgkmcnt(0) that ends that wave's MMA of step st - 2 (interval 2 st - 2).
#define RING_STEP(st_, set, RD, DMA)                                        \
    do {                                                                    \
        RING_FENCE();                                                       \
        if (RD && !(APEXMI_GEMM_ABLATE & 8)) RING_RD(s_rd, (set) ^ 1);      \
        RING_FENCE();                                                       \
        if (DMA && !(APEXMI_GEMM_ABLATE & 4)) dma((st_) + RD_, s_dma);      \
        RING_FENCE();                                                       \
        if (DMA) {                                                          \
            if (RD_ == 3) RING_VM(4);                                       \
            else RING_VM(8);                                                \
        } else {                                                            \
            RING_VM(0);                                                     \
        }                                                                   \
        RING_FENCE();                                                       \
        RING_BAR();                                                         \
        RING_FENCE();                                                       \
        RING_MMA(set);                                                      \
        if (RD) RING_WAIT(0, (set) ^ 1);                                    \
        RING_FENCE();                                                       \
        RING_BAR();                                                         \
        RING_FENCE();                                                       \
        s_rd = s_rd + 1 == NSLOT ? 0 : s_rd + 1;                            \
        s_dma = s_dma + 1 == NSLOT ? 0 : s_dma + 1;                         \
    } while (0)
        const int ns = nkt * 2;                                // sub-tiles (K % 64 == 0: always even, >= 2)
        // prologue: the first RD_ sub-tiles go out, 0 and 1 must have landed before the first reads
#pragma unroll
        for (int i = 0; i < RD_; ++i)
            if (i < ns) dma(i, i);
        if (ns >= 4 && RD_ == 4) RING_VM(8);
        else if (ns >= 3) RING_VM(4);
        else RING_VM(0);
        RING_FENCE();
        __builtin_amdgcn_s_barrier();
        RING_FENCE();
        RING_RD(0, 0);
        RING_WAIT(0, 0);
        RING_FENCE();
        if (wm == 1) RING_BAR();                               // M-half 1 runs one interval behind M-half 0
        RING_FENCE();
        int s_rd = 1, s_dma = RD_ % NSLOT;                     // slots of sub-tile st + 1 / st + RD_ at step st
        int st = 0;
        for (; st + 1 + RD_ < ns; st += 2) {                   // both steps of the pair read ahead and stage ahead: no branch inside
            RING_STEP(st, 0, true, true);
            RING_STEP(st + 1, 1, true, true);
        }
        // the last one or two pairs, straight-line (a loop with variant branches here costs the allocator 400+ spills): after the
        // branch-free loop 2 or 4 steps remain; with 4, only the three-ahead ring still has a piece to issue (sub-tile ns - 1)
        if (st + 4 <= ns) {
            if constexpr (RD_ == 3) RING_STEP(st, 0, true, true);
            else RING_STEP(st, 0, true, false);
            RING_STEP(st + 1, 1, true, false);
            st += 2;
        }
        RING_STEP(st, 0, true, false);
        RING_STEP(st + 1, 1, false, false);
        RING_FENCE();
        if (wm == 0) RING_BAR();                               // balance the barrier count of the two halves
        RING_FENCE();
#undef RING_VM
#undef RING_BAR
#undef RING_STEP
#undef RING_MMA
#undef RING_RD
#undef RING_FENCE
#undef RING_DSR
#undef RING_WAIT
    } else if constexpr (CFG::SCHED == 5) {
        // ---- ping-pong schedule on v_mfma_f32_16x16x32_bf16 ----
        // Same phases, regions and waits as SCHED 1; the quadrant (64 rows x 32 columns x K 64) is 16 MFMAs
        // of 16x16x32 instead of 8 of 32x32x16.  Measured at the chip's power limit (tools/ubench/
        // mfma_power.hip, register-resident random operands, matrix pipe 95 % busy in both cases): the
        // 16x16x32 form sustains 2.03 GHz against 1.79 GHz for 32x32x16 — the step is power-bound, so the
        // cheaper instruction is the faster one.
        const int l15 = lane & 15, g4 = lane >> 4;
        const int sw16 = (l15 >> 1) & 7;
        int a16[8], w16[4];  // byte offsets of this lane's row in each 16-row tile of the wave's sub-tiles
#pragma unroll
        for (int t = 0; t < 8; ++t) a16[t] = (wm * 128 + t * 16 + l15) * 128;
#pragma unroll
        for (int t = 0; t < 4; ++t) w16[t] = (wn * 64 + t * 16 + l15) * 128;
        const int ch16[2] = {((0 + g4) ^ sw16) << 4, ((4 + g4) ^ sw16) << 4};
        bf16x8 af[4][2], wf[2][2][2];  // wf: [32-column n-tile][16-column half][k-step], both n-tiles stay resident
        auto rd_a = [&](const char* As, int half) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) af[t][ks] = *(const bf16x8*)(As + a16[half * 4 + t] + ch16[ks]);
        };
        auto rd_w = [&](const char* Ws, int nt) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) wf[nt][u][ks] = *(const bf16x8*)(Ws + w16[nt * 2 + u] + ch16[ks]);
        };
        auto mma = [&](int half, int nt) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        // (accumulators pinned to AGPRs through inline-asm MFMAs measured +-0 in the step)
                        acc16[nt * 2 + u][half * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            wf[nt][u][ks], af[t][ks], acc16[nt * 2 + u][half * 4 + t], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
#define PP_SYNC()                                                                   \
    do {                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                          \
        __builtin_amdgcn_s_barrier();                                               \
        __builtin_amdgcn_sched_barrier(0);                                          \
    } while (0)
#define PP_BAR()                               \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)
        // staging by global_load_lds with per-lane 64-bit pointers (buffer_load .. lds with scalar piece
        // offsets measured 1.7 % slower in the Flux step: 72.3 vs 71.1 ms)
        const char* ra_src[2][2];
        const char* rw_src[2][2];
        int ra_row[2][2], rw_row[2][2];  // first row of the wave's 8-row piece j of region reg
#pragma unroll
        for (int reg = 0; reg < 2; ++reg)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = j * 8 + wave;
                const int ra0 = (q < 8 ? q * 8 : 128 + (q - 8) * 8) + reg * 64;
                const int rw0 = (q >> 2) * 64 + reg * 32 + (q & 3) * 8;
                const int ra = ra0 + (lane >> 3), rw = rw0 + (lane >> 3);
                ra_src[reg][j] = (G.wpacked & 2)
                    ? (const char*)(P.A + ((int64_t)pm * nkt_all * 256 + ra) * 64 + (((lane & 7) ^ ((ra >> 1) & 7)) * 8))
                    : (const char*)(P.A + (int64_t)min(m0 + ra, M - 1) * P.lda + (((lane & 7) ^ ((ra >> 1) & 7)) * 8));
                rw_src[reg][j] = (G.wpacked & 1)
                    ? (const char*)(P.W + ((int64_t)pn * nkt_all * 256 + rw) * 64 + (((lane & 7) ^ ((rw >> 1) & 7)) * 8))
                    : (const char*)(P.W + (int64_t)min(n0 + rw, N - 1) * P.ldw + (((lane & 7) ^ ((rw >> 1) & 7)) * 8));
                ra_row[reg][j] = ra0;
                rw_row[reg][j] = rw0;
            }
        const int64_t w_kstep = (G.wpacked & 1) ? 256 * 128 : BK * 2;     // bytes between consecutive K-tiles of a W / A row
        const int64_t a_kstep = (G.wpacked & 2) ? 256 * 128 : BK * 2;
        if (sk.kt0) {                                       // a stream-K segment starts inside the tile's K range
#pragma unroll
            for (int reg = 0; reg < 2; ++reg)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    ra_src[reg][j] += (int64_t)sk.kt0 * a_kstep;
                    rw_src[reg][j] += (int64_t)sk.kt0 * w_kstep;
                }
        }
#if APEXMI_GEMM_PEEL >= 2
        // Round 6 (shipped): a piece's address = a UNIFORM base in scalar registers (the tile's first row at the segment's first
        // K-tile, advanced per K-tile by scalar adds) + a 32-bit lane offset that never changes, issued as the scalar-base form of
        // global_load_lds — no 64-bit vector add per piece (8 per K-tile and wave before).  The lane offsets stay inside the tile's
        // 256 rows (row clamps included), so 32 bits hold them for any leading dimension below 2^22 elements.
        // (the tile's coordinates are workgroup-uniform but not always provably so for the compiler — the persistent forms read
        // them from an atomic —: v_readfirstlane pins the bases to scalar registers.)
        auto uniform_ptr = [](const char* p) {
            const uint64_t u = (uint64_t)p;
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
            return (const char*)(((uint64_t)hi << 32) | lo);
        };
        const char* const a_base_u = uniform_ptr((G.wpacked & 2) ? (const char*)(P.A + (int64_t)pm * nkt_all * 256 * 64)
                                                                 : (const char*)(P.A + (int64_t)min(m0, M - 1) * P.lda));
        const char* const w_base_u = uniform_ptr((G.wpacked & 1) ? (const char*)(P.W + (int64_t)pn * nkt_all * 256 * 64)
                                                                 : (const char*)(P.W + (int64_t)min(n0, N - 1) * P.ldw));
        uint32_t ra_off[2][2], rw_off[2][2];
#pragma unroll
        for (int reg = 0; reg < 2; ++reg)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                ra_off[reg][j] = (uint32_t)(ra_src[reg][j] - a_base_u);
                rw_off[reg][j] = (uint32_t)(rw_src[reg][j] - w_base_u);
            }
        // hipcc selects the 64-bit vector-address form for the builtin whatever the pointer looks like: the scalar-base form by hand
        // (M0 = the piece's LDS address, one state between the M0 write and the load)
        const uint32_t smem_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem);
        // M0 is on the clobber list so that hipcc's own M0 bookkeeping (it merges and hoists identical M0 initialisations of its
        // LDS-DMA builtins) sees a definition here; clang flags a reserved register on a clobber list, which is the point
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
        auto glds16_s = [&](const char* sbase, uint32_t voff, uint32_t la) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"(voff), "s"(sbase), "s"(la) : "memory", "m0");
        };
#pragma clang diagnostic pop
        auto stage_a = [&](int buf, int kt, int reg) {
            const uint32_t base = smem_lds + buf * CFG::STAGE;
            const char* gb = a_base_u + (int64_t)kt * a_kstep;          // (ra_off holds the segment's first K-tile, sk.kt0)
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16_s(gb, ra_off[reg][j], base + ra_row[reg][j] * 128);
        };
        auto stage_w = [&](int buf, int kt, int reg) {
            const uint32_t base = smem_lds + buf * CFG::STAGE + CFG::A_BYTES;
            const char* gb = w_base_u + (int64_t)kt * w_kstep;
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16_s(gb, rw_off[reg][j], base + rw_row[reg][j] * 128);
        };
#else
        auto stage_a = [&](int buf, int kt, int reg) {
            char* base = smem + buf * CFG::STAGE;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                glds16(ra_src[reg][j] + (int64_t)kt * a_kstep, base + ra_row[reg][j] * 128);
            }
        };
        auto stage_w = [&](int buf, int kt, int reg) {
            char* base = smem + buf * CFG::STAGE + CFG::A_BYTES;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                glds16(rw_src[reg][j] + (int64_t)kt * w_kstep, base + rw_row[reg][j] * 128);
            }
        };
#endif
#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
        stage_a(0, 0, 0);
        stage_w(0, 0, 0);
        stage_w(0, 0, 1);
        stage_a(0, 0, 1);
        VMCNT(0);
        PP_BAR();
        if (wm == 1) PP_BAR();
        // The last K-tile is peeled off (round 6): the steady-state body carries no `more` test — the compiled loop had five branches
        // and their scalar set-up per K-tile and wave; with the scalar-base pieces above: Flux GEMM sequence 45.7 -> 44.5 ms
        // (profiles/r06_gemm_peel_ab.log).  Phase 4 needs no fragment read: n-tile 0's weight fragments of phase 1 are still in
        // registers (-0.8 % per step against re-reading them).
        auto ktile = [&](int kt, auto more_c) {
            constexpr bool more = decltype(more_c)::value;
            const char* As = smem + (kt & 1) * CFG::STAGE;
            const char* Ws = As + CFG::A_BYTES;
            const int nb = (kt + 1) & 1;
            rd_a(As, 0);
            rd_w(Ws, 0);
            if constexpr (more) { stage_a(nb, kt + 1, 0); VMCNT(4); } else { VMCNT(2); }
            PP_SYNC();
            mma(0, 0);
            PP_BAR();
            rd_w(Ws, 1);
            if constexpr (more) { stage_w(nb, kt + 1, 0); VMCNT(4); } else { VMCNT(0); }
            PP_SYNC();
            mma(0, 1);
            PP_BAR();
            rd_a(As, 1);
            if constexpr (more) { stage_w(nb, kt + 1, 1); VMCNT(4); }
            PP_SYNC();
            mma(1, 1);
            PP_BAR();
            if constexpr (more) { stage_a(nb, kt + 1, 1); VMCNT(4); }
            PP_SYNC();
            mma(1, 0);
            PP_BAR();
        };
        for (int kt = 0; kt + 1 < nkt; ++kt) ktile(kt, std::true_type{});
        if (nkt > 0) ktile(nkt - 1, std::false_type{});
        if (wm == 0) PP_BAR();
#undef VMCNT
#undef PP_SYNC
#undef PP_BAR
    } else {
        // ---- ping-pong schedule (TM = 4, TN = 2): 4 phases per K-tile ----
        static_assert(!CFG::PP || (TM == 4 && TN == 2), "ping-pong schedule is written for 128x64 wave tiles");
        bf16x8 af[2][4], wf[4];  // [m-tile in half][k-step], [k-step]
        auto rd_a = [&](const char* As, int half) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    af[t][ks] = *(const bf16x8*)(As + a_off[half * 2 + t] + (((ks * 2 + hi) ^ a_sw[half * 2 + t]) << 4));
        };
        auto rd_w = [&](const char* Ws, int nt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                wf[ks] = *(const bf16x8*)(Ws + w_off[nt] + (((ks * 2 + hi) ^ w_sw[nt]) << 4));
        };
        auto mma = [&](int half, int nt) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[nt][half * 2 + t] =
                        __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], af[t][ks], acc[nt][half * 2 + t], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
#define PP_SYNC()                                                                   \
    do {                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                          \
        __builtin_amdgcn_s_barrier();                                               \
        __builtin_amdgcn_sched_barrier(0);                                          \
    } while (0)
#define PP_BAR()                               \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

        // Region-wise staging: the tile image is cut into the four sub-tiles the phases read
        //   RA0/RA1 = activation rows with (row & 64) == 0 / != 0   (m-half 0 / 1 of every wave)
        //   RW0/RW1 = weight rows with (row & 32) == 0 / != 0       (n-tile 0 / 1 of every wave)
        // 16 KiB each = 2 LDS-DMA per thread.  Phase 1 issues the next tile's RA0, phase 2 RW0, phase 3
        // RW1, phase 4 RA1, so at most three regions are in flight and every L segment carries 2 DMA.
        const char* ra_src[2][2];
        const char* rw_src[2][2];
        int ra_lds[2][2], rw_lds[2][2];
#pragma unroll
        for (int reg = 0; reg < 2; ++reg)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = j * 8 + wave;  // 1 KiB slot (8 rows) inside the region
                const int ra0 = (q < 8 ? q * 8 : 128 + (q - 8) * 8) + reg * 64;
                const int rw0 = (q >> 2) * 64 + reg * 32 + (q & 3) * 8;
                const int ra = ra0 + (lane >> 3), rw = rw0 + (lane >> 3);
                ra_src[reg][j] = (const char*)(P.A + (int64_t)min(m0 + ra, M - 1) * P.lda + (((lane & 7) ^ ((ra >> 1) & 7)) * 8));
                rw_src[reg][j] = (const char*)(P.W + (int64_t)min(n0 + rw, N - 1) * P.ldw + (((lane & 7) ^ ((rw >> 1) & 7)) * 8));
                ra_lds[reg][j] = ra0 * 128;
                rw_lds[reg][j] = CFG::A_BYTES + rw0 * 128;
            }
        auto stage_a = [&](int buf, int kt, int reg) {
            char* base = smem + buf * CFG::STAGE;
            const int64_t koff = (int64_t)kt * (BK * 2);
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(ra_src[reg][j] + koff, base + ra_lds[reg][j]);
        };
        auto stage_w = [&](int buf, int kt, int reg) {
            char* base = smem + buf * CFG::STAGE;
            const int64_t koff = (int64_t)kt * (BK * 2);
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(rw_src[reg][j] + koff, base + rw_lds[reg][j]);
        };
#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

        stage_a(0, 0, 0);
        stage_w(0, 0, 0);
        stage_w(0, 0, 1);
        stage_a(0, 0, 1);
        VMCNT(0);
        PP_BAR();                 // tile 0 visible to every wave
        if (wm == 1) PP_BAR();    // M-half 1 runs one barrier behind M-half 0
        for (int kt = 0; kt < nkt; ++kt) {
            const char* As = smem + (kt & 1) * CFG::STAGE;
            const char* Ws = As + CFG::A_BYTES;
            const bool more = kt + 1 < nkt;
            const int nb = (kt + 1) & 1;
            // Each wait leaves only the two most recent regions in flight: the region the NEXT phase
            // reads was issued three phases ago.  All waves pass the wait before the barrier that
            // precedes the first reader (the other M-half runs one barrier behind).  The DMA goes out
            // AFTER the fragment reads of the segment (measured: DMA-first costs 15 %, the LDS-DMA
            // issue back-pressures the ds_reads behind it), and every region is re-staged >= 2 phases
            // after its last read.  Ablation on 8192^3 (profiles/r01_gemm_ablation.md): waits +3 %,
            // DMA issue +20 %, fragment reads +13 % of the MFMA-only schedule (1.72 PF).
            // phase 1: quadrant (m-half 0, n-tile 0)
            rd_a(As, 0);
            rd_w(Ws, 0);
            if (more) stage_a(nb, kt + 1, 0);
            if (more) { VMCNT(4); } else { VMCNT(2); }
            PP_SYNC();
            mma(0, 0);
            PP_BAR();
            // phase 2: (m-half 0, n-tile 1)
            rd_w(Ws, 1);
            if (more) stage_w(nb, kt + 1, 0);
            if (more) { VMCNT(4); } else { VMCNT(0); }
            PP_SYNC();
            mma(0, 1);
            PP_BAR();
            // phase 3: (m-half 1, n-tile 1)
            rd_a(As, 1);
            if (more) stage_w(nb, kt + 1, 1);
            if (more) VMCNT(4);
            PP_SYNC();
            mma(1, 1);
            PP_BAR();
            // phase 4: (m-half 1, n-tile 0)
            rd_w(Ws, 0);
            if (more) stage_a(nb, kt + 1, 1);
            if (more) VMCNT(4);
            PP_SYNC();
            mma(1, 0);
            PP_BAR();
        }
        if (wm == 0) PP_BAR();    // balance the barrier count of the two halves
#undef VMCNT
#undef PP_SYNC
#undef PP_BAR
    }

#if APEXMI_GEMM_TRACE
    if (G.trace != nullptr) {
        __builtin_amdgcn_sched_barrier(0);
        tr_l1 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
    if (G.clk != nullptr) {
        const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            atomicAdd(G.clk, c1 - clk_c0);
            atomicAdd(G.clk + 1, r1 - clk_r0);
        }
    }

    // ---- stream-K hand-over (SCHED 5): partial sums in accumulator order, 16 bytes per lane, fully coalesced ----
    // CDNA guide §6 Guideline 16, the fence-free form: write-through (sc1) slab stores -> every wave vmcnt(0) -> workgroup barrier
    // -> ONE lane raises the flag (relaxed, agent scope); the owner polls relaxed, then reads the slab with sc1 loads.  No
    // release / acquire fence: an agent-scope release writes the XCD's whole L2 back (every dirty output line of the launch) and
    // cost ~20 us per hand-over here.  Slab addressing through a buffer descriptor: one VGPR offset (16 * tid) + a scalar offset per
    // 8 KiB row, instead of thirty-two 64-bit address pairs (which the allocator spilled the accumulators for).
    if constexpr (CFG::SCHED == 5 && APEXMI_GEMM_STREAMK) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
        if (sk.mode == TILE_WRITER) {
            auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(G.sk_slab + (size_t)sk.cu * 65536), 0, 262144, 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, acc16[i][j]), rs, tid * 16, (i * 8 + j) * 8192, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(G.sk_flag + sk.cu, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (sk.mode == TILE_OWNER) {
            if (tid == 0)
                for (int p = 1; p <= sk.nparts; ++p)
                    while (__hip_atomic_load(G.sk_flag + sk.cu - p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
            __syncthreads();
            // fixed order: own K range + the range before it + the one before that (deterministic, launch to launch)
            for (int p = 1; p <= sk.nparts; ++p) {
                auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(G.sk_slab + (size_t)(sk.cu - p) * 65536), 0, 262144, 0x00020000);
#pragma unroll
                for (int i = 0; i < 4; ++i) {           // eight 16-byte loads in flight at a time (the accumulators hold 128 VGPRs)
                    u32x4_ v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, (i * 8 + j) * 8192, 16);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc16[i][j] += __builtin_bit_cast(f32x4_t, v[j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
            if (tid == 0)            // every slab has exactly one reader: lower its flag for the next launch on this stream
                for (int p = 1; p <= sk.nparts; ++p) __hip_atomic_store(G.sk_flag + sk.cu - p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- epilogue ----
    if constexpr (CFG::SCHED == 5 || CFG::SCHED >= 7) {
        if constexpr (EPI == APEXMI_EPI_BIAS) {
            if (P.qkv) {                                // block-uniform
                qkv_epilogue16(acc16, P, G.qs, M, m0, n0, wave, wm, wn, lane, smem);
                APEXMI_TRACE_END();
                return;
            }
        }
        int mrow16[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            mrow16[mt] = m0 + wm * 128 + mt * 16 + (lane & 15);
            if (mrow16[mt] >= M) mrow16[mt] = -1;
        }
        if constexpr (EPI == APEXMI_EPI_BIAS_GATE_RES) {
            // the residual rows of BOTH 32-column slabs before the first store: C may alias R, so hipcc keeps slab 1's loads behind
            // slab 0's stores, a second memory round trip with every matrix pipe idle (gemm_tile_trace: 6.3-8 us against 3-4 us
            // for the other epilogues)
            u32x4 rr2[2][8];
            const int g = lane >> 4;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int nst = n0 + wn * 64 + p * 32 + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
                    rr2[p][mt] = *(const u32x4*)(P.R + (int64_t)max(mrow16[mt], 0) * P.ldr + min(nst, N - 8));
            }
#pragma unroll
            for (int p = 0; p < 2; ++p)
                store_slab16<EPI, 8, 0>(acc16[2 * p], acc16[2 * p + 1], P, N, mrow16, n0 + wn * 64 + p * 32, g, rr2[p]);
        } else {
            APEXMI_ACT_DISPATCH((EPI == APEXMI_EPI_BIAS || EPI == APEXMI_EPI_BIAS_F32) ? P.gelu : 0,
                                _Pragma("unroll") for (int p = 0; p < 2; ++p)
                                    store_slab16<EPI, 8, ACT>(acc16[2 * p], acc16[2 * p + 1], P, N, mrow16, n0 + wn * 64 + p * 32, lane >> 4));
        }
        APEXMI_TRACE_END();
        return;
    }
    int mrow[TM];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        mrow[mt] = m0 + wm * (BM / CFG::WM) + mt * 32 + l31;
        if (mrow[mt] >= M) mrow[mt] = -1;  // lane stays alive for the cross-lane exchange, stores nothing
    }
    APEXMI_ACT_DISPATCH((EPI == APEXMI_EPI_BIAS || EPI == APEXMI_EPI_BIAS_F32) ? P.gelu : 0,
                        _Pragma("unroll") for (int nt = 0; nt < TN; ++nt)
                            store_ntile<EPI, TM, ACT>(acc[nt], P, N, mrow, n0 + wn * (BN / CFG::WN) + nt * 32, hi));
}

// The launch.  Normally one workgroup per tile.  STREAM-K (G.sk_r > 0, SCHED 5): 256 persistent workgroups, 32 per XCD.
// Launches whose tile count is not a multiple of the 256 CUs end in a part-filled round — 216 tiles (attention-out, FF-down,
// the single block's proj_out: 42 % of the Flux step's GEMM flops) leave 40 CUs idle for the whole launch, 648 tiles (QKV of a
// double block) run a third round at 53 % — and the live clock probe shows the GEMM is NOT power-bound (2.05 GHz), so those idle
// CUs are lost time, not clock given back.  Here the sk_r tiles past the last full round are shared out by K: XCD x takes
// r_x = sk_r / 8 (+1) of them, its 32 workgroups cut the r_x * nkt K-tiles into 32 equal unit ranges.  A workgroup's range is
// [tail of a tile begun by the workgroup before it | whole tiles | head of a tile the next workgroup finishes]:
//   * the HEAD (a tile whose last K-tiles lie in a higher workgroup) is computed FIRST and leaves as an f32 slab + flag (WRITER);
//   * then whole tiles of the range and the workgroup's tiles of the full rounds (ordinary epilogue);
//   * the TAIL (the tile's last K-tiles are here) is computed LAST, adds the slabs of the one or two workgroups before it, and runs
//     the epilogue (OWNER).
// An owner only ever waits for workgroups with a LOWER block id on its own XCD (dispatched before it, slab written at the start of
// their work), so the hand-over is deadlock-free whatever else shares the chip, slabs stay in the XCD's L2, and the f32 sum order
// (own range, then the ranges before it) is fixed: results are deterministic; they differ from the one-workgroup-per-tile launch
// by f32 summation order only.
template <typename CFG, int EPI>
__global__ __launch_bounds__(CFG::NT, CFG::OCC) void gemm_bf16_kernel(const GemmGroup G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nkt = G.K / BK;
    SkCtx sk{TILE_FULL, 0, nkt, 0, 0};
    if constexpr (CFG::SCHED != 5) {
        gemm_tile<CFG, EPI>(G, smem, xcd_remap(blockIdx.x, G.total), sk);
    } else {
        if (!APEXMI_GEMM_STREAMK || (G.sk_r <= 0 && G.sk_tfull <= 0)) {
            gemm_tile<CFG, EPI>(G, smem, xcd_remap(blockIdx.x, G.total), sk);
            return;
        }
        // The work list of this workgroup is RE-DERIVED from (block id, item index) at the top of every iteration, behind an opaque
        // copy of the block id: a dozen scalars kept live across the K-loop instead cost SGPR spills into VGPR lanes and, with
        // them, VGPR spills inside the K-loop (whose scratch loads would also break the counted vmcnt waits).
        for (int it = 0;; ++it) {
            int bid = blockIdx.x;
            asm volatile("" : "+s"(bid));
            const int x = bid & 7, i = bid >> 3;                                // XCD, workgroup within it (dispatch order)
            const int rx = G.sk_r / 8 + (x < G.sk_r % 8 ? 1 : 0);               // remainder tiles of this XCD ...
            const int r0 = G.sk_tfull + x * (G.sk_r / 8) + min(x, G.sk_r % 8);  // ... starting at this sequence index
            const int64_t units = (int64_t)rx * nkt;
            auto ustart = [&](int w) { return (int)((units * w) / 32); };
            const int u0 = ustart(i), u1 = ustart(i + 1);
            int j0 = 0, j1 = -1;                                                // first / last remainder tile the range touches
            if (u1 > u0) {
                j0 = u0 / nkt;
                j1 = (u1 - 1) / nkt;
            }
            // in order: [head -> WRITER] [whole tiles of the range] [its tiles of the full rounds] [tail -> OWNER]
            const bool has_head = u1 > u0 && u1 < (j1 + 1) * nkt;               // the range's LAST tile is finished by a higher workgroup
            const bool has_tail = u1 > u0 && u0 > j0 * nkt && !(has_head && j0 == j1);   // its FIRST tile was begun by lower ones
            const int w_lo = j0 + ((u1 > u0 && u0 > j0 * nkt) ? 1 : 0);         // whole tiles of the range: [w_lo, w_hi]
            const int w_hi = j1 - (has_head ? 1 : 0);
            const int n_whole = max(w_hi - w_lo + 1, 0);
            const int per_xcd = G.sk_tfull / 8;
            const int n_full = per_xcd > i ? (per_xcd - i + 31) / 32 : 0;
            const int n_items = (has_head ? 1 : 0) + n_whole + n_full + (has_tail ? 1 : 0);
            if (it >= n_items) break;
            int k = it, tile;
            sk.mode = TILE_FULL;
            sk.kt0 = 0;
            sk.kt1 = nkt;
            sk.nparts = 0;
            sk.cu = x * 32 + i;
            if (has_head && k == 0) {
                sk.mode = TILE_WRITER;
                sk.kt0 = max(u0, j1 * nkt) - j1 * nkt;
                sk.kt1 = u1 - j1 * nkt;
                tile = r0 + j1;
            } else {
                k -= has_head ? 1 : 0;
                if (k < n_whole) {
                    tile = r0 + w_lo + k;
                } else if (k < n_whole + n_full) {
                    tile = x * per_xcd + i + 32 * (k - n_whole);
                } else {
                    sk.mode = TILE_OWNER;
                    sk.kt0 = u0 - j0 * nkt;
                    sk.kt1 = min(u1, (j0 + 1) * nkt) - j0 * nkt;
                    for (int w = i - 1; w >= 0 && sk.nparts < 3; --w) {         // workgroups holding K-tiles [0, kt0) of tile j0
                        if (ustart(w + 1) <= j0 * nkt) break;
                        ++sk.nparts;
                    }
                    tile = r0 + j0;
                }
            }
            if (it) __syncthreads();            // the staging LDS (and the QKV epilogue's scratch) of the previous item is free
            gemm_tile<CFG, EPI>(G, smem, tile, sk);
        }
    }
}

// ---- exact bf16 split of an f32 operand (f32-storage verification mode) ------------------------------------------
// out[m][j K + k] = part_j(x[m][k]), j = 0..2: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid).  The three
// parts sum to x exactly (3 x 8 mantissa bits), so a bf16 MFMA GEMM over the K-concatenated operand against
// [W | W | W] computes sum_k x[m][k] W[n][k] with EXACT products and f32 accumulation: the production main loop, fed
// an operand that carries no activation rounding.
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ x, int64_t ldx, int64_t M, int K,
                                                            bf16_t* __restrict__ out, int64_t ldo) {
    const int kc = K >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * kc) return;
    const int64_t m = idx / kc;
    const int c = (int)(idx % kc) * 8;
    float v[8], h[8], md[8], lo[8];
    load8<float>(x + m * ldx + c, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h[j] = bf16_to_f32(f32_to_bf16(v[j]));
        const float r1 = v[j] - h[j];
        md[j] = bf16_to_f32(f32_to_bf16(r1));
        lo[j] = r1 - md[j];
    }
    bf16_t* o = out + m * ldo + c;
    store8<bf16_t>(o, h);
    store8<bf16_t>(o + K, md);
    store8<bf16_t>(o + 2 * (int64_t)K, lo);
}

// ---- 288 x 192 tile: the EXACT-FILL tiling (round 5) --------------------------------------------------------
// The Flux single block's proj_out (M 4608 joint rows, N 3072, K 15360: 28 % of the step's GEMM flops) is 18 x 12 = 216 tiles of
// 256 x 256 on 256 CUs: one round with 40 CUs idle.  As 288 x 192 it is 16 x 16 = 256 tiles, one FULL round of tiles that carry
// 0.84 of a 256 x 256 tile's work.  Same machine as the shipped schedule (SCHED 5): 8 waves (2 x 4), v_mfma_f32_16x16x32_bf16,
// 16-byte LDS-DMA pieces into a double-buffered image with the (row >> 1) & 7 chunk swizzle, the two M-halves of the block one
// barrier apart so that every SIMD has one wave in its MFMA segment and one in its LDS / DMA segment.  What differs:
//   * a wave owns 144 x 48 = 9 x 3 accumulator tiles (108 VGPRs); a K-tile is THREE phases (m-groups of 3 tiles x all 3 n-tiles:
//     18 MFMAs each); the weight fragments (6 reads) are read once per K-tile in phase 0, 6 activation-fragment reads per phase:
//     24 ds_read_b128 per 54 MFMAs (shipped: 24 per 64);
//   * a K-tile is 60 pieces of 8 rows (36 activation + 24 weight) = 7.5 per wave: issued as 8 per wave, the last four of the
//     phase-0 class being duplicates of its first four (same bytes to the same LDS address), so that every wave's counted
//     vmcnt waits are the same: pieces j = 0..4 of a wave are what the NEXT K-tile's phase 0 reads (all weight rows + m-group 0),
//     j = 5, 6 complete m-group 1, j = 7 m-group 2; issued 3 + 3 + 2 over the three phases of the current K-tile;
//   * the third n-tile of a wave has no partner for the 8-consecutive-columns exchange: 8-byte epilogue accesses for that third.
// Every output element is accumulated over K in the same order as on the 256 x 256 tiling (K-tiles in sequence, two 32-deep
// MFMAs each), so results are BIT-IDENTICAL to it (tests/test_gpu_ops.py::test_gemm_x288_tiling_is_bit_identical).
//
// MEASURED (profiles/r05_gemm_x288_ab.log, r05_gemm_fill_probe.log) AND NOT SHIPPED (`gemm.x288` = 0): proj_out 351 us against
// 339 us on 256 x 256 (-3.5 %), every other part-filled shape -0.1 .. -7 %, the Flux step 67.6 against 66.9 ms.  The premise —
// "40 idle CUs are lost time" — is wrong for this kernel: a K-tile of a 256 x 256 workgroup takes 1.23 us while <= 128 CUs hold a
// tile, 1.34 us at 192, 1.43 us at 224 and 1.53 us at 256 (the same curve for 288 x 192: 1.23 -> 1.42 us), i.e. the chip delivers
// ~11 TB/s of L2 -> LDS staging in aggregate however many CUs ask for it, and a K-tile never takes less than ~1.23 us on a CU
// although its MFMAs need 1.0 us (0.84 us here).  Filling the last 40 CUs therefore slows the other 216 down by what the 40 add,
// and a tile that stages 11 % more bytes per flop loses.  The only lever left on this term is FEWER staged bytes per flop
// (a 384 x 256 tile: 192 accumulator registers of 256; fp8 weight pieces: profiles/r05_gemm_roof_fp8.log).
// Kept as a selectable tiling (`gemm.x288` = 1 | 2) for the record and for the A/B tools.
constexpr int XBM = 288, XBN = 192;
constexpr int X_A_BYTES = XBM * BK * 2, X_W_BYTES = XBN * BK * 2, X_STAGE = X_A_BYTES + X_W_BYTES, X_LDS = 2 * X_STAGE;

// one 16-column n-tile of a wave (accumulator layout: lane (g, c) holds C[m = 16 mt + c][nb + 4 g + (0..3)]): 8-byte accesses
template <int EPI, int MT, int ACT = 0>
APEXMI_DEVICE void store_single16(const f32x4_t (&x)[MT], const GemmProblem& P, int N, const int (&m)[MT], int nb, int g,
                                  const u32x2* rr_pre) {
    const int nl = min(nb + 4 * g, N - 4);
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    if (P.bias != nullptr) {
        const u32x2 b = *(const u32x2*)(P.bias + nl);
        bs[0] = bf16_lo(b[0]);
        bs[1] = bf16_hi(b[0]);
        bs[2] = bf16_lo(b[1]);
        bs[3] = bf16_hi(b[1]);
    }
    f32x4 gt = {0.f, 0.f, 0.f, 0.f};
    if (EPI == APEXMI_EPI_BIAS_GATE_RES) gt = *(const f32x4*)(P.gate + nl);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_c<ACT>(x[mt][j] + bs[j]);
        if (EPI == APEXMI_EPI_BIAS_GATE_RES) {
            const u32x2 r = rr_pre[mt];
            v[0] = bf16_lo(r[0]) + gt[0] * v[0];
            v[1] = bf16_hi(r[0]) + gt[1] * v[1];
            v[2] = bf16_lo(r[1]) + gt[2] * v[2];
            v[3] = bf16_hi(r[1]) + gt[3] * v[3];
        }
        if (m[mt] >= 0 && nb + 4 * g < N)
            *(u32x2*)(P.C + (int64_t)m[mt] * P.ldc + nb + 4 * g) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
    }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_x288_kernel(const GemmGroup G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;      // waves w and w + 4 share a SIMD: same column strip, the other M-half

    // ---- tile id -> problem, (pm, pn): XCD-contiguous, grouped group_m tall (as gemm_tile) ----
    int s = xcd_remap(blockIdx.x, G.total);
    int gi = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUPS; ++i)
        if (i < G.count && s >= G.p[i].tile0) gi = i;
    const GemmProblem P = G.p[gi];
    s -= P.tile0;
    const int nn = P.nn, N = P.N, M = P.M;
    const int GM = G.group_m;
    const int width = GM * nn;
    const int first_m = (s / width) * GM;
    const int gsz = min(P.nm - first_m, GM);
    const int pm = first_m + (s % width) % gsz;
    const int pn = (s % width) / gsz;
    const int m0 = pm * XBM, n0 = pn * XBN;

    // ---- the wave's 8 LDS-DMA pieces of a K-tile (8 rows x 128 B each), in need order ----
    const char* src[8];
    int dst[8];                                   // byte offset of the piece inside a stage (wave-uniform)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int row0;
        bool isw = false;
        if (j < 3) {                              // all 192 weight rows
            isw = true;
            row0 = (j * 8 + wave) * 8;
        } else if (j < 5) {                       // m-group 0 of both halves (rows 0..47, 144..191) + four duplicates
            const int q = (j - 3) * 8 + wave;
            row0 = q < 6 ? q * 8 : q < 12 ? 144 + (q - 6) * 8 : (q - 12) * 8;
        } else {                                  // m-group 1 (rows 48..95, 192..239), then m-group 2 (96..143, 240..287)
            const int q = (j - 5) * 8 + wave;
            row0 = q < 6 ? 48 + q * 8 : q < 12 ? 192 + (q - 6) * 8 : q < 18 ? 96 + (q - 12) * 8 : 240 + (q - 18) * 8;
        }
        const int row = row0 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        src[j] = isw ? (const char*)(P.W + (int64_t)min(n0 + row, N - 1) * P.ldw + c * 8)
                     : (const char*)(P.A + (int64_t)min(m0 + row, M - 1) * P.lda + c * 8);
        dst[j] = (isw ? X_A_BYTES : 0) + row0 * 128;
    }
    auto stage = [&](int buf, int kt, int j) { glds16(src[j] + (int64_t)kt * (BK * 2), smem + buf * X_STAGE + dst[j]); };

    f32x4_t acc[3][9];                            // [16-column n-tile][16-row m-tile]
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nkt = G.K / BK;
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (G.clk != nullptr) {
        clk_c0 = __builtin_readcyclecounter();
        clk_r0 = __builtin_amdgcn_s_memrealtime();
    }

    // fragment reads: lane (l15, g4) reads row (tile base + l15), 16-byte chunk (4 ks + g4) ^ swizzle
    const int l15 = lane & 15, g4 = lane >> 4;
    const int sw16 = (l15 >> 1) & 7;
    const int a_base = (wm * 144 + l15) * 128, w_base = X_A_BYTES + (wn * 48 + l15) * 128;
    const int ch16[2] = {((0 + g4) ^ sw16) << 4, ((4 + g4) ^ sw16) << 4};
    bf16x8 af[3][2], wf[3][2];
    auto rd_a = [&](const char* St, int grp) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[t][ks] = *(const bf16x8*)(St + a_base + (grp * 3 + t) * 2048 + ch16[ks]);
    };
    auto rd_w = [&](const char* St) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wf[t][ks] = *(const bf16x8*)(St + w_base + t * 2048 + ch16[ks]);
    };
    auto mma = [&](int grp) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    acc[nt][grp * 3 + t] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], af[t][ks], acc[nt][grp * 3 + t], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
#define X_SYNC()                                                                    \
    do {                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                          \
        __builtin_amdgcn_s_barrier();                                               \
        __builtin_amdgcn_sched_barrier(0);                                          \
    } while (0)
#define X_BAR()                                \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)
#define X_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#pragma unroll
    for (int j = 0; j < 8; ++j) stage(0, 0, j);
    X_VMCNT(0);
    X_BAR();
    if (wm == 1) X_BAR();
    for (int kt = 0; kt < nkt; ++kt) {
        const char* St = smem + (kt & 1) * X_STAGE;
        const bool more = kt + 1 < nkt;
        const int nb = (kt + 1) & 1;
        // phase 0: m-group 0.  A wave's waits cover what the NEXT phase reads (the partner half passes this barrier before it reads)
        rd_w(St);
        rd_a(St, 0);
        if (more) {
            stage(nb, kt + 1, 0);
            stage(nb, kt + 1, 1);
            stage(nb, kt + 1, 2);
            X_VMCNT(4);                           // in flight at most: j = 7 of this K-tile + the three just issued -> m-group 1 landed
        } else {
            X_VMCNT(1);
        }
        X_SYNC();
        mma(0);
        X_BAR();
        // phase 1: m-group 1
        rd_a(St, 1);
        if (more) {
            stage(nb, kt + 1, 3);
            stage(nb, kt + 1, 4);
            stage(nb, kt + 1, 5);
            X_VMCNT(6);                           // the six pieces of the next K-tile only -> m-group 2 landed
        } else {
            X_VMCNT(0);
        }
        X_SYNC();
        mma(1);
        X_BAR();
        // phase 2: m-group 2
        rd_a(St, 2);
        if (more) {
            stage(nb, kt + 1, 6);
            stage(nb, kt + 1, 7);
            X_VMCNT(3);                           // j = 5, 6, 7 of the next K-tile may fly: its weights and m-group 0 landed
        }
        X_SYNC();
        mma(2);
        X_BAR();
    }
    if (wm == 0) X_BAR();                         // balance the barrier count of the two halves
#undef X_VMCNT
#undef X_SYNC
#undef X_BAR

    if (G.clk != nullptr) {
        const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            atomicAdd(G.clk, c1 - clk_c0);
            atomicAdd(G.clk + 1, r1 - clk_r0);
        }
    }

    // ---- epilogue: a 32-column slab (n-tiles 0, 1: 16-byte accesses) + the single n-tile 2 (8-byte accesses) ----
    int mrow[9];
#pragma unroll
    for (int mt = 0; mt < 9; ++mt) {
        mrow[mt] = m0 + wm * 144 + mt * 16 + l15;
        if (mrow[mt] >= M) mrow[mt] = -1;
    }
    const int nbase = n0 + wn * 48;
    if constexpr (EPI == APEXMI_EPI_BIAS_GATE_RES) {
        // every residual row before the first store (C may alias R: the compiler cannot move loads above stores itself)
        u32x4 rr[9];
        u32x2 r1[9];
        const int nst = nbase + 16 * (g4 & 1) + 8 * (g4 >> 1);
#pragma unroll
        for (int mt = 0; mt < 9; ++mt) {
            const bf16_t* row = P.R + (int64_t)max(mrow[mt], 0) * P.ldr;
            rr[mt] = *(const u32x4*)(row + min(nst, N - 8));
            r1[mt] = *(const u32x2*)(row + min(nbase + 32 + 4 * g4, N - 4));
        }
        store_slab16<EPI, 9, 0>(acc[0], acc[1], P, N, mrow, nbase, g4, rr);
        store_single16<EPI, 9, 0>(acc[2], P, N, mrow, nbase + 32, g4, r1);
    } else {
        APEXMI_ACT_DISPATCH(P.gelu, store_slab16<EPI, 9, ACT>(acc[0], acc[1], P, N, mrow, nbase, g4);
                            store_single16<EPI, 9, ACT>(acc[2], P, N, mrow, nbase + 32, g4, nullptr));
    }
}

// ---- 384 x 256 tile: FEWER STAGED BYTES PER FLOP (round 5) ----------------------------------------------------------------------
// The shipped 256 x 256 launch is bound by L2 -> LDS staging, not by its MFMAs (see the 288 x 192 kernel's header: a K-tile takes
// 1.23 us on a lightly loaded chip and 1.53 us with all 256 CUs busy, ~11 TB/s in aggregate, while its MFMAs need 1.0 us), and the
// time per flop of the two tilings measured so far is proportional to their staged bytes per flop (1.10 x for 1.11 x).  A 384 x 256
// tile stages (384 + 256) x 128 B per 2 x 384 x 256 x 64 flops: 0.833 of the 256 x 256 tile's bytes per flop.  What it costs:
//   * 192 accumulator registers per lane (8 waves of 192 x 64 = 12 x 4 tiles of 16 x 16) of the 256 a wave has at two waves per
//     SIMD, so everything else is lean: fragments single-buffered per k-step (6 activation + 4 weight reads feed 24 MFMAs), the
//     LDS-DMA pieces addressed through two buffer descriptors with ONE per-lane offset each (a wave only takes pieces of its own
//     row parity, so the source swizzle is the same for all of them) and scalar piece / K offsets; rows past M / N fall outside
//     the descriptor's range and read as zero (no clamping);
//   * the whole LDS of a CU: 2 x (48 + 32) KiB.
// Schedule: the ping-pong of the shipped kernel — the two M-halves of the block one barrier apart, every SIMD with one wave in its
// MFMA segment and one in its LDS / DMA segment — with four phases per K-tile = (k-step, m-half): 24 MFMAs each; a wave's 10 pieces
// of the NEXT K-tile go out 3 + 3 + 2 + 2 over the phases: 4 weight + 3 activation pieces of m-half 0 first (what the next phase 0
// reads), then the 3 of m-half 1 (phase 1), behind counted vmcnt(3) waits.
// Same K order per output element as the other tilings: results are BIT-IDENTICAL to the 256 x 256 launch.
constexpr int YBM = 384, YBN = 256;
constexpr int Y_A_BYTES = YBM * BK * 2, Y_W_BYTES = YBN * BK * 2, Y_STAGE = Y_A_BYTES + Y_W_BYTES, Y_LDS = 2 * Y_STAGE;

template <int EPI, int DIST = 1, int QKV = 0>      // QKV = 1: launches that carry a fused q/k/v problem (their own instantiation: its epilogue
                                                   // spills a few registers, which the plain launches must not pay for)
__global__ __launch_bounds__(512, 2) void gemm_bf16_x384_kernel(const GemmGroup G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;      // waves w and w + 4 share a SIMD: same column strip, the other M-half

    int s = xcd_remap(blockIdx.x, G.total);
    int gi = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUPS; ++i)
        if (i < G.count && s >= G.p[i].tile0) gi = i;
    const GemmProblem P = G.p[gi];
    s -= P.tile0;
    const int nn = P.nn, N = P.N, M = P.M;
    const int GM = G.group_m;
    const int width = GM * nn;
    const int first_m = (s / width) * GM;
    const int gsz = min(P.nm - first_m, GM);
    const int pm = first_m + (s % width) % gsz;
    const int pn = (s % width) / gsz;
    const int m0 = pm * YBM, n0 = pn * YBN;

    // ---- LDS-DMA: two buffer descriptors, one per-lane offset each, scalar piece offsets ----
    const int rows_a = min(M - m0, YBM), rows_w = min(N - n0, YBN);
    auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A + (int64_t)m0 * P.lda), 0,
                                                    (int)(((int64_t)(rows_a - 1) * P.lda + G.K) * 2), 0x00020000);
    auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)(P.W + (int64_t)n0 * P.ldw), 0,
                                                    (int)(((int64_t)(rows_w - 1) * P.ldw + G.K) * 2), 0x00020000);
    // a wave takes pieces (8 rows each) of ONE parity: (row >> 1) & 7 = 4 (piece & 1) + (lane >> 4) for every one of them
    const int par = wave & 1, widx = wave >> 1;
    const int sw_src = ((lane & 7) ^ (4 * par + (lane >> 4))) * 16;
    const int voff_a = (lane >> 3) * (int)P.lda * 2 + sw_src;
    const int voff_w = (lane >> 3) * (int)P.ldw * 2 + sw_src;
    const int lda8 = (int)P.lda * 16, ldw8 = (int)P.ldw * 16;         // bytes per 8-row piece step
    // piece j of the wave: 0..3 weight pieces, 4..6 activation pieces of m-half 0 (rows 0..95 and 192..287), 7..9 of m-half 1
    auto piece_of = [&](int j) -> int {
        if (j < 4) return 2 * (4 * j + widx) + par;
        const int e = 3 * widx + (j < 7 ? j - 4 : j - 7);            // 0..11 within the parity's list
        return (e < 6 ? 0 : 24) + (j < 7 ? 0 : 12) + 2 * (e % 6) + par;
    };
    auto stage = [&](int buf, int kt, int j) {
        const int pc = piece_of(j);
        char* dst = smem + buf * Y_STAGE + (j < 4 ? Y_A_BYTES : 0) + pc * 1024;
        if (j < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)dst, 16, voff_w, pc * ldw8 + kt * (BK * 2), 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)dst, 16, voff_a, pc * lda8 + kt * (BK * 2), 0, 0);
    };

    f32x4_t acc[4][12];                           // [16-column n-tile][16-row m-tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 12; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nkt = G.K / BK;
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (G.clk != nullptr) {
        clk_c0 = __builtin_readcyclecounter();
        clk_r0 = __builtin_amdgcn_s_memrealtime();
    }
#if APEXMI_GEMM_TRACE
    // per-workgroup timeline (tools/gemm_x384_trace.py): [kind, entry, K-loop end, epilogue stores acknowledged] in 10 ns ticks;
    // kind = 0 q / 1 k / 2 v tile of a fused q/k/v problem, 3 any other epilogue
    const unsigned long long y_t_in = G.trace != nullptr ? __builtin_amdgcn_s_memrealtime() : 0;
    unsigned long long y_t_loop = 0;
#define Y_TRACE_END(kind)                                                                                   \
    do {                                                                                                    \
        if (G.trace != nullptr) {                                                                           \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
            __syncthreads();                                                                                \
            if (tid == 0) {                                                                                 \
                unsigned long long* o = G.trace + (size_t)blockIdx.x * 4;                                   \
                o[0] = (unsigned long long)(kind);                                                          \
                o[1] = y_t_in;                                                                              \
                o[2] = y_t_loop;                                                                            \
                o[3] = __builtin_amdgcn_s_memrealtime();                                                    \
            }                                                                                               \
        }                                                                                                   \
    } while (0)
#else
#define Y_TRACE_END(kind) do { } while (0)
#endif

    const int l15 = lane & 15, g4 = lane >> 4;
    const int sw16 = (l15 >> 1) & 7;
    const int a_base = (wm * 192 + l15) * 128, w_base = Y_A_BYTES + (wn * 64 + l15) * 128;
    const int ch16[2] = {((0 + g4) ^ sw16) << 4, ((4 + g4) ^ sw16) << 4};
    bf16x8 af[6], wf[4];
    auto rd_a = [&](const char* St, int h, int ks) {
#pragma unroll
        for (int t = 0; t < 6; ++t) af[t] = *(const bf16x8*)(St + a_base + (h * 6 + t) * 2048 + ch16[ks]);
    };
    auto rd_w = [&](const char* St, int ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) wf[t] = *(const bf16x8*)(St + w_base + t * 2048 + ch16[ks]);
    };
    auto mma = [&](int h) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int t = 0; t < 6; ++t)
                acc[nt][h * 6 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[t], acc[nt][h * 6 + t], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
#define Y_SYNC()                                                                    \
    do {                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                          \
        __builtin_amdgcn_s_barrier();                                               \
        __builtin_amdgcn_sched_barrier(0);                                          \
    } while (0)
#define Y_BAR()                                \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)
#define Y_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#pragma unroll
    for (int j = 0; j < 10; ++j) stage(0, 0, j);
    Y_VMCNT(0);
    Y_BAR();
    if (wm == 1) Y_BAR();
    // last K-tile peeled off: no `more` test in the steady-state body (round 6)
    auto ktile = [&](int kt, auto more_c) {
        constexpr bool more = decltype(more_c)::value;
        const char* St = smem + (kt & 1) * Y_STAGE;
        const int nb = (kt + 1) & 1;
        // The wave's 10 pieces of the next K-tile go out over the four phases as 2 + 3 + 2 + 3 (DIST 1: the phases that read 10
        // fragments issue two pieces, those that read 6 issue three) or 3 + 3 + 2 + 2 (DIST 0).  A wave's wait covers what the NEXT
        // phase reads: phase 0 -> m-half 1 of THIS K-tile (its pieces 7..9, issued during the previous K-tile), phase 3 -> the
        // weights and m-half 0 of the next one (pieces 0..6: 7..9 may fly).
        // phase 0: k-step 0, m-half 0
        rd_w(St, 0);
        rd_a(St, 0, 0);
        if constexpr (more) {
            stage(nb, kt + 1, 0);
            stage(nb, kt + 1, 1);
            if (DIST == 0) {
                stage(nb, kt + 1, 2);
                Y_VMCNT(3);
            } else {
                Y_VMCNT(2);
            }
        } else {
            Y_VMCNT(0);
        }
        Y_SYNC();
        mma(0);
        Y_BAR();
        // phase 1: k-step 0, m-half 1
        rd_a(St, 1, 0);
        if constexpr (more) {
            if (DIST != 0) stage(nb, kt + 1, 2);
            stage(nb, kt + 1, 3);
            stage(nb, kt + 1, 4);
            if (DIST == 0) stage(nb, kt + 1, 5);
        }
        Y_SYNC();
        mma(1);
        Y_BAR();
        // phase 2: k-step 1, m-half 0
        rd_w(St, 1);
        rd_a(St, 0, 1);
        if constexpr (more) {
            if (DIST != 0) stage(nb, kt + 1, 5);
            stage(nb, kt + 1, 6);
            if (DIST == 0) stage(nb, kt + 1, 7);
        }
        Y_SYNC();
        mma(0);
        Y_BAR();
        // phase 3: k-step 1, m-half 1
        rd_a(St, 1, 1);
        if constexpr (more) {
            if (DIST != 0) stage(nb, kt + 1, 7);
            stage(nb, kt + 1, 8);
            stage(nb, kt + 1, 9);
            Y_VMCNT(3);
        }
        Y_SYNC();
        mma(1);
        Y_BAR();
    };
    for (int kt = 0; kt + 1 < nkt; ++kt) ktile(kt, std::true_type{});
    if (nkt > 0) ktile(nkt - 1, std::false_type{});
    if (wm == 0) Y_BAR();
#undef Y_VMCNT
#undef Y_SYNC
#undef Y_BAR

    if (G.clk != nullptr) {
        const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            atomicAdd(G.clk, c1 - clk_c0);
            atomicAdd(G.clk + 1, r1 - clk_r0);
        }
    }

#if APEXMI_GEMM_TRACE
    if (G.trace != nullptr) y_t_loop = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (EPI == APEXMI_EPI_BIAS && QKV == 1) {
        if (P.qkv) {                                   // block-uniform: the fused q / k / v preparation (two passes for V^T)
            qkv_epilogue16<12, 192>(acc, P, G.qs, M, m0, n0, wave, wm, wn, lane, smem);
            Y_TRACE_END(n0 / G.qs.inner);
            return;
        }
    }
    // ---- epilogue: per 32-column slab and a few m-tiles at a time (the accumulators leave 64 registers for everything else) ----
    constexpr int MTC = EPI == APEXMI_EPI_BIAS_GATE_RES ? 3 : 6;      // m-tiles per call (gate / residual also holds the residual rows)
#pragma unroll
    for (int h = 0; h < 12 / MTC; ++h) {
        int mrow[MTC];
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt) {
            mrow[mt] = m0 + wm * 192 + (h * MTC + mt) * 16 + l15;
            if (mrow[mt] >= M) mrow[mt] = -1;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int nbase = n0 + wn * 64 + p * 32;
            const f32x4_t(&x)[MTC] = *(const f32x4_t(*)[MTC]) & acc[2 * p][h * MTC];
            const f32x4_t(&y)[MTC] = *(const f32x4_t(*)[MTC]) & acc[2 * p + 1][h * MTC];
            if constexpr (EPI == APEXMI_EPI_BIAS_GATE_RES) {
                u32x4 rr[MTC];
                const int nst = nbase + 16 * (g4 & 1) + 8 * (g4 >> 1);
#pragma unroll
                for (int mt = 0; mt < MTC; ++mt) rr[mt] = *(const u32x4*)(P.R + (int64_t)max(mrow[mt], 0) * P.ldr + min(nst, N - 8));
                store_slab16<EPI, MTC, 0>(x, y, P, N, mrow, nbase, g4, rr);
            } else {
                APEXMI_ACT_DISPATCH(P.gelu, store_slab16<EPI, MTC, ACT>(x, y, P, N, mrow, nbase, g4));
            }
        }
    }
    Y_TRACE_END(3);
}
#undef Y_TRACE_END

int g_group_m = GROUP_M;  // tiles per column group of the tile order (tune key gemm.group_m)
#if APEXMI_GEMM_TRACE
uintptr_t g_gemm_trace = 0;
#endif
int g_large_cfg = 7;  // tiling the auto path picks for large problems (tune key gemm.large)
int g_force_cfg = 0;  // 0 auto, else the tiling number of the header comment
int g_wpacked = 0;    // experiment: W operands are tile-major packed (see GemmGroup::wpacked)
int g_tail_max = 96;   // tune key gemm.tail_max: largest tail problem (in 256x256 tiles) that goes out as its own launch (96 = the text
                       // stream of Flux's FF-up, 512 x 12288: 864 tiles = 3.4 rounds as one launch; 71.5 -> 70.9 ms per step split)
int g_x288 = 0;       // tune key gemm.x288: the 288 x 192 exact-fill tiling — 0 never (SHIPPED: it measured slower, see the kernel's header) |
                      // 1 where it saves a round's worth of tile-work | 2 always (A/B, tests)
int g_x384_group_m = 3; // tune key gemm.x384_group_m: tile-group height of the 384 x 256 launches
int g_x384_dist = -1; // tune key gemm.x384_dist: how a wave's 10 pieces are spread over the 4 phases (0: 3+3+2+2, 1: 2+3+2+3, -1: by launch size)
int g_x384_qkv = 1;   // tune key gemm.x384_qkv: launches with the fused q/k/v epilogue may use the 384 x 256 tiling too (A/B)
int g_x384 = 1;       // tune key gemm.x384: the 384 x 256 tiling — 0 never | 1 where its staged bytes win (x384_pays; SHIPPED) | 2 always (A/B, tests)
int g_small_max = 112; // tune key gemm.small_max: launches of at most this many 256 x 256 tiles go out on the 128 x 128 tiling (0: never)
int g_tail_split = 2; // tune key gemm.tail: a small last problem of a grouped launch goes out on the 128x128 tiling (1: four waves, 2: eight)

// stream-K workspace: 256 slabs of 256 x 256 f32 + 256 flags per (device, stream) — launches on different streams may overlap and
// must not share slabs; allocated (and the flags zeroed) at the first stream-K launch on that stream, kept for the process
int g_streamk = 1;    // tune key gemm.streamk (APEXMI_GEMM_STREAMK builds only): 0 off | 1 launches with 88..232 tiles in their last round | 2 persistent always
struct SkWorkspace {
    float* slab;
    unsigned* flag;
};
static int sk_workspace(hipStream_t stream, SkWorkspace* out) {
    static std::mutex mu;
    static std::vector<std::tuple<int, hipStream_t, SkWorkspace>> all;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    for (auto& e : all)
        if (std::get<0>(e) == dev && std::get<1>(e) == stream) {
            *out = std::get<2>(e);
            return 0;
        }
    SkWorkspace w{nullptr, nullptr};
    if (hipMalloc((void**)&w.slab, (size_t)256 * 65536 * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&w.flag, 256 * sizeof(unsigned)) != hipSuccess || hipMemset(w.flag, 0, 256 * sizeof(unsigned)) != hipSuccess) {
        apexmi_set_error("gemm_bf16: stream-K workspace allocation failed");
        return 1;
    }
    all.emplace_back(dev, stream, w);
    *out = w;
    return 0;
}

template <typename CFG, int EPI>
int launch_cfg(GemmGroup& G, const int* Ms, hipStream_t stream) {
    if constexpr (APEXMI_GEMM_PEEL >= 2 && (CFG::SCHED == 0 || CFG::SCHED == 5)) {
        // the scalar-base pieces carry 32-bit lane offsets inside a tile: 256 rows x leading dimension x 2 bytes must stay below 2^32
        for (int i = 0; i < G.count; ++i)
            if (G.p[i].lda > (1 << 22) || G.p[i].ldw > (1 << 22)) {
                apexmi_set_error("gemm_bf16: leading dimensions above 2^22 elements are not supported (lda %lld, ldw %lld)",
                                 (long long)G.p[i].lda, (long long)G.p[i].ldw);
                return 1;
            }
    }
    static uint64_t attr_set = 0;
    APEXMI_SET_ATTR_ONCE(attr_set,
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<CFG, EPI>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS));
    int t = 0;
    for (int i = 0; i < G.count; ++i) {
        G.p[i].M = Ms[i];
        G.p[i].nm = (Ms[i] + CFG::BM - 1) / CFG::BM;
        G.p[i].nn = (G.p[i].N + CFG::BN - 1) / CFG::BN;
        G.p[i].tile0 = t;
        t += G.p[i].nm * G.p[i].nn;
    }
    G.total = t;
    G.group_m = g_group_m;
    G.clk = apexmi_clk_ptr();
#if APEXMI_GEMM_TRACE
    G.trace = (unsigned long long*)g_gemm_trace;
#endif
    G.wpacked = (CFG::SCHED == 5) ? g_wpacked : 0;
    G.sk_r = 0;
    G.sk_tfull = 0;
    G.sk_slab = nullptr;
    G.sk_flag = nullptr;
    if constexpr (CFG::SCHED == 5 && APEXMI_GEMM_STREAMK) {
        // stream-K when the last round of 256 tiles is part-filled enough to matter and full enough that a tile is shared by at
        // most four workgroups (the owner adds at most three slabs): 11..29 remainder tiles per XCD
        const int r = t % 256;
        const bool forced = g_streamk == 2 && G.batch == 1 && t >= 256 && (r == 0 || (r >= 88 && r <= 232));   // experiment: persistent always
        if (forced || (g_streamk && G.batch == 1 && r >= 88 && r <= 232 && G.K / BK >= 8)) {
            SkWorkspace w;
            if (sk_workspace(stream, &w)) return 1;
            G.sk_r = r;
            G.sk_tfull = t - r;
            G.sk_slab = w.slab;
            G.sk_flag = w.flag;
            hipLaunchKernelGGL((gemm_bf16_kernel<CFG, EPI>), dim3(256, 1), dim3(CFG::NT), CFG::LDS, stream, G);
            return apexmi_check_launch("gemm_bf16");
        }
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<CFG, EPI>), dim3(t, G.batch), dim3(CFG::NT), CFG::LDS, stream, G);
    return apexmi_check_launch("gemm_bf16");
}

template <int EPI>
int launch_x288(GemmGroup& G, const int* Ms, hipStream_t stream) {
    static uint64_t attr_set = 0;
    APEXMI_SET_ATTR_ONCE(attr_set,
        (void)hipFuncSetAttribute((const void*)gemm_bf16_x288_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS));
    int t = 0;
    for (int i = 0; i < G.count; ++i) {
        G.p[i].M = Ms[i];
        G.p[i].nm = (Ms[i] + XBM - 1) / XBM;
        G.p[i].nn = (G.p[i].N + XBN - 1) / XBN;
        G.p[i].tile0 = t;
        t += G.p[i].nm * G.p[i].nn;
    }
    G.total = t;
    G.group_m = g_group_m;
    G.clk = apexmi_clk_ptr();
    G.wpacked = 0;
    G.sk_r = G.sk_tfull = 0;
    G.sk_slab = nullptr;
    G.sk_flag = nullptr;
    hipLaunchKernelGGL((gemm_bf16_x288_kernel<EPI>), dim3(t, 1), dim3(512), X_LDS, stream, G);
    return apexmi_check_launch("gemm_bf16 (288x192)");
}

template <int EPI>
int launch_x384(GemmGroup& G, const int* Ms, hipStream_t stream) {
    static uint64_t attr_set = 0;
    APEXMI_SET_ATTR_ONCE(attr_set,
        (void)hipFuncSetAttribute((const void*)gemm_bf16_x384_kernel<EPI, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, Y_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_x384_kernel<EPI, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, Y_LDS);
        if constexpr (EPI == APEXMI_EPI_BIAS)
            (void)hipFuncSetAttribute((const void*)gemm_bf16_x384_kernel<EPI, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, Y_LDS));
    int t = 0;
    for (int i = 0; i < G.count; ++i) {
        G.p[i].M = Ms[i];
        G.p[i].nm = (Ms[i] + YBM - 1) / YBM;
        G.p[i].nn = (G.p[i].N + YBN - 1) / YBN;
        G.p[i].tile0 = t;
        t += G.p[i].nm * G.p[i].nn;
    }
    G.total = t;
    // tile groups 3 tall: an XCD's 32 concurrent tiles as 3 x 10.7 = 1152 activation rows x 2730 weight rows per K-slice (6 tall: 2304 x
    // 1365) — on the 75 600-row launches +2..4 % (profiles/r05_gemm_x384_group_m.log), the same on the 12-row-tile Flux launch
    G.group_m = g_x384_group_m;
    G.clk = apexmi_clk_ptr();
#if APEXMI_GEMM_TRACE
    G.trace = (unsigned long long*)g_gemm_trace;
#endif
    G.wpacked = 0;
    G.sk_r = G.sk_tfull = 0;
    G.sk_slab = nullptr;
    G.sk_flag = nullptr;
    bool any_qkv = false;
    for (int i = 0; i < G.count; ++i) any_qkv |= G.p[i].qkv != 0;
    if constexpr (EPI == APEXMI_EPI_BIAS) {
        if (any_qkv) {
            hipLaunchKernelGGL((gemm_bf16_x384_kernel<EPI, 1, 1>), dim3(t, 1), dim3(512), Y_LDS, stream, G);
            return apexmi_check_launch("gemm_bf16 (384x256, fused q/k/v)");
        }
    }
    const int dist = g_x384_dist >= 0 ? g_x384_dist : (t >= 2048 ? 0 : 1);     // measured: 3+3+2+2 +1 % on Wan's launches, 2+3+2+3 +1..2 % on Flux's
    if (dist == 0) hipLaunchKernelGGL((gemm_bf16_x384_kernel<EPI, 0>), dim3(t, 1), dim3(512), Y_LDS, stream, G);
    else hipLaunchKernelGGL((gemm_bf16_x384_kernel<EPI, 1>), dim3(t, 1), dim3(512), Y_LDS, stream, G);
    return apexmi_check_launch("gemm_bf16 (384x256)");
}

// Does the 288 x 192 tiling finish this launch in less tile-work than 256 x 256?  Cost = rounds of 256 concurrent tiles x work per
// tile (a 288 x 192 tile stages 11 % more bytes per flop: priced at +4 %); it has to win by 5 %.
inline bool x288_pays(const GemmGroup& G, const int* Ms) {
    int64_t t256 = 0, t288 = 0;
    for (int i = 0; i < G.count; ++i) {
        if (G.p[i].qkv) return false;            // the fused q/k/v epilogue is written for 256-column tiles (two heads)
        t256 += (int64_t)((Ms[i] + 255) / 256) * ((G.p[i].N + 255) / 256);
        t288 += (int64_t)((Ms[i] + XBM - 1) / XBM) * ((G.p[i].N + XBN - 1) / XBN);
    }
    const double c256 = (double)((t256 + 255) / 256) * 65536.0;
    const double c288 = (double)((t288 + 255) / 256) * (XBM * XBN) * 1.04;
    return c288 < 0.95 * c256;
}

// bf16-epilogue launches only (the float-I/O classes of the verification mode stay on 256 x 256; their sums are the same anyway)
inline bool use_x288(const GemmGroup& G, const int* Ms) {
    if (g_x288 == 0 || G.batch != 1 || G.K % BK != 0 || !(g_force_cfg == 0 || g_force_cfg == 7) || g_large_cfg != 7) return false;
    int64_t mtot = 0;
    int nmax = 0;
    for (int i = 0; i < G.count; ++i) {
        if (G.p[i].qkv) return false;
        mtot += Ms[i];
        nmax = G.p[i].N > nmax ? G.p[i].N : nmax;
    }
    if (g_x288 == 2) return true;
    return G.K >= 256 && mtot >= 1024 && nmax >= 1024 && x288_pays(G, Ms);
}

// The 384 x 256 tiling pays where the launch is bound by the chip's aggregate staging rate — several FULL rounds of tiles — and
// loses where tiles are few (a K-tile costs a CU 1.78 us at best against 1.23 us): every problem's row tiles nearly full (<= 1.5 %
// of padding rows), at least three rounds of 256 tiles, and a last round that is full, mostly full, or one of many
// (profiles/r05_gemm_x384_ab*.log: 4608 x 21504 +8..10 %, 4608 x 9216 +9..10 %, Wan's 75 600-row launches +3..7 %; 2.25 rounds -3 %,
// part-filled single rounds -22 %).
inline bool x384_pays(const GemmGroup& G, const int* Ms) {
    int64_t t = 0;
    for (int i = 0; i < G.count; ++i) {
        const int64_t nm = (Ms[i] + YBM - 1) / YBM;
        if ((nm * YBM - Ms[i]) * 200 > 3 * (int64_t)Ms[i]) return false;
        t += nm * ((G.p[i].N + YBN - 1) / YBN);
    }
    const int64_t rem = t % 256;
    return t >= 3 * 256 && (rem == 0 || rem >= 154 || t >= 8 * 256);
}
inline bool use_x384(const GemmGroup& G, const int* Ms) {
    if (g_x384 == 0 || G.batch != 1 || G.K % BK != 0 || !(g_force_cfg == 0 || g_force_cfg == 7) || g_large_cfg != 7) return false;
    for (int i = 0; i < G.count; ++i)
        // (the kernel's piece offsets are 32-bit: 384 rows x leading dimension x 2 bytes must stay below 2^31)
        if (G.p[i].qkv || G.p[i].lda > (1 << 21) || G.p[i].ldw > (1 << 21)) return false;
    return g_x384 == 2 || x384_pays(G, Ms);
}

template <int EPI>
int launch_epi(GemmGroup& G, const int* Ms, hipStream_t stream) {
    int cfg = g_force_cfg;
    if constexpr (EPI == APEXMI_EPI_BIAS || EPI == APEXMI_EPI_BIAS_GATE_RES) {
        if (use_x288(G, Ms)) return launch_x288<EPI>(G, Ms, stream);
        if (use_x384(G, Ms)) return launch_x384<EPI>(G, Ms, stream);
    }
    if (cfg == 0) {
        int64_t mtot = 0;
        int nmax = 0;
        for (int i = 0; i < G.count; ++i) {
            mtot += Ms[i];
            nmax = G.p[i].N > nmax ? G.p[i].N : nmax;
        }
        // large problems: 256x256 tiles, one per CU per round; otherwise the 128x128 tiling
        cfg = (mtot >= 1024 && nmax >= 1024 && G.K >= 256) ? g_large_cfg : 1;
        // A launch that covers well under half of the CUs with 256 x 256 tiles (the 512^2 geometry: proj_out 1536 x 3072 = 72 tiles,
        // FF-down 48 + 24) runs its K-loop on a few CUs at the per-CU floor of ~1.23 us per K-tile while the rest idle; four times
        // as many 128 x 128 tiles (two workgroups per CU) finish sooner until ~118 tiles (profiles/r05_gemm_small_ab.log: 72 tiles
        // K 15360 295 -> 213 us, 48 tiles K 12288 237 -> 160 us, 96 tiles 238 -> 192 us; 144 tiles 236 vs 342 us: stays).
        // Gate / residual launches only (attention-out, FF-down, proj_out — the part-filled ones of the MM-DiT blocks): the
        // bias-class projections keep the tiling whose sums the fused q/k/v epilogue reproduces bit for bit.
        if (EPI == APEXMI_EPI_BIAS_GATE_RES && cfg == 7 && G.batch == 1 && g_small_max > 0) {
            int64_t t256 = 0;
            bool any_qkv = false;
            for (int i = 0; i < G.count; ++i) {
                t256 += (int64_t)((Ms[i] + 255) / 256) * ((G.p[i].N + 255) / 256);
                any_qkv |= G.p[i].qkv != 0;
            }
            if (!any_qkv && t256 <= g_small_max) cfg = 1;
        }
        // with a 24 KiB row stride (K = 12288, the MLP down-projection) the one-wave-per-SIMD kernel loses
        // 17-23 % to channel aliasing that the ping-pong schedules do not see
        if (cfg == 6 && G.K % 12288 == 0) cfg = 7;
        // A grouped launch whose leading problems fill whole rounds of 256 tiles and whose LAST problem is a small tail
        // (QwenImage's feed-forward up-projection: image stream 32 x 48 = 1536 tiles = 6 rounds, text stream 48 tiles):
        // a partial round costs about half a full one however few tiles it holds (tools/gemm_rounds.py: +66 us for 48
        // tiles on a 577 us launch), so the tail problem goes out as its own 128x128-tiled launch (192 quarter-size
        // tiles, under one round of the two-workgroups-per-CU kernel).  Round 3: with the eight-wave 128x128 tiling the same holds
        // for Flux's FF-up (3 x 256 image tiles + 96 text tiles): g_tail_max 64 -> 96.
        if ((cfg == 7 || cfg == 9 || cfg == 10) && g_tail_split && G.count >= 2 && G.batch == 1) {
            int64_t lead = 0;
            for (int i = 0; i + 1 < G.count; ++i)
                lead += (int64_t)((Ms[i] + 255) / 256) * ((G.p[i].N + 255) / 256);
            const int li = G.count - 1;
            const int64_t last = (int64_t)((Ms[li] + 255) / 256) * ((G.p[li].N + 255) / 256);
            if (lead >= 256 && lead % 256 == 0 && last <= g_tail_max) {
                GemmGroup T = G;
                T.count = 1;
                T.p[0] = G.p[li];
                G.count = li;
                if (int rc = (cfg == 9    ? launch_cfg<CFG_256R, EPI>(G, Ms, stream)
                              : cfg == 10 ? launch_cfg<CFG_256R5, EPI>(G, Ms, stream)
                                          : launch_cfg<CFG_256P16, EPI>(G, Ms, stream)))
                    return rc;
                // the tail launch is under one workgroup per CU, i.e. a latency chain per K-tile whose length is the LDS-DMA
                // pieces a wave issues (~150 cycles each): eight waves per 128x128 tile issue 4 per K-tile, four waves 8
                if (g_tail_split == 2) return launch_cfg<CFG_128E, EPI>(T, Ms + li, stream);
                return launch_cfg<CFG_128, EPI>(T, Ms + li, stream);
            }
        }
    }
    switch (cfg) {
        case 1: return launch_cfg<CFG_128, EPI>(G, Ms, stream);
        case 2: return launch_cfg<CFG_256, EPI>(G, Ms, stream);
        case 8: return launch_cfg<CFG_128E, EPI>(G, Ms, stream);
        case 6: return launch_cfg<CFG_256W, EPI>(G, Ms, stream);
        case 7: return launch_cfg<CFG_256P16, EPI>(G, Ms, stream);
        case 9: return launch_cfg<CFG_256R, EPI>(G, Ms, stream);
        case 10: return launch_cfg<CFG_256R5, EPI>(G, Ms, stream);
        default:
            return launch_cfg<CFG_256P, EPI>(G, Ms, stream);
    }
}

int launch_group(GemmGroup& G, const int* Ms, int kind, hipStream_t stream) {
    switch (kind) {  // epilogue class: bias (+gelu flag) | gate/residual | f32 output
        case APEXMI_EPI_BIAS_GATE_RES: return launch_epi<APEXMI_EPI_BIAS_GATE_RES>(G, Ms, stream);
        case APEXMI_EPI_BIAS_F32: return launch_epi<APEXMI_EPI_BIAS_F32>(G, Ms, stream);
        case EPI_GATE_RES_F32: return launch_epi<EPI_GATE_RES_F32>(G, Ms, stream);
        default: return launch_epi<APEXMI_EPI_BIAS>(G, Ms, stream);
    }
}

// epilogue argument -> kernel class.  APEXMI_EPI_F32_IO (C and R are float) maps the bf16 classes onto the float ones.
int epi_base(int epilogue) { return epilogue & ~APEXMI_EPI_F32_IO; }
bool epi_f32(int epilogue) { return (epilogue & APEXMI_EPI_F32_IO) != 0 || epi_base(epilogue) == APEXMI_EPI_BIAS_F32; }
int epi_kind(int epilogue) {
    const int b = epi_base(epilogue);
    if (epilogue & APEXMI_EPI_F32_IO) return b == APEXMI_EPI_BIAS_GATE_RES ? EPI_GATE_RES_F32 : APEXMI_EPI_BIAS_F32;
    return act_mode(b) ? APEXMI_EPI_BIAS : b;
}

int check_problem(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int M,
                  int N, int K, int epilogue, const float* gate, const void* R, int64_t ldr) {
    APEXMI_REQUIRE(A && W && C, "gemm_bf16: null operand");
    APEXMI_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_bf16: empty problem M=%d N=%d K=%d", M, N, K);
    APEXMI_REQUIRE(K % BK == 0, "gemm_bf16: K=%d must be a multiple of %d", K, BK);
    APEXMI_REQUIRE(N % 8 == 0, "gemm_bf16: N=%d must be a multiple of 8", N);
    APEXMI_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0,
                   "gemm_bf16: leading dimensions must keep rows 16-byte aligned");
    APEXMI_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)C % 8) == 0,
                   "gemm_bf16: operands must be 16-byte aligned");
    APEXMI_REQUIRE(epi_base(epilogue) >= 0 && epi_base(epilogue) <= 6 && (epilogue & ~(APEXMI_EPI_F32_IO | 7)) == 0,
                   "gemm_bf16: unknown epilogue %d", epilogue);
    APEXMI_REQUIRE(!epi_f32(epilogue) || ((uintptr_t)C % 16) == 0, "gemm_bf16: f32 output must be 16-byte aligned");
    if (epi_base(epilogue) == APEXMI_EPI_BIAS_GATE_RES) {
        APEXMI_REQUIRE(gate && R, "gemm_bf16: gate/residual epilogue needs gate and R");
        APEXMI_REQUIRE(ldr % 4 == 0 && ((uintptr_t)gate % 16) == 0 && ((uintptr_t)R % (epi_f32(epilogue) ? 16 : 8)) == 0,
                       "gemm_bf16: gate/R alignment");
    }
    return 0;
}

}  // namespace

extern "C" int apexmi_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw,
                                const void* bias, void* C, int64_t ldc, int M, int N, int K,
                                int epilogue, const float* gate, const void* R, int64_t ldr,
                                apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = check_problem(A, lda, W, ldw, C, ldc, M, N, K, epilogue, gate, R, ldr)) return rc;
    GemmGroup G;
    G.count = 1;
    G.K = K;
    G.batch = 1;
    G.bsA = G.bsW = G.bsC_bytes = 0;
    G.p[0] = GemmProblem{(const bf16_t*)A, (const bf16_t*)W, (const bf16_t*)bias, (bf16_t*)C, gate,
                         (const bf16_t*)R, lda, ldw, ldc, ldr, M, N, 0, 0, 0,
                         act_mode(epi_base(epilogue))};
    ApexmiProfScope prof(0, stream, 2.0 * M * N * (double)K,
                         2.0 * ((double)M * K + (double)N * K + (double)M * N));
    return launch_group(G, &M, epi_kind(epilogue), stream);
}

extern "C" int apexmi_gemm_bf16_batched(const void* A, int64_t lda, int64_t stride_a, const void* W, int64_t ldw,
                                        int64_t stride_w, void* C, int64_t ldc, int64_t stride_c, int batch, int M,
                                        int N, int K, int epilogue, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = check_problem(A, lda, W, ldw, C, ldc, M, N, K, epilogue, nullptr, nullptr, 0)) return rc;
    APEXMI_REQUIRE(epilogue == APEXMI_EPI_BIAS || epilogue == APEXMI_EPI_BIAS_F32,
                   "gemm_bf16_batched: epilogue %d unsupported (plain bf16 or f32 output only)", epilogue);
    APEXMI_REQUIRE(batch >= 1 && batch <= 65535, "gemm_bf16_batched: batch=%d not in [1, 65535]", batch);
    const int esz = epilogue == APEXMI_EPI_BIAS_F32 ? 4 : 2;
    APEXMI_REQUIRE(stride_a % 8 == 0 && stride_w % 8 == 0 && (stride_c * esz) % 16 == 0,
                   "gemm_bf16_batched: batch strides must keep every element 16-byte aligned");
    APEXMI_REQUIRE(epilogue != APEXMI_EPI_BIAS_F32 || ((uintptr_t)C % 16) == 0, "gemm_bf16_batched: f32 output must be 16-byte aligned");
    GemmGroup G;
    G.count = 1;
    G.K = K;
    G.batch = batch;
    G.bsA = stride_a;
    G.bsW = stride_w;
    G.bsC_bytes = stride_c * esz;
    G.p[0] = GemmProblem{(const bf16_t*)A, (const bf16_t*)W, nullptr, (bf16_t*)C, nullptr, nullptr, lda, ldw, ldc, 0,
                         M, N, 0, 0, 0, 0};
    ApexmiProfScope prof(0, stream, 2.0 * batch * M * N * (double)K,
                         2.0 * batch * ((double)M * K + (double)N * K + (double)M * N));
    return launch_group(G, &M, epi_kind(epilogue), stream);
}

extern "C" int apexmi_gemm_bf16_grouped(int count, const void* const* A, const int64_t* lda,
                                        const void* const* W, const int64_t* ldw,
                                        const void* const* bias, void* const* C, const int64_t* ldc,
                                        const int* M, const int* N, int K, const int* epilogue,
                                        const float* const* gate, const void* const* R,
                                        const int64_t* ldr, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(count >= 1 && count <= MAX_GROUPS, "gemm_bf16_grouped: count=%d not in [1,%d]", count, MAX_GROUPS);
    GemmGroup G;
    G.count = count;
    G.K = K;
    G.batch = 1;
    G.bsA = G.bsW = G.bsC_bytes = 0;
    double flops = 0, bytes = 0;
    const int kind = epi_kind(epilogue[0]);
    for (int i = 0; i < count; ++i) {
        APEXMI_REQUIRE(epi_kind(epilogue[i]) == kind,
                       "gemm_bf16_grouped: gate/residual (or float / bf16 output) problems cannot be mixed with the others");
        const float* g = gate ? gate[i] : nullptr;
        const void* r = R ? R[i] : nullptr;
        const int64_t lr = ldr ? ldr[i] : 0;
        if (int rc = check_problem(A[i], lda[i], W[i], ldw[i], C[i], ldc[i], M[i], N[i], K, epilogue[i], g, r, lr))
            return rc;
        G.p[i] = GemmProblem{(const bf16_t*)A[i], (const bf16_t*)W[i], (const bf16_t*)(bias ? bias[i] : nullptr),
                             (bf16_t*)C[i], g, (const bf16_t*)r, lda[i], ldw[i], ldc[i], lr, M[i], N[i], 0, 0, 0,
                             act_mode(epi_base(epilogue[i]))};
        flops += 2.0 * M[i] * N[i] * (double)K;
        bytes += 2.0 * ((double)M[i] * K + (double)N[i] * K + (double)M[i] * N[i]);
    }
    ApexmiProfScope prof(0, stream, flops, bytes);
    return launch_group(G, M, kind, stream);
}

extern "C" int apexmi_gemm_uses_x288(int M, int N, int K) {
    GemmGroup G;
    memset(&G, 0, sizeof(G));
    G.count = 1;
    G.K = K;
    G.batch = 1;
    G.p[0].N = N;
    return (M > 0 && N > 0 && K > 0 && use_x288(G, &M)) ? 1 : 0;
}

extern "C" int apexmi_gemm_qkv_fusable(int64_t m_total, int n_max, int K) {
    return (g_force_cfg == 0 || g_force_cfg == 7 || g_force_cfg == 9 || g_force_cfg == 10) &&
           (g_large_cfg == 7 || g_large_cfg == 9 || g_large_cfg == 10) && m_total >= 1024 && n_max >= 1024 && K >= 256 && K % BK == 0;
}

extern "C" int apexmi_gemm_bf16_grouped_qkv(int count, const void* const* A, const int64_t* lda, const void* const* W,
                                            const int64_t* ldw, const void* const* bias, void* const* C, const int64_t* ldc,
                                            const int* M, const int* N, int K, const int* epilogue, const int* is_qkv,
                                            const void* const* norm_q, const void* const* norm_k, const int* row0, int H,
                                            float eps, const float* rope, void* q_out, void* k_out, void* vt_out, int S_out,
                                            int Skp, apexmi_stream_t stream_) {
    return apexmi_gemm_bf16_grouped_qkv_pairs(count, A, lda, W, ldw, bias, C, ldc, M, N, K, epilogue, is_qkv, norm_q, norm_k, row0, H,
                                              eps, rope, nullptr, q_out, k_out, vt_out, S_out, Skp, stream_);
}

extern "C" int apexmi_gemm_bf16_grouped_qkv_pairs(int count, const void* const* A, const int64_t* lda, const void* const* W,
                                                  const int64_t* ldw, const void* const* bias, void* const* C, const int64_t* ldc,
                                                  const int* M, const int* N, int K, const int* epilogue, const int* is_qkv,
                                                  const void* const* norm_q, const void* const* norm_k, const int* row0, int H,
                                                  float eps, const float* rope, const float* rope_pairs, void* q_out, void* k_out,
                                                  void* vt_out, int S_out, int Skp, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(((uintptr_t)rope_pairs % 16) == 0, "gemm_bf16_grouped_qkv: rope_pairs must be 16-byte aligned");
    APEXMI_REQUIRE(count >= 1 && count <= MAX_GROUPS, "gemm_bf16_grouped_qkv: count=%d not in [1,%d]", count, MAX_GROUPS);
    APEXMI_REQUIRE(is_qkv && row0 && rope && q_out && k_out && vt_out && H > 0, "gemm_bf16_grouped_qkv: null argument");
    APEXMI_REQUIRE(H % 2 == 0 && S_out > 0 && Skp >= S_out && Skp % 8 == 0, "gemm_bf16_grouped_qkv: H=%d must be even, Skp=%d a multiple of 8 >= S_out=%d",
                   H, Skp, S_out);
    APEXMI_REQUIRE(((uintptr_t)rope % 16) == 0 && ((uintptr_t)q_out % 16) == 0 && ((uintptr_t)k_out % 16) == 0 && ((uintptr_t)vt_out % 16) == 0,
                   "gemm_bf16_grouped_qkv: rope / outputs must be 16-byte aligned");
    GemmGroup G;
    G.count = count;
    G.K = K;
    G.batch = 1;
    G.bsA = G.bsW = G.bsC_bytes = 0;
    G.qs = QkvShared{(bf16_t*)q_out, (bf16_t*)k_out, (bf16_t*)vt_out, rope, S_out, Skp, H * 128, eps, rope_pairs};
    double flops = 0, bytes = 0;
    int64_t mtot = 0;
    int nmax = 0;
    for (int i = 0; i < count; ++i) {
        APEXMI_REQUIRE(epi_kind(epilogue[i]) == APEXMI_EPI_BIAS, "gemm_bf16_grouped_qkv: bias-class epilogues only (got %d)", epilogue[i]);
        void* c = is_qkv[i] ? q_out : C[i];               // a fused problem writes no C
        const int64_t lc = is_qkv[i] ? 128 : ldc[i];
        if (int rc = check_problem(A[i], lda[i], W[i], ldw[i], c, lc, M[i], N[i], K, epilogue[i], nullptr, nullptr, 0)) return rc;
        G.p[i] = GemmProblem{(const bf16_t*)A[i], (const bf16_t*)W[i], (const bf16_t*)(bias ? bias[i] : nullptr), (bf16_t*)c,
                             nullptr, nullptr, lda[i], ldw[i], lc, 0, M[i], N[i], 0, 0, 0, act_mode(epi_base(epilogue[i]))};
        if (is_qkv[i]) {
            APEXMI_REQUIRE(N[i] == 3 * H * 128 && epi_base(epilogue[i]) == APEXMI_EPI_BIAS,
                           "gemm_bf16_grouped_qkv: a fused problem has N = 3 H 128 = %d (got %d) and the plain bias epilogue", 3 * H * 128, N[i]);
            APEXMI_REQUIRE(row0[i] >= 0 && row0[i] + M[i] <= S_out,
                           "gemm_bf16_grouped_qkv: rows [%d, %d) must lie inside S_out=%d", row0[i], row0[i] + M[i], S_out);
            const void* nq = norm_q ? norm_q[i] : nullptr;
            const void* nk = norm_k ? norm_k[i] : nullptr;
            APEXMI_REQUIRE(((uintptr_t)nq % 16) == 0 && ((uintptr_t)nk % 16) == 0, "gemm_bf16_grouped_qkv: norm weights must be 16-byte aligned");
            G.p[i].qkv = 1;
            G.p[i].row0 = row0[i];
            G.p[i].nq = (const bf16_t*)nq;
            G.p[i].nk = (const bf16_t*)nk;
        }
        mtot += M[i];
        nmax = N[i] > nmax ? N[i] : nmax;
        flops += 2.0 * M[i] * N[i] * (double)K;
        bytes += 2.0 * ((double)M[i] * K + (double)N[i] * K + (double)M[i] * N[i]);
    }
    // the fused epilogue exists on the shipped 256x256 / 16x16x32 tiling only: the caller keeps the separate pass otherwise
    APEXMI_REQUIRE(apexmi_gemm_qkv_fusable(mtot, nmax, K),
                   "gemm_bf16_grouped_qkv: needs the 256x256 v_mfma_f32_16x16x32 tiling (gemm.config / gemm.large = 7, >= 1024 rows)");
    ApexmiProfScope prof(0, stream, flops, bytes);
    // the single block's QKV + MLP-up launch (M 4608 = 12 x 384: 1008 tiles of 384 x 256 = 3.9 rounds instead of 1512 = 5.9)
    bool ld_ok = true;
    for (int i = 0; i < count; ++i) ld_ok &= lda[i] <= (1 << 21) && ldw[i] <= (1 << 21);
    if (g_x384 && g_x384_qkv && ld_ok && g_large_cfg == 7 && (g_force_cfg == 0 || g_force_cfg == 7) && (g_x384 == 2 || x384_pays(G, M)))
        return launch_x384<APEXMI_EPI_BIAS>(G, M, stream);
    if (g_large_cfg == 9 || g_force_cfg == 9) return launch_cfg<CFG_256R, APEXMI_EPI_BIAS>(G, M, stream);
    if (g_large_cfg == 10 || g_force_cfg == 10) return launch_cfg<CFG_256R5, APEXMI_EPI_BIAS>(G, M, stream);
    return launch_cfg<CFG_256P16, APEXMI_EPI_BIAS>(G, M, stream);
}

extern "C" int apexmi_split_bf16x3(const float* x, int64_t ldx, int64_t M, int K, void* out, int64_t ldo,
                                   apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(x && out && M > 0 && K > 0, "split_bf16x3: bad arguments");
    APEXMI_REQUIRE(K % 8 == 0 && ldx % 4 == 0 && ldo % 8 == 0 && ldo >= 3 * (int64_t)K && ((uintptr_t)x % 16) == 0 &&
                       ((uintptr_t)out % 16) == 0,
                   "split_bf16x3: K=%d must be a multiple of 8 and rows 16-byte aligned", K);
    const int64_t n = M * (K / 8);
    ApexmiProfScope prof(5, stream, 0.0, 10.0 * (double)M * K);
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, ldx, M, K,
                       (bf16_t*)out, ldo);
    return apexmi_check_launch("split_bf16x3");
}

// tuning keys of this file (dispatched from apexmi_tune_set, runtime.hip)
int apexmi_set_gemm_key(const char* key, int value) {
    if (!strcmp(key, "gemm.tail")) g_tail_split = value;
    else if (!strcmp(key, "gemm.tail_max")) g_tail_max = value;
    else if (!strcmp(key, "gemm.group_m")) g_group_m = value > 0 ? value : GROUP_M;
    else if (!strcmp(key, "gemm.large")) g_large_cfg = value;
    else if (!strcmp(key, "gemm.config")) g_force_cfg = value;
    else if (!strcmp(key, "gemm.wpacked")) g_wpacked = value;
    else if (!strcmp(key, "gemm.x288")) g_x288 = value;
    else if (!strcmp(key, "gemm.small_max")) g_small_max = value;
    else if (!strcmp(key, "gemm.x384")) g_x384 = value;
    else if (!strcmp(key, "gemm.x384_dist")) g_x384_dist = value;
    else if (!strcmp(key, "gemm.x384_group_m")) g_x384_group_m = value > 0 ? value : 3;
    else if (!strcmp(key, "gemm.x384_qkv")) g_x384_qkv = value;
#if APEXMI_GEMM_TRACE
    else if (!strcmp(key, "gemm.trace_lo")) g_gemm_trace = (g_gemm_trace & ~(uintptr_t)0xffffffffu) | (uint32_t)value;
    else if (!strcmp(key, "gemm.trace_hi")) g_gemm_trace = (g_gemm_trace & (uintptr_t)0xffffffffu) | ((uintptr_t)(uint32_t)value << 32);
#endif
    else if (!strcmp(key, "gemm.streamk")) g_streamk = value;
    else return 1;
    return 0;
}
