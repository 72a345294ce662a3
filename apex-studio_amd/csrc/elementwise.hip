// HBM-bound kernels of the denoise path: AdaLN-style modulated LayerNorm / RMSNorm, the
// q/k RMSNorm + RoPE + layout pass, V transpose, conditioning GEMV, timestep embedding, casts and
// the Euler scheduler axpy.  All of them move 16 bytes per lane and keep statistics in f32.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// out = LN(x) [* gamma + beta] [* (1 + scale) + shift]   or   RMSNorm(x) * gamma
// One wave per row, the whole row in registers (C <= 8192), two-pass statistics.
// Reference: AdaLayerNormZero / Single / Continuous, flux model.py:299-302,
// wan model.py:56-165 (FP32LayerNorm + scale/shift), efficiency/mod.py:24-35 (RMSNorm).
// ------------------------------------------------------------------------------------------------
constexpr int LN_MAX_C = 8192;

// One 256-thread workgroup per row; NIT = 16-byte chunks per thread (ceil(C / 2048)).  A row per
// workgroup (instead of a row per wave) puts 4x more waves on the chip for the 4608-row Flux buffers,
// which is what an HBM-latency-bound pass needs; the two statistics cost two LDS reductions.
APEXMI_DEVICE float block_sum_256(float x, float* red) {
    x = wave_sum(x);
    const int w = threadIdx.x >> 6;
    __syncthreads();  // protect `red` from the previous reduction's readers
    if ((threadIdx.x & 63) == 0) red[w] = x;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

template <typename T, int NIT>
__global__ __launch_bounds__(256) void ln_modulate_kernel(
    const T* __restrict__ x, int64_t ldx, T* __restrict__ out, int64_t ldo, int M, int C,
    const float* __restrict__ scale, const float* __restrict__ shift,
    const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta, float eps, int rms, int split,
    const float* __restrict__ scale2, const float* __restrict__ shift2) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    if (row < split) {  // rows [0, split) take the second modulation set (text stream of a joint buffer)
        scale = scale2;
        shift = shift2;
    }
    const int nchunk = C >> 3;
    const T* xp = x + (int64_t)row * ldx;
    float v[NIT][8];
    float sum = 0.0f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 256 + threadIdx.x;
        if (c < nchunk) {
            load8<T>(xp + c * 8, v[it]);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[it][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[it][j] = 0.0f;
        }
    }
    float mean = 0.0f;
    if (!rms) mean = block_sum_256(sum, red) / (float)C;
    float sq = 0.0f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 256 + threadIdx.x;
        if (c < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[it][j] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(block_sum_256(sq, red) / (float)C + eps);
    T* op = out + (int64_t)row * ldo;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 256 + threadIdx.x;
        if (c < nchunk) {
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = (v[it][j] - mean) * rstd;
            if (gamma != nullptr) {
                float g[8];
                unpack8(*(const u32x4*)(gamma + c * 8), g);
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] *= g[j];
            }
            if (beta != nullptr) {
                float bt[8];
                unpack8(*(const u32x4*)(beta + c * 8), bt);
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] += bt[j];
            }
            if (scale != nullptr) {
                const f32x4 s0 = *(const f32x4*)(scale + c * 8);
                const f32x4 s1 = *(const f32x4*)(scale + c * 8 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    y[j] *= 1.0f + s0[j];
                    y[j + 4] *= 1.0f + s1[j];
                }
            }
            if (shift != nullptr) {
                const f32x4 s0 = *(const f32x4*)(shift + c * 8);
                const f32x4 s1 = *(const f32x4*)(shift + c * 8 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    y[j] += s0[j];
                    y[j + 4] += s1[j];
                }
            }
            store8<T>(op + c * 8, y);
        }
    }
}

// Wave-per-row variant for C = 64 * 8 * NCH (3072, 3584, 5120): a lane owns NCH 16-byte chunks, the two
// statistics are wave reductions (no LDS, no barrier), four rows per workgroup.  Same arithmetic order per lane
// as the block kernel's per-thread part; the cross-lane sums differ in shape, both are f32.
template <typename T, int NCH, int NR>
__global__ __launch_bounds__(256) void ln_modulate_wave_kernel(
    const T* __restrict__ x, int64_t ldx, T* __restrict__ out, int64_t ldo, int M, int C,
    const float* __restrict__ scale, const float* __restrict__ shift,
    const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta, float eps, int rms, int split,
    const float* __restrict__ scale2, const float* __restrict__ shift2) {
    // NR rows per wave (`ln.wave` = NR): all NR rows' loads are issued before the first reduction, so a wave has NR x NCH
    // 16-byte loads in flight and the second row's statistics / stores overlap the first row's store drain.  Per row the
    // arithmetic is the same for every NR (bit-identical outputs).
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * NR;
    if (row0 >= M) return;
    float v[NR][NCH][8];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int row = min(row0 + r, M - 1);
        const T* xp = x + (int64_t)row * ldx;
#pragma unroll
        for (int it = 0; it < NCH; ++it) load8<T>(xp + (it * 64 + lane) * 8, v[r][it]);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int row = row0 + r;
        if (row >= M) break;
        const float* sc = row < split ? scale2 : scale;
        const float* sh = row < split ? shift2 : shift;
        float sum = 0.0f;
#pragma unroll
        for (int it = 0; it < NCH; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[r][it][j];
        const float mean = rms ? 0.0f : wave_sum(sum) / (float)C;
        float sq = 0.0f;
#pragma unroll
        for (int it = 0; it < NCH; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[r][it][j] - mean;
                sq += d * d;
            }
        const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
        T* op = out + (int64_t)row * ldo;
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            const int c = it * 64 + lane;
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = (v[r][it][j] - mean) * rstd;
            if (gamma != nullptr) {
                float g[8];
                unpack8(*(const u32x4*)(gamma + c * 8), g);
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] *= g[j];
            }
            if (beta != nullptr) {
                float bt[8];
                unpack8(*(const u32x4*)(beta + c * 8), bt);
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] += bt[j];
            }
            if (sc != nullptr) {
                const f32x4 s0 = *(const f32x4*)(sc + c * 8);
                const f32x4 s1 = *(const f32x4*)(sc + c * 8 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    y[j] *= 1.0f + s0[j];
                    y[j + 4] *= 1.0f + s1[j];
                }
            }
            if (sh != nullptr) {
                const f32x4 s0 = *(const f32x4*)(sh + c * 8);
                const f32x4 s1 = *(const f32x4*)(sh + c * 8 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    y[j] += s0[j];
                    y[j + 4] += s1[j];
                }
            }
            store8<T>(op + c * 8, y);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// q/k: per-head RMSNorm (f32) * weight, rotary embedding, write [H, S_out, 128].
// 16 lanes per (row, which, head) unit, 8 elements (4 rotary pairs) per lane.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(
    const T* __restrict__ q, const T* __restrict__ k, int64_t ld_in, int S, int H,
    int split, const bf16_t* __restrict__ wq, const bf16_t* __restrict__ wk,
    const bf16_t* __restrict__ wq2, const bf16_t* __restrict__ wk2, float eps,
    const float* __restrict__ rope, int rope_mode, T* __restrict__ qo,
    T* __restrict__ ko, int S_out, int row0) {
    constexpr int D = 128;
    const int64_t unit = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l16 = threadIdx.x & 15;
    const int nk = (k != nullptr) ? 2 : 1;  // q only (cross-attention query / key sides are separate calls)
    const int64_t nunit = (int64_t)S * nk * H;
    const bool live = unit < nunit;
    const int64_t u = live ? unit : nunit - 1;
    const int s = (int)(u / (nk * H));
    const int rem = (int)(u % (nk * H));
    const int which = rem / H, h = rem % H;
    const int d = l16 * 8;
    const T* src = (which ? k : q) + (int64_t)s * ld_in + h * D + d;
    float x[8];
    load8<T>(src, x);
    const bf16_t* w = which ? (s < split ? wk2 : wk) : (s < split ? wq2 : wq);
    if (w != nullptr) {
        float sq = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sq += x[j] * x[j];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
        const float r = rsqrtf(sq * (1.0f / D) + eps);
        float wv[8];
        unpack8(*(const u32x4*)(w + d), wv);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = x[j] * r * wv[j];
    }
    const int srow = row0 + s;
    float y[8];
    if (rope_mode == APEXMI_ROPE_INTERLEAVED) {
        const float* cp = rope + (int64_t)srow * D + d;
        const float* sp = rope + (int64_t)S_out * D + (int64_t)srow * D + d;
        const f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4);
        const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
        float cs[8], sn[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cs[j] = c0[j];
            cs[j + 4] = c1[j];
            sn[j] = s0[j];
            sn[j + 4] = s1[j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            y[2 * i] = fmaf(x[2 * i], cs[2 * i], -(x[2 * i + 1] * sn[2 * i]));
            y[2 * i + 1] = fmaf(x[2 * i + 1], cs[2 * i + 1], x[2 * i] * sn[2 * i + 1]);
        }
    } else if (rope_mode == APEXMI_ROPE_COMPLEX) {
        const float* tp = rope + ((int64_t)srow * (D / 2) + l16 * 4) * 2;
        const f32x4 t0 = *(const f32x4*)tp, t1 = *(const f32x4*)(tp + 4);
        const float cs[4] = {t0[0], t0[2], t1[0], t1[2]};
        const float sn[4] = {t0[1], t0[3], t1[1], t1[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            y[2 * i] = fmaf(x[2 * i], cs[i], -(x[2 * i + 1] * sn[i]));
            y[2 * i + 1] = fmaf(x[2 * i], sn[i], x[2 * i + 1] * cs[i]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = x[j];
    }
    if (live) {
        T* dst = (which ? ko : qo) + ((int64_t)h * S_out + srow) * D + d;
        store8<T>(dst, y);
    }
}

// Same arithmetic, four heads per 16-lane group: the four 16-byte loads are issued together (4x the bytes in flight per
// wave: the one-head kernel is latency-bound at ~3.9 TB/s) and the rope table row and norm weights are fetched once for
// the four heads.  H % 4 == 0.
template <typename T>
APEXMI_DEVICE void qk_norm_rope4_body(
    int bx, const T* __restrict__ q, const T* __restrict__ k, int64_t ld_in, int S, int H,
    int split, const bf16_t* __restrict__ wq, const bf16_t* __restrict__ wk,
    const bf16_t* __restrict__ wq2, const bf16_t* __restrict__ wk2, float eps,
    const float* __restrict__ rope, int rope_mode, T* __restrict__ qo,
    T* __restrict__ ko, int S_out, int row0) {
    constexpr int D = 128, G = 4;
    const int64_t grp = (int64_t)bx * 16 + (threadIdx.x >> 4);
    const int l16 = threadIdx.x & 15;
    const int nk = (k != nullptr) ? 2 : 1;
    const int gpr = nk * H / G;                       // groups per row
    const int64_t ngrp = (int64_t)S * gpr;
    const bool live = grp < ngrp;
    const int64_t u = live ? grp : ngrp - 1;
    const int s = (int)(u / gpr);
    const int rem = (int)(u % gpr);
    const int which = rem / (H / G), h0 = (rem % (H / G)) * G;
    const int d = l16 * 8;
    const T* src = (which ? k : q) + (int64_t)s * ld_in + h0 * D + d;
    Raw8<T> raw[G];
#pragma unroll
    for (int i = 0; i < G; ++i) raw[i] = ldraw8(src + i * D);
    const bf16_t* w = which ? (s < split ? wk2 : wk) : (s < split ? wq2 : wq);
    float wv[8];
    if (w != nullptr) unpack8(*(const u32x4*)(w + d), wv);
    const int srow = row0 + s;
    float cs[8], sn[8];
    if (rope_mode == APEXMI_ROPE_INTERLEAVED) {
        const float* cp = rope + (int64_t)srow * D + d;
        const float* sp = rope + (int64_t)S_out * D + (int64_t)srow * D + d;
        const f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4);
        const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cs[j] = c0[j];
            cs[j + 4] = c1[j];
            sn[j] = s0[j];
            sn[j + 4] = s1[j];
        }
    } else if (rope_mode == APEXMI_ROPE_COMPLEX) {
        const float* tp = rope + ((int64_t)srow * (D / 2) + l16 * 4) * 2;
        const f32x4 t0 = *(const f32x4*)tp, t1 = *(const f32x4*)(tp + 4);
        cs[0] = t0[0]; cs[1] = t0[2]; cs[2] = t1[0]; cs[3] = t1[2];
        sn[0] = t0[1]; sn[1] = t0[3]; sn[2] = t1[1]; sn[3] = t1[3];
    }
#pragma unroll
    for (int i = 0; i < G; ++i) {
        float x[8], y[8];
        unraw8(raw[i], x);
        if (w != nullptr) {
            float sq = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sq += x[j] * x[j];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
            const float r = rsqrtf(sq * (1.0f / D) + eps);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = x[j] * r * wv[j];
        }
        if (rope_mode == APEXMI_ROPE_INTERLEAVED) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                y[2 * p] = fmaf(x[2 * p], cs[2 * p], -(x[2 * p + 1] * sn[2 * p]));
                y[2 * p + 1] = fmaf(x[2 * p + 1], cs[2 * p + 1], x[2 * p] * sn[2 * p + 1]);
            }
        } else if (rope_mode == APEXMI_ROPE_COMPLEX) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                y[2 * p] = fmaf(x[2 * p], cs[p], -(x[2 * p + 1] * sn[p]));
                y[2 * p + 1] = fmaf(x[2 * p], sn[p], x[2 * p + 1] * cs[p]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = x[j];
        }
        if (live) store8<T>((which ? ko : qo) + ((int64_t)(h0 + i) * S_out + srow) * D + d, y);
    }
}

// ------------------------------------------------------------------------------------------------
// vt[h][d][col0 + s] = v[s][h][d]; columns in [S, round_up(S, 64)) are zero-filled (the attention
// kernel multiplies them by p = 0, so they must be finite).  64 x 128 tile through LDS.
// ------------------------------------------------------------------------------------------------
template <typename T>
APEXMI_DEVICE void v_transpose_body(int bx, int by, const T* __restrict__ v, int64_t v_sh, int64_t v_ss, int S,
                                    int D, T* __restrict__ vt, int Skp, int col0) {
    // Row r = 8 sc + j of the tile is stored ROTATED by 8 sc elements (16 bytes x sc): in the transposed read the 8
    // lanes of one 128-byte output segment (sc = 0..7, same j, same d) then hit 8 different bank groups instead of
    // one (row stride 64 dwords = 0 mod 32 banks; PMC: SQ_LDS_BANK_CONFLICT was 80 % of SQ_LDS_IDX_ACTIVE before).
    constexpr int LDW = 128;
    __shared__ __attribute__((aligned(16))) T tile[64 * LDW];
    const int tid = threadIdx.x;
    const int s0 = bx * 64, h = by;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = i * 256 + tid;
        const int r = idx >> 4, dc = idx & 15;
        Raw8<T> val = zero_raw8<T>();
        if (s0 + r < S) val = ldraw8(v + (int64_t)h * v_sh + (int64_t)(s0 + r) * v_ss + dc * 8);
        straw8(tile + r * LDW + (((dc + (r >> 3)) & 15) << 3), val);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = i * 256 + tid;
        const int d = idx >> 3, sc = idx & 7;
        T e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = tile[(sc * 8 + j) * LDW + ((d + 8 * sc) & 127)];
        T* dst = vt + ((int64_t)h * D + d) * Skp + col0 + s0 + sc * 8;
        if constexpr (sizeof(T) == 2) {
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (uint32_t)e[2 * j] | ((uint32_t)e[2 * j + 1] << 16);
            *(u32x4*)dst = o;
        } else {
            *(f32x4*)dst = f32x4{e[0], e[1], e[2], e[3]};
            *(f32x4*)(dst + 4) = f32x4{e[4], e[5], e[6], e[7]};
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Wan's q / k path in ONE pass (reference transformer/wan/base/attention.py:305-413 with InplaceRMSNorm,
// transformer/efficiency/mod.py:24-35): RMSNorm over ALL H * 128 channels of a row (affine), the result rounded to the storage
// type exactly where the in-place norm writes it, rotary embedding, [H, S_out, 128] layout — what ln_modulate(rms) on q,
// ln_modulate(rms) on k and qk_norm_rope do as three passes, with the same arithmetic in the same order (one wave per row, lane
// `l` holds elements (it * 64 + l) * 8 .. + 7; wave_sum of the squares): bit-identical.  Units: row * nk + which.  The leading
// nb_v workgroups transpose V (v_transpose_body) when there is one.
// ------------------------------------------------------------------------------------------------
template <typename T>
APEXMI_DEVICE void round_to_storage(float (&y)[8]) {
    if constexpr (sizeof(T) == 2) {
        const u32x4 p = pack8(y);
        unpack8(p, y);
    }
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void qk_rms_rope_rows_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, int64_t ld_in, int S, int H,
    const bf16_t* __restrict__ wq, const bf16_t* __restrict__ wk, float eps, const float* __restrict__ rope, int rope_mode,
    T* __restrict__ qo, T* __restrict__ ko, T* __restrict__ vt, int S_out, int Skp, int row0, int nb_v, int nst) {
    if ((int)blockIdx.x < nb_v) {
        v_transpose_body<T>(blockIdx.x % nst, blockIdx.x / nst, v, 128, ld_in, S, 128, vt, Skp, row0);
        return;
    }
    constexpr int D = 128;
    const int C = NCH * 512;
    const int lane = threadIdx.x & 63;
    const int nk = k != nullptr ? 2 : 1;
    const int64_t unit = (int64_t)(blockIdx.x - nb_v) * 4 + (threadIdx.x >> 6);
    if (unit >= (int64_t)S * nk) return;                 // whole waves leave together
    const int s = (int)(unit / nk), which = (int)(unit % nk);
    const T* xp = (which ? k : q) + (int64_t)s * ld_in;
    const bf16_t* w = which ? wk : wq;
    float x[NCH][8];
    float sum = 0.0f;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        load8<T>(xp + (it * 64 + lane) * 8, x[it]);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += x[it][j];
    }
    (void)sum;
    float sq = 0.0f;
#pragma unroll
    for (int it = 0; it < NCH; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = x[it][j] - 0.0f;
            sq += d * d;
        }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
    const int srow = row0 + s;
    T* dst = which ? ko : qo;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int c = it * 64 + lane;                    // 8-element chunk of the row
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = (x[it][j] - 0.0f) * rstd;
        if (w != nullptr) {
            float g[8];
            unpack8(*(const u32x4*)(w + c * 8), g);
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] *= g[j];
        }
        round_to_storage<T>(y);                          // the storage point of the in-place norm
        const int h = c >> 4, l16 = c & 15, d = l16 * 8;
        float o[8];
        if (rope_mode == APEXMI_ROPE_INTERLEAVED) {
            const float* cp = rope + (int64_t)srow * D + d;
            const float* sp = rope + (int64_t)S_out * D + (int64_t)srow * D + d;
            const f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4);
            const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
            float cs[8], sn[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cs[j] = c0[j];
                cs[j + 4] = c1[j];
                sn[j] = s0[j];
                sn[j + 4] = s1[j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[2 * i] = fmaf(y[2 * i], cs[2 * i], -(y[2 * i + 1] * sn[2 * i]));
                o[2 * i + 1] = fmaf(y[2 * i + 1], cs[2 * i + 1], y[2 * i] * sn[2 * i + 1]);
            }
        } else if (rope_mode == APEXMI_ROPE_COMPLEX) {
            const float* tp = rope + ((int64_t)srow * (D / 2) + l16 * 4) * 2;
            const f32x4 t0 = *(const f32x4*)tp, t1 = *(const f32x4*)(tp + 4);
            const float cs[4] = {t0[0], t0[2], t1[0], t1[2]};
            const float sn[4] = {t0[1], t0[3], t1[1], t1[3]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[2 * i] = fmaf(y[2 * i], cs[i], -(y[2 * i + 1] * sn[i]));
                o[2 * i + 1] = fmaf(y[2 * i], sn[i], y[2 * i + 1] * cs[i]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = y[j];
        }
        store8<T>(dst + ((int64_t)h * S_out + srow) * D + d, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void v_transpose_kernel(const T* __restrict__ v, int64_t v_sh, int64_t v_ss, int S,
                                                          int D, T* __restrict__ vt, int Skp, int col0) {
    v_transpose_body<T>(blockIdx.x, blockIdx.y, v, v_sh, v_ss, S, D, vt, Skp, col0);
}

template <typename T>
__global__ __launch_bounds__(256) void qk_norm_rope4_kernel(
    const T* __restrict__ q, const T* __restrict__ k, int64_t ld_in, int S, int H,
    int split, const bf16_t* __restrict__ wq, const bf16_t* __restrict__ wk,
    const bf16_t* __restrict__ wq2, const bf16_t* __restrict__ wk2, float eps,
    const float* __restrict__ rope, int rope_mode, T* __restrict__ qo,
    T* __restrict__ ko, int S_out, int row0) {
    qk_norm_rope4_body<T>(blockIdx.x, q, k, ld_in, S, H, split, wq, wk, wq2, wk2, eps, rope, rope_mode, qo, ko, S_out, row0);
}

// q/k norm + RoPE and the V transpose of one attention layer in ONE launch: both are short, latency-bound passes
// over disjoint data (29 + 16 us at the Flux shape when launched back to back), so their workgroups share the chip.
// Blocks [0, nb_v) transpose V (64-key tiles x heads), the rest run the q/k groups.
template <typename T>
__global__ __launch_bounds__(256) void qkv_prepare_fused_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, int64_t ld_in, int S, int H,
    int split, const bf16_t* __restrict__ wq, const bf16_t* __restrict__ wk,
    const bf16_t* __restrict__ wq2, const bf16_t* __restrict__ wk2, float eps,
    const float* __restrict__ rope, int rope_mode, T* __restrict__ qo,
    T* __restrict__ ko, T* __restrict__ vt, int S_out, int Skp, int row0, int nb_v, int nst) {
    if ((int)blockIdx.x < nb_v)
        v_transpose_body<T>(blockIdx.x % nst, blockIdx.x / nst, v, 128, ld_in, S, 128, vt, Skp, row0);
    else
        qk_norm_rope4_body<T>(blockIdx.x - nb_v, q, k, ld_in, S, H, split, wq, wk, wq2, wk2, eps, rope, rope_mode, qo, ko,
                           S_out, row0);
}

// strided [B,H,S,D] view -> packed copy
__global__ __launch_bounds__(256) void pack_bhsd_kernel(const bf16_t* __restrict__ x, int64_t sb,
                                                        int64_t sh, int64_t ss, int H, int S,
                                                        int D, bf16_t* __restrict__ out,
                                                        int64_t nchunk) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nchunk) return;
    const int dc = D >> 3;
    const int c = (int)(idx % dc);
    int64_t r = idx / dc;
    const int s = (int)(r % S);
    r /= S;
    const int h = (int)(r % H);
    const int b = (int)(r / H);
    *(u32x4*)(out + idx * 8) = *(const u32x4*)(x + b * sb + h * sh + s * ss + c * 8);
}

// ------------------------------------------------------------------------------------------------
// y[m][n] = post(dot(W[n,:], pre(x[m,:])) + bias[n]); one wave per output row, x in LDS (f32).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ W, int64_t ldw,
                                                   const bf16_t* __restrict__ bias,
                                                   const float* __restrict__ x, int64_t ldx,
                                                   float* __restrict__ y, int64_t ldy, int N, int K,
                                                   int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = (float*)smem;
    const int m = blockIdx.y;
    const float* xp = x + (int64_t)m * ldx;
    for (int i = threadIdx.x; i < K; i += 256) {
        float v = xp[i];
        if (flags & APEXMI_GEMV_PRE_SILU) v = silu_f(v);
        xs[i] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nw = gridDim.x * 4;
    const int nchunk = K >> 3;
    for (int n = gw; n < N; n += nw) {
        const bf16_t* wp = W + (int64_t)n * ldw;
        float acc = 0.0f;
        for (int c = lane; c < nchunk; c += 64) {
            float w[8];
            unpack8(*(const u32x4*)(wp + c * 8), w);
            const f32x4 x0 = *(const f32x4*)(xs + c * 8);
            const f32x4 x1 = *(const f32x4*)(xs + c * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = fmaf(w[j], x0[j], acc);
                acc = fmaf(w[j + 4], x1[j], acc);
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            if (bias != nullptr) acc += bf16_to_f32(bias[n]);
            if (flags & APEXMI_GEMV_POST_SILU) acc = silu_f(acc);
            if (flags & APEXMI_GEMV_POST_GELU) acc = gelu_tanh_f(acc);
            float* yp = y + (int64_t)m * ldy + n;
            if (flags & APEXMI_GEMV_ACCUM) acc += *yp;
            *yp = acc;
        }
    }
}

// The same product for MB rows of x in ONE pass over W (the modulation table of a whole clip: every step's conditioning
// vector against the 6.4 GB of stacked AdaLN projections, read once instead of once per step).  Per (n, m) the arithmetic is
// gemv_kernel's, operation for operation: a lane owns chunks lane, lane + 64, ... in ascending order, the same fmaf chain
// inside a chunk, the same butterfly — so row m of this kernel is BIT-IDENTICAL to a single-row launch on x[m].
// x rows sit in LDS split into the chunks' low / high halves (xs0 / xs1: consecutive lanes read consecutive 16 bytes).
template <int MB>
__global__ __launch_bounds__(512) void gemv_rows_kernel(const bf16_t* __restrict__ W, int64_t ldw,
                                                        const bf16_t* __restrict__ bias,
                                                        const float* __restrict__ x, int64_t ldx,
                                                        float* __restrict__ y, int64_t ldy, int M, int N, int K,
                                                        int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = (float*)smem;                       // [MB][2][K / 2]
    const int m0 = blockIdx.y * MB;
    const int mb = min(MB, M - m0);
    const int half = K >> 1;
    for (int i = threadIdx.x; i < MB * K; i += 512) {
        const int r = i / K, k = i - r * K;
        float v = 0.0f;
        if (r < mb) {
            v = x[(int64_t)(m0 + r) * ldx + k];
            if (flags & APEXMI_GEMV_PRE_SILU) v = silu_f(v);
        }
        xs[r * K + ((k >> 2) & 1) * half + (k >> 3) * 4 + (k & 3)] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 8 + (threadIdx.x >> 6);
    const int nw = gridDim.x * 8;
    const int nchunk = K >> 3;
    // Round 6: the weight row of the NEXT output column is in flight while this one is multiplied (up to eight 16-byte chunks per
    // lane, K <= 4096): a wave used to have one chunk in flight, 8 KiB per CU against the >= 50 KiB an HBM stream needs — the
    // 20-row table of a Flux clip took 9.4 ms for three passes over 6.5 GB.  Same chunks per lane, same order of the sums.
    constexpr int PF = 8;
    const bool pf = nchunk <= PF * 64;
    u32x4 wnext[PF];
    auto load_row = [&](int n) {
        const bf16_t* wp = W + (int64_t)n * ldw;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) wnext[i] = *(const u32x4*)(wp + c * 8);
        }
    };
    if (pf && gw < N) load_row(gw);
    for (int n = gw; n < N; n += nw) {
        const bf16_t* wp = W + (int64_t)n * ldw;
        float acc[MB];
#pragma unroll
        for (int r = 0; r < MB; ++r) acc[r] = 0.0f;
        if (pf) {
            u32x4 wcur[PF];
#pragma unroll
            for (int i = 0; i < PF; ++i) wcur[i] = wnext[i];
            if (n + nw < N) load_row(n + nw);
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int c = lane + 64 * i;
                if (c < nchunk) {
                    float w[8];
                    unpack8(wcur[i], w);
#pragma unroll
                    for (int r = 0; r < MB; ++r) {
                        const f32x4 x0 = *(const f32x4*)(xs + r * K + c * 4);
                        const f32x4 x1 = *(const f32x4*)(xs + r * K + half + c * 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[r] = fmaf(w[j], x0[j], acc[r]);
                            acc[r] = fmaf(w[j + 4], x1[j], acc[r]);
                        }
                    }
                }
            }
        } else {
        for (int c = lane; c < nchunk; c += 64) {
            float w[8];
            unpack8(*(const u32x4*)(wp + c * 8), w);
#pragma unroll
            for (int r = 0; r < MB; ++r) {
                const f32x4 x0 = *(const f32x4*)(xs + r * K + c * 4);
                const f32x4 x1 = *(const f32x4*)(xs + r * K + half + c * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[r] = fmaf(w[j], x0[j], acc[r]);
                    acc[r] = fmaf(w[j + 4], x1[j], acc[r]);
                }
            }
        }
        }
        float mine = 0.0f;
#pragma unroll
        for (int r = 0; r < MB; ++r) {
            const float t = wave_sum(acc[r]);
            if (lane == r) mine = t;
        }
        if (lane < mb) {
            if (bias != nullptr) mine += bf16_to_f32(bias[n]);
            if (flags & APEXMI_GEMV_POST_SILU) mine = silu_f(mine);
            if (flags & APEXMI_GEMV_POST_GELU) mine = gelu_tanh_f(mine);
            float* yp = y + (int64_t)(m0 + lane) * ldy + n;
            if (flags & APEXMI_GEMV_ACCUM) mine += *yp;
            *yp = mine;
        }
    }
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, float* __restrict__ out,
                                          int M, int dim, float scale, int flip, float shift,
                                          const float* __restrict__ freqs) {
    const int half = dim / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * half) return;
    const int m = idx / half, i = idx % half;
    // exponent = -ln(10000) * i / (half - shift)
    const float e = freqs ? freqs[i] : expf(-9.210340371976184f * (float)i / ((float)half - shift));
    const float a = __fmul_rn(__fmul_rn(t[m], e), scale);   // (t * freq) * scale, two f32 roundings as in the reference
    const float sn = sinf(a), cs = cosf(a);
    float* o = out + (int64_t)m * dim;
    if (flip) {
        o[i] = cs;
        o[half + i] = sn;
    } else {
        o[i] = sn;
        o[half + i] = cs;
    }
}

struct AxesDims {
    int n;
    int dim[4];
};
__global__ void rope_table_axes_kernel(const float* __restrict__ ids, int S, AxesDims ax, int D,
                                       double log_theta, float* __restrict__ out) {
    const int half = D / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * half) return;
    const int s = idx / half;
    int j = idx % half;  // pair index inside the row
    int a = 0;
    while (a < ax.n - 1 && j >= ax.dim[a] / 2) {
        j -= ax.dim[a] / 2;
        ++a;
    }
    // freq = 1 / theta^(2j / dim_a), angle = pos * freq, all in f64
    const double freq = exp(-log_theta * (double)(2 * j) / (double)ax.dim[a]);
    const double ang = (double)ids[(int64_t)s * ax.n + a] * freq;
    const float c = (float)cos(ang), sn = (float)sin(ang);
    const int col = 2 * (idx % half);
    float* cp = out + (int64_t)s * D + col;
    float* sp = out + (int64_t)S * D + (int64_t)s * D + col;
    cp[0] = c;
    cp[1] = c;
    sp[0] = sn;
    sp[1] = sn;
}

// out[l][i] = a[l][i] + b[i]  (per-block modulation tables + the step's timestep projection)
__global__ void add_bcast_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                     float* __restrict__ out, int64_t rows, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * n) out[i] = a[i] + b[i % n];
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ o, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = f32_to_bf16(x[i]);
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ o, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = bf16_to_f32(x[i]);
}

// prev = sample + dt * model_output (f32 math); FlowMatchEulerDiscreteScheduler.step
template <int F32>
__global__ void euler_step_kernel(const void* __restrict__ sample, const bf16_t* __restrict__ v,
                                  void* __restrict__ out, int64_t n, float dt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (F32) {
        ((float*)out)[i] = ((const float*)sample)[i] + dt * bf16_to_f32(v[i]);
    } else {
        const float sv = bf16_to_f32(((const bf16_t*)sample)[i]);
        ((bf16_t*)out)[i] = f32_to_bf16(sv + dt * bf16_to_f32(v[i]));
    }
}

// float8_e4m3fn (OCP: bias 7, no infinity, S.1111.111 = NaN) and float8_e5m2 (IEEE-like, bias 15) -> f32, by bits,
// so the result is the one torch's .to(float32) gives on every value including subnormals.
APEXMI_DEVICE float fp8_e4m3fn_to_f32(uint32_t b) {
    const uint32_t sign = (b & 0x80u) << 24, e = (b >> 3) & 0xFu, m = b & 7u;
    if (e == 0xFu && m == 7u) return __uint_as_float(sign | 0x7FC00000u);
    if (e == 0u) return __uint_as_float(sign | __float_as_uint((float)m * 0.001953125f));  // m * 2^-9
    return __uint_as_float(sign | ((e + 120u) << 23) | (m << 20));
}
APEXMI_DEVICE float fp8_e5m2_to_f32(uint32_t b) {
    const _Float16 h = __builtin_bit_cast(_Float16, (uint16_t)(b << 8));  // e5m2 is the top byte of an fp16
    return (float)h;
}

__global__ __launch_bounds__(256) void dequant_fp8_scaled_kernel(const uint8_t* __restrict__ w, int format,
                                                                 const bf16_t* __restrict__ scale, int per_row,
                                                                 int64_t rows, int64_t cols, bf16_t* __restrict__ out,
                                                                 int64_t ldo) {
    const int64_t r = blockIdx.y;
    const float s = bf16_to_f32(scale[per_row ? r : 0]);
    const uint8_t* wr = w + r * cols;
    bf16_t* orow = out + r * ldo;
    for (int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; c < cols; c += (int64_t)gridDim.x * 256 * 8) {
        if (c + 8 <= cols && ((cols | ldo) & 7) == 0) {
            const u32x2 q = *(const u32x2*)(wr + c);
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t byte = (q[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                // the reference multiplies two bf16 tensors: fp8 -> bf16 is exact, the product rounds once
                f[j] = (format == 0 ? fp8_e4m3fn_to_f32(byte) : bf16_to_f32(f32_to_bf16(fp8_e5m2_to_f32(byte)))) * s;
            }
            *(u32x4*)(orow + c) = pack8(f);
        } else {
            for (int64_t j = c; j < min(c + 8, cols); ++j) {
                const float v = format == 0 ? fp8_e4m3fn_to_f32(wr[j]) : bf16_to_f32(f32_to_bf16(fp8_e5m2_to_f32(wr[j])));
                orow[j] = f32_to_bf16(v * s);
            }
        }
    }
}

// out[r, c] = x[r, c] + v[c] (bf16 in / out, f32 add): `tokens + cond_type_embed(type)` of the HunyuanVideo-1.5
// conditioning streams (transformer/hunyuanvideo15/base/model.py:1013-1056)
template <typename T>     // T: storage type of x / out (bf16_t in production, float in the f32-storage verification mode); v is a weight
__global__ __launch_bounds__(256) void add_rowvec_kernel(const T* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ v,
                                                         T* __restrict__ out, int64_t ldo, int64_t rows, int cols) {
    const int nch = cols >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * nch) return;
    const int64_t r = idx / nch;
    const int c = (int)(idx % nch) * 8;
    float a[8], b[8];
    load8<T>(x + r * ldx + c, a);
    unpack8(*(const u32x4*)(v + c), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += b[j];
    store8<T>(out + r * ldo + c, a);
}

// out = a + b (bf16, f32 add): `h + shortcut` of HunyuanVideo15Upsample.forward / Decoder3D.forward after the DCAE
// rearranges (vae/hunyuanvideo15/model.py:274, :709-711), where the two addends come out of different layouts
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float x[8], y[8];
    load8<T>(a + i * 8, x);
    load8<T>(b + i * 8, y);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] += y[j];
    store8<T>(out + i * 8, x);
}

// out = a * b (bf16, f32 product): `hidden_gelu * hidden_linear` of T5DenseGatedActDense (transformers
// models/t5/modeling_t5.py, gated-gelu feed-forward of T5 v1.1 / UMT5)
__global__ __launch_bounds__(256) void mul_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                       bf16_t* __restrict__ out, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float x[8], y[8];
    unpack8(*(const u32x4*)(a + i * 8), x);
    unpack8(*(const u32x4*)(b + i * 8), y);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] *= y[j];
    *(u32x4*)(out + i * 8) = pack8(x);
}

// f32-storage verification forms of the two above (text encoders): float out, the sum / product unrounded
__global__ __launch_bounds__(256) void mul_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                      int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}
__global__ __launch_bounds__(256) void gather_rows_f32_kernel(const bf16_t* __restrict__ table, int64_t ldt, const int64_t* __restrict__ ids,
                                                              int64_t vocab, const bf16_t* __restrict__ pos, int64_t ldp, int period,
                                                              float* __restrict__ out, int64_t ldo, int64_t rows, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * C) return;
    const int64_t r = i / C;
    const int c = (int)(i % C);
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    float v = bf16_to_f32(table[id * ldt + c]);
    if (pos) v += bf16_to_f32(pos[(r % period) * ldp + c]);
    out[r * ldo + c] = v;
}

// out[r, :] = table[ids[r], :] (+ pos[r % period, :]): nn.Embedding lookups of the text encoders (token embedding;
// CLIPTextEmbeddings adds the learned position embedding of the row's position)
__global__ __launch_bounds__(256) void gather_rows_bf16_kernel(const bf16_t* __restrict__ table, int64_t ldt,
                                                               const int64_t* __restrict__ ids, int64_t vocab,
                                                               const bf16_t* __restrict__ pos, int64_t ldp, int period,
                                                               bf16_t* __restrict__ out, int64_t ldo, int64_t rows,
                                                               int c8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * c8) return;
    const int64_t r = i / c8;
    const int c = (int)(i % c8) * 8;
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);   // out-of-range ids are rejected on the host; never fault here
    u32x4 val = *(const u32x4*)(table + id * ldt + c);
    if (pos) {
        float x[8], y[8];
        unpack8(val, x);
        unpack8(*(const u32x4*)(pos + (r % period) * ldp + c), y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += y[j];
        val = pack8(x);
    }
    *(u32x4*)(out + r * ldo + c) = val;
}

// bias[h, i, j] = weight[bucket[j - i + Sq - 1], h]: T5Attention.compute_bias (modeling_t5.py) with the bucket of every
// relative distance j - i in [-(Sq-1), Sk-1] computed once on the host (integer / log arithmetic, 2S-1 values)
__global__ __launch_bounds__(256) void relpos_bias_kernel(const bf16_t* __restrict__ weight, int H,
                                                          const int* __restrict__ bucket, int Sq, int Sk,
                                                          float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = (int64_t)H * Sq * Sk;
    if (i >= n) return;
    const int j = (int)(i % Sk);
    const int q = (int)((i / Sk) % Sq);
    const int h = (int)(i / ((int64_t)Sk * Sq));
    out[i] = bf16_to_f32(weight[(int64_t)bucket[j - q + Sq - 1] * H + h]);
}

// x[r, h, i] <- x[r, h, i] cos[r, i] - x[r, h, i + D/2] sin[r, i];  x[r, h, i + D/2] <- x[r, h, i + D/2] cos[r, i + D/2]
// + x[r, h, i] sin[r, i + D/2]  — `q * cos + rotate_half(q) * sin` of transformers' Qwen2.5-VL
// (apply_rotary_pos_emb_vision / apply_multimodal_rotary_pos_emb, modeling_qwen2_5_vl.py), in place on a packed
// projection [rows, heads * head_stride] whose heads rotate their first D columns (the vision tower's 80-wide heads
// live in 128-wide slots).  f32 arithmetic, one rounding to bf16.
template <typename TS>
__global__ __launch_bounds__(256) void rope_half_kernel(TS* __restrict__ x, int64_t ldx, int64_t rows, int heads,
                                                        int head_stride, int D, const float* __restrict__ cs,
                                                        const float* __restrict__ sn) {
    const int half = D >> 1;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * heads * half) return;
    const int c = (int)(i % half);
    const int h = (int)((i / half) % heads);
    const int64_t r = i / ((int64_t)half * heads);
    TS* p = x + r * ldx + (int64_t)h * head_stride;
    const float a = load1<TS>(p + c), b = load1<TS>(p + c + half);
    const float* cr = cs + r * D;
    const float* sr = sn + r * D;
    store1<TS>(p + c, a * cr[c] - b * sr[c]);
    store1<TS>(p + c + half, b * cr[c + half] + a * sr[c + half]);
}

// frames[t, y, x, c] = uint8(round(clamp(v * 0.5 + 0.5, 0, 1) * 255)) for v = video[c, t, y, x] (any strides): the
// denormalize -> permute -> (x 255).round().astype(uint8) chain of diffusers VideoProcessor.postprocess_video that
// BaseEngine._tensor_to_frames calls (engine/base_engine.py:2945-2949).  The reference runs denormalize in the decode
// dtype, so for bf16 input the sum v/2 + 1/2 is rounded to bf16 before the scaling — reproduced here, bit for bit.
template <typename TV>
__global__ __launch_bounds__(256) void frames_to_u8_kernel(const TV* __restrict__ x, int64_t sc, int64_t st, int64_t sy,
                                                           int64_t sx, int C, int T, int H, int W,
                                                           uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = (int64_t)T * H * W;
    if (i >= n) return;
    const int xw = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int t = (int)(i / ((int64_t)W * H));
    const TV* p = x + t * st + y * sy + xw * sx;
    for (int c = 0; c < C; ++c) {
        float u;
        if constexpr (sizeof(TV) == 2) {   // denormalize runs in the decode dtype: two bf16 roundings
            const float half = bf16_to_f32(f32_to_bf16(bf16_to_f32(p[c * sc]) * 0.5f));
            u = bf16_to_f32(f32_to_bf16(half + 0.5f));
        } else {                          // float video: the same chain in f32
            u = __fadd_rn(__fmul_rn(p[c * sc], 0.5f), 0.5f);
        }
        u = fminf(fmaxf(u, 0.0f), 1.0f);
        out[i * C + c] = (uint8_t)rintf(u * 255.0f);
    }
}

}  // namespace

extern "C" int apexmi_ln_modulate(const void* x, int64_t ldx, void* out, int64_t ldo, int M, int C,
                                  const float* scale, const float* shift, const void* gamma,
                                  const void* beta, float eps, int rms, apexmi_stream_t stream_) {
    return apexmi_ln_modulate2(x, ldx, out, ldo, M, C, scale, shift, gamma, beta, eps, rms, 0, nullptr,
                               nullptr, stream_);
}

int g_ln_wave = 1;  // apexmi_tune_set("ln.wave", 0/1): wave-per-row kernel for C in {3072, 3584, 5120}
void apexmi_set_ln_wave(int v) { g_ln_wave = v; }

template <typename T>
static int ln_modulate2_impl(const void* x, int64_t ldx, void* out, int64_t ldo, int M, int C,
                             const float* scale, const float* shift, const void* gamma,
                             const void* beta, float eps, int rms, int split,
                             const float* scale2, const float* shift2, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(split >= 0 && split <= M, "ln_modulate: split=%d outside [0, M]", split);
    APEXMI_REQUIRE(split == 0 || ((!scale2 || ((uintptr_t)scale2 % 16) == 0) && (!shift2 || ((uintptr_t)shift2 % 16) == 0)),
                   "ln_modulate: scale2/shift2 must be 16-byte aligned");
    APEXMI_REQUIRE(x && out, "ln_modulate: null operand");
    APEXMI_REQUIRE(M > 0 && C > 0, "ln_modulate: empty problem");
    APEXMI_REQUIRE(C % 8 == 0 && C <= LN_MAX_C, "ln_modulate: C=%d must be a multiple of 8 and <= %d", C,
                   LN_MAX_C);
    APEXMI_REQUIRE(ldx % (16 / (int)sizeof(T)) == 0 && ldo % (16 / (int)sizeof(T)) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0,
                   "ln_modulate: rows must be 16-byte aligned");
    APEXMI_REQUIRE((!scale || ((uintptr_t)scale % 16) == 0) && (!shift || ((uintptr_t)shift % 16) == 0),
                   "ln_modulate: scale/shift must be 16-byte aligned");
    ApexmiProfScope prof(3, stream, 0.0, 2.0 * sizeof(T) * (double)M * C);
#define LNW_LAUNCH_R(N, R)                                                                                          \
    hipLaunchKernelGGL((ln_modulate_wave_kernel<T, N, R>), dim3((M + 4 * R - 1) / (4 * R)), dim3(256), 0, stream, (const T*)x, ldx, \
                       (T*)out, ldo, M, C, scale, shift, (const bf16_t*)gamma, (const bf16_t*)beta, eps, rms, \
                       split, scale2, shift2)
#define LNW_LAUNCH(N)                         \
    do {                                      \
        if (g_ln_wave >= 2 && sizeof(T) == 2) \
            LNW_LAUNCH_R(N, 2);               \
        else                                  \
            LNW_LAUNCH_R(N, 1);               \
    } while (0)
    if (g_ln_wave && C % 512 == 0) {
        bool done = true;
        switch (C / 512) {
            case 6: LNW_LAUNCH(6); break;
            case 7: LNW_LAUNCH(7); break;
            case 10: LNW_LAUNCH(10); break;
            default: done = false;
        }
        if (done) return apexmi_check_launch("ln_modulate");
    }
#undef LNW_LAUNCH_R
#undef LNW_LAUNCH
    const int nit = (C / 8 + 255) / 256;
#define LN_LAUNCH(N)                                                                                  \
    hipLaunchKernelGGL((ln_modulate_kernel<T, N>), dim3(M), dim3(256), 0, stream, (const T*)x, ldx,   \
                       (T*)out, ldo, M, C, scale, shift, (const bf16_t*)gamma,                   \
                       (const bf16_t*)beta, eps, rms, split, scale2, shift2)
    if (nit <= 1) LN_LAUNCH(1);
    else if (nit <= 2) LN_LAUNCH(2);
    else if (nit <= 3) LN_LAUNCH(3);
    else LN_LAUNCH(4);
#undef LN_LAUNCH
    return apexmi_check_launch("ln_modulate");
}

extern "C" int apexmi_ln_modulate2(const void* x, int64_t ldx, void* out, int64_t ldo, int M, int C,
                                   const float* scale, const float* shift, const void* gamma,
                                   const void* beta, float eps, int rms, int split,
                                   const float* scale2, const float* shift2, apexmi_stream_t stream_) {
    return ln_modulate2_impl<bf16_t>(x, ldx, out, ldo, M, C, scale, shift, gamma, beta, eps, rms, split, scale2, shift2, stream_);
}

// f32-storage verification mode: x and out are float (ldx / ldo in floats); gamma / beta stay bf16 weights
extern "C" int apexmi_ln_modulate2_f32(const void* x, int64_t ldx, void* out, int64_t ldo, int M, int C,
                                       const float* scale, const float* shift, const void* gamma,
                                       const void* beta, float eps, int rms, int split,
                                       const float* scale2, const float* shift2, apexmi_stream_t stream_) {
    return ln_modulate2_impl<float>(x, ldx, out, ldo, M, C, scale, shift, gamma, beta, eps, rms, split, scale2, shift2, stream_);
}

template <typename T>
static int v_transpose_impl(const void* v, int64_t v_stride_h, int64_t v_stride_s, int S, int H,
                            int D, void* vt, int Skp, int row0, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(v && vt, "v_transpose: null operand");
    APEXMI_REQUIRE(D == 128, "v_transpose: D=%d unsupported (128 only)", D);
    APEXMI_REQUIRE(row0 % 64 == 0 && Skp % 64 == 0 && row0 + ((S + 63) / 64) * 64 <= Skp,
                   "v_transpose: row0=%d S=%d Skp=%d not tile aligned", row0, S, Skp);
    APEXMI_REQUIRE(v_stride_h % (16 / (int)sizeof(T)) == 0 && v_stride_s % (16 / (int)sizeof(T)) == 0 && ((uintptr_t)v % 16) == 0 &&
                       ((uintptr_t)vt % 16) == 0,
                   "v_transpose: rows must be 16-byte aligned");
    ApexmiProfScope prof(4, stream, 0.0, 2.0 * sizeof(T) * (double)S * H * D);
    hipLaunchKernelGGL(v_transpose_kernel<T>, dim3((S + 63) / 64, H), dim3(256), 0, stream,
                       (const T*)v, v_stride_h, v_stride_s, S, D, (T*)vt, Skp, row0);
    return apexmi_check_launch("v_transpose");
}

extern "C" int apexmi_v_transpose(const void* v, int64_t v_stride_h, int64_t v_stride_s, int S, int H,
                                  int D, void* vt, int Skp, int row0, apexmi_stream_t stream_) {
    return v_transpose_impl<bf16_t>(v, v_stride_h, v_stride_s, S, H, D, vt, Skp, row0, stream_);
}

int apexmi_pack_bhsd(const void* x, const int64_t* st, int B, int H, int S, int D, void* out,
                     hipStream_t stream) {
    APEXMI_REQUIRE(D % 8 == 0 && st[0] % 8 == 0 && st[1] % 8 == 0 && st[2] % 8 == 0 &&
                       ((uintptr_t)x % 16) == 0,
                   "attn_fwd: strided operand rows must be 16-byte aligned");
    const int64_t nchunk = (int64_t)B * H * S * (D / 8);
    ApexmiProfScope prof(4, stream, 0.0, 4.0 * (double)B * H * S * D);
    hipLaunchKernelGGL(pack_bhsd_kernel, dim3((unsigned)((nchunk + 255) / 256)), dim3(256), 0, stream,
                       (const bf16_t*)x, st[0], st[1], st[2], H, S, D, (bf16_t*)out, nchunk);
    return apexmi_check_launch("pack_bhsd");
}

int g_qk_group = 1;  // apexmi_tune_set("qk.group", n): 0 one head per lane group | 2 four heads | 1 four heads + V transpose in the same launch
void apexmi_set_qk_group(int v) { g_qk_group = v; }

template <typename T>
static int qkv_prepare_impl(const void* q, const void* k, const void* v, int64_t ld_in, int S,
                            int H, int D, int split, const void* wq, const void* wk,
                            const void* wq2, const void* wk2, float eps, const float* rope,
                            int rope_mode, void* qo, void* ko, void* vt, int S_out, int Skp,
                            int row0, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(q && qo && (!k || ko), "qkv_prepare: null operand");
    APEXMI_REQUIRE(D == 128, "qkv_prepare: D=%d unsupported (128 only)", D);
    APEXMI_REQUIRE(S > 0 && H > 0 && row0 >= 0 && row0 + S <= S_out, "qkv_prepare: bad row range");
    APEXMI_REQUIRE(ld_in % (16 / (int)sizeof(T)) == 0 && ((uintptr_t)q % 16) == 0 && (!k || ((uintptr_t)k % 16) == 0),
                   "qkv_prepare: rows must be 16-byte aligned");
    APEXMI_REQUIRE(rope_mode == APEXMI_ROPE_NONE || (rope && ((uintptr_t)rope % 16) == 0),
                   "qkv_prepare: rope table missing or misaligned");
    APEXMI_REQUIRE(split <= 0 || (wq2 && wk2) || (!wq && !wk), "qkv_prepare: split needs the second weight set");
    if (g_qk_group == 1 && H % 4 == 0 && v != nullptr && vt != nullptr && row0 % 64 == 0 && Skp % 64 == 0 &&
        row0 + ((S + 63) / 64) * 64 <= Skp && ((uintptr_t)v % 16) == 0 && ((uintptr_t)vt % 16) == 0) {
        ApexmiProfScope prof(4, stream, 0.0, 6.0 * sizeof(T) * (double)S * H * D);
        const int64_t ngrp = (int64_t)S * (k ? 2 : 1) * H / 4;
        const int nst = (S + 63) / 64, nb_v = nst * H;
        hipLaunchKernelGGL(qkv_prepare_fused_kernel<T>, dim3((unsigned)(nb_v + (ngrp + 15) / 16)), dim3(256), 0, stream,
                           (const T*)q, (const T*)k, (const T*)v, ld_in, S, H, split, (const bf16_t*)wq,
                           (const bf16_t*)wk, (const bf16_t*)wq2, (const bf16_t*)wk2, eps, rope, rope_mode, (T*)qo,
                           (T*)ko, (T*)vt, S_out, Skp, row0, nb_v, nst);
        return apexmi_check_launch("qkv_prepare_fused");
    }
    {
        ApexmiProfScope prof(4, stream, 0.0, 4.0 * sizeof(T) * (double)S * H * D);
        const int64_t nunit = (int64_t)S * (k ? 2 : 1) * H;
        if (g_qk_group && H % 4 == 0)
            hipLaunchKernelGGL(qk_norm_rope4_kernel<T>, dim3((unsigned)((nunit / 4 + 15) / 16)), dim3(256), 0, stream,
                               (const T*)q, (const T*)k, ld_in, S, H, split, (const bf16_t*)wq,
                               (const bf16_t*)wk, (const bf16_t*)wq2, (const bf16_t*)wk2, eps, rope,
                               rope_mode, (T*)qo, (T*)ko, S_out, row0);
        else
            hipLaunchKernelGGL(qk_norm_rope_kernel<T>, dim3((unsigned)((nunit + 15) / 16)), dim3(256), 0, stream,
                               (const T*)q, (const T*)k, ld_in, S, H, split, (const bf16_t*)wq,
                               (const bf16_t*)wk, (const bf16_t*)wq2, (const bf16_t*)wk2, eps, rope,
                               rope_mode, (T*)qo, (T*)ko, S_out, row0);
        if (int rc = apexmi_check_launch("qk_norm_rope")) return rc;
    }
    if (v != nullptr && vt != nullptr)
        return v_transpose_impl<T>(v, D, ld_in, S, H, D, vt, Skp, row0, stream_);
    return 0;
}

extern "C" int apexmi_qkv_prepare(const void* q, const void* k, const void* v, int64_t ld_in, int S,
                                  int H, int D, int split, const void* wq, const void* wk,
                                  const void* wq2, const void* wk2, float eps, const float* rope,
                                  int rope_mode, void* qo, void* ko, void* vt, int S_out, int Skp,
                                  int row0, apexmi_stream_t stream_) {
    return qkv_prepare_impl<bf16_t>(q, k, v, ld_in, S, H, D, split, wq, wk, wq2, wk2, eps, rope, rope_mode, qo, ko, vt,
                                    S_out, Skp, row0, stream_);
}

// f32-storage verification mode: q, k, v, qo, ko, vt are float (ld_in in floats); the norm weights stay bf16
extern "C" int apexmi_qkv_prepare_f32(const void* q, const void* k, const void* v, int64_t ld_in, int S,
                                      int H, int D, int split, const void* wq, const void* wk,
                                      const void* wq2, const void* wk2, float eps, const float* rope,
                                      int rope_mode, void* qo, void* ko, void* vt, int S_out, int Skp,
                                      int row0, apexmi_stream_t stream_) {
    return qkv_prepare_impl<float>(q, k, v, ld_in, S, H, D, split, wq, wk, wq2, wk2, eps, rope, rope_mode, qo, ko, vt,
                                   S_out, Skp, row0, stream_);
}

template <typename T>
static int qk_rms_rope_rows_impl(const void* q, const void* k, const void* v, int64_t ld_in, int S, int H, const void* wq,
                                 const void* wk, float eps, const float* rope, int rope_mode, void* qo, void* ko, void* vt,
                                 int S_out, int Skp, int row0, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    constexpr int al = 16 / (int)sizeof(T);
    APEXMI_REQUIRE(q && qo && S > 0 && H > 0, "qk_rms_rope_rows: bad arguments");
    APEXMI_REQUIRE((k == nullptr) == (ko == nullptr) && (v == nullptr) == (vt == nullptr), "qk_rms_rope_rows: k / ko and v / vt come in pairs");
    // the widths whose stand-alone RMSNorm runs on the one-wave-per-row kernel: the fused pass keeps ITS summation order
    APEXMI_REQUIRE(H * 128 == 3072 || H * 128 == 5120,
                   "qk_rms_rope_rows: H * 128 = %d is not 3072 / 5120; use the three-pass path", H * 128);
    APEXMI_REQUIRE(ld_in % al == 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 &&
                       ((uintptr_t)qo % 16) == 0 && ((uintptr_t)ko % 16) == 0 && ((uintptr_t)vt % 16) == 0 &&
                       ((uintptr_t)wq % 16) == 0 && ((uintptr_t)wk % 16) == 0 && ((uintptr_t)rope % 16) == 0,
                   "qk_rms_rope_rows: operands must be 16-byte aligned");
    APEXMI_REQUIRE(row0 >= 0 && row0 + S <= S_out && (rope_mode == APEXMI_ROPE_NONE || rope != nullptr), "qk_rms_rope_rows: rows / rope");
    APEXMI_REQUIRE(vt == nullptr || (Skp >= row0 + S && Skp % 8 == 0 && row0 % 8 == 0), "qk_rms_rope_rows: V^T needs Skp >= rows, 8-aligned");
    const int nst = (S + 63) / 64;
    const int nb_v = vt ? nst * H : 0;
    const int64_t units = (int64_t)S * (k ? 2 : 1);
    const unsigned grid = (unsigned)(nb_v + (units + 3) / 4);
    ApexmiProfScope prof(4, stream, 0.0, (double)sizeof(T) * 2.0 * (double)S * H * 128 * (k ? 2 : 1) + (v ? (double)sizeof(T) * 2.0 * S * H * 128 : 0.0));
#define QRR_LAUNCH(N)                                                                                                         \
    hipLaunchKernelGGL((qk_rms_rope_rows_kernel<T, N>), dim3(grid), dim3(256), 0, stream, (const T*)q, (const T*)k, (const T*)v, \
                       ld_in, S, H, (const bf16_t*)wq, (const bf16_t*)wk, eps, rope, rope_mode, (T*)qo, (T*)ko, (T*)vt, S_out,    \
                       Skp, row0, nb_v, nst)
    if (H * 128 == 3072) QRR_LAUNCH(6);
    else QRR_LAUNCH(10);
#undef QRR_LAUNCH
    return apexmi_check_launch("qk_rms_rope_rows");
}

extern "C" int apexmi_qk_rms_rope_rows(const void* q, const void* k, const void* v, int64_t ld_in, int S, int H, const void* wq,
                                       const void* wk, float eps, const float* rope, int rope_mode, void* qo, void* ko,
                                       void* vt, int S_out, int Skp, int row0, apexmi_stream_t stream_) {
    return qk_rms_rope_rows_impl<bf16_t>(q, k, v, ld_in, S, H, wq, wk, eps, rope, rope_mode, qo, ko, vt, S_out, Skp, row0, stream_);
}

extern "C" int apexmi_qk_rms_rope_rows_f32(const void* q, const void* k, const void* v, int64_t ld_in, int S, int H,
                                           const void* wq, const void* wk, float eps, const float* rope, int rope_mode, void* qo,
                                           void* ko, void* vt, int S_out, int Skp, int row0, apexmi_stream_t stream_) {
    return qk_rms_rope_rows_impl<float>(q, k, v, ld_in, S, H, wq, wk, eps, rope, rope_mode, qo, ko, vt, S_out, Skp, row0, stream_);
}

extern "C" int apexmi_gemv(const void* W, int64_t ldw, const void* bias, const float* x, int64_t ldx,
                           float* y, int64_t ldy, int M, int N, int K, int flags,
                           apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(W && x && y, "gemv: null operand");
    APEXMI_REQUIRE(M > 0 && M <= 4096 && N > 0 && K > 0, "gemv: bad shape M=%d N=%d K=%d", M, N, K);
    APEXMI_REQUIRE(K % 8 == 0 && K <= 16384, "gemv: K=%d must be a multiple of 8 and <= 16384", K);
    APEXMI_REQUIRE(ldw % 8 == 0 && ((uintptr_t)W % 16) == 0, "gemv: W rows must be 16-byte aligned");
    ApexmiProfScope prof(2, stream, 2.0 * M * (double)N * K, 2.0 * (double)N * K);
    if (M == 1) {
        int grid = (N + 3) / 4;
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(gemv_kernel, dim3(grid, M), dim3(256), (size_t)K * 4, stream, (const bf16_t*)W,
                           ldw, (const bf16_t*)bias, x, ldx, y, ldy, N, K, flags);
        return apexmi_check_launch("gemv");
    }
    // several rows of x: one pass over W per group of MB rows (MB x K floats of LDS), each row bit-identical to the
    // single-row kernel
    int grid = (N + 7) / 8;
    if (grid > 2048) grid = 2048;
#define GEMV_ROWS(MB)                                                                                                   \
    do {                                                                                                                \
        static uint64_t attr_##MB = 0;                                                                                  \
        APEXMI_SET_ATTR_ONCE(attr_##MB, (void)hipFuncSetAttribute((const void*)gemv_rows_kernel<MB>,                    \
                                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)); \
        hipLaunchKernelGGL((gemv_rows_kernel<MB>), dim3(grid, (M + MB - 1) / MB), dim3(512), (size_t)MB * K * 4, stream, \
                           (const bf16_t*)W, ldw, (const bf16_t*)bias, x, ldx, y, ldy, M, N, K, flags);                 \
    } while (0)
    const int cap = (128 * 1024) / (K * 4);
    // (twelve rows per pass — 144 KiB of LDS at K = 3072, one pass fewer for 20 / 28 rows — measured 3x SLOWER: the wave's registers)
    if (cap >= 8 && M > 4) GEMV_ROWS(8);
    else if (cap >= 4 && M > 2) GEMV_ROWS(4);
    else GEMV_ROWS(2);
#undef GEMV_ROWS
    return apexmi_check_launch("gemv");
}

extern "C" int apexmi_timestep_embedding(const float* t, float* out, int M, int dim, float scale,
                                         int flip_sin_to_cos, float downscale_freq_shift, const float* freqs,
                                         apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(t && out && M > 0 && dim > 0 && dim % 2 == 0, "timestep_embedding: bad arguments");
    const int n = M * (dim / 2);
    ApexmiProfScope prof(5, stream, 0.0, 4.0 * M * dim);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, t, out,
                       M, dim, scale, flip_sin_to_cos, downscale_freq_shift, freqs);
    return apexmi_check_launch("timestep_embedding");
}

extern "C" int apexmi_rope_table_axes(const float* ids, int S, int n_axes, const int* axes_dim,
                                      float theta, float* out, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(ids && out && axes_dim && S > 0, "rope_table_axes: bad arguments");
    APEXMI_REQUIRE(n_axes >= 1 && n_axes <= 4, "rope_table_axes: n_axes=%d unsupported", n_axes);
    AxesDims ax;
    ax.n = n_axes;
    int D = 0;
    for (int i = 0; i < 4; ++i) {
        ax.dim[i] = i < n_axes ? axes_dim[i] : 0;
        D += ax.dim[i];
        APEXMI_REQUIRE(ax.dim[i] % 2 == 0, "rope_table_axes: odd axis dim");
    }
    const int n = S * (D / 2);
    ApexmiProfScope prof(5, stream, 0.0, 8.0 * S * D);
    hipLaunchKernelGGL(rope_table_axes_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, ids, S, ax, D,
                       log((double)theta), out);
    return apexmi_check_launch("rope_table_axes");
}

// [2, S, D] -> [2, S, D / 2]: every second entry of a rotary table whose entries come in equal pairs; a pair that is NOT equal bumps *mismatch
__global__ __launch_bounds__(256) void rope_pairs_kernel(const float* __restrict__ rope, int64_t n_pairs, float* __restrict__ out,
                                                         int* __restrict__ mismatch) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pairs) return;
    const float a = rope[2 * i], b = rope[2 * i + 1];
    out[i] = a;
    if (__float_as_uint(a) != __float_as_uint(b)) atomicAdd(mismatch, 1);
}

extern "C" int apexmi_rope_pairs(const float* rope, int S, int D, float* pairs, int* mismatch, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(rope && pairs && mismatch && S > 0 && D > 0 && D % 2 == 0, "rope_pairs: bad arguments");
    const int64_t n = (int64_t)2 * S * (D / 2);
    ApexmiProfScope prof(5, stream, 0.0, 12.0 * (double)n);
    hipLaunchKernelGGL(rope_pairs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rope, n, pairs, mismatch);
    return apexmi_check_launch("rope_pairs");
}

extern "C" int apexmi_add_bcast_f32(const float* a, const float* b, float* out, int64_t rows, int64_t n,
                                    apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(a && b && out && rows > 0 && n > 0, "add_bcast_f32: bad arguments");
    ApexmiProfScope prof(5, stream, 0.0, 8.0 * rows * n);
    hipLaunchKernelGGL(add_bcast_f32_kernel, dim3((unsigned)((rows * n + 255) / 256)), dim3(256), 0, stream, a, b,
                       out, rows, n);
    return apexmi_check_launch("add_bcast_f32");
}

extern "C" int apexmi_dequant_fp8_scaled(const void* w, int format, const void* scale, int64_t scale_count,
                                         int64_t rows, int64_t cols, void* out, int64_t ldo, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(w && scale && out && rows > 0 && cols > 0 && ldo >= cols, "dequant_fp8_scaled: bad arguments");
    APEXMI_REQUIRE(format == 0 || format == 1, "dequant_fp8_scaled: format %d is neither e4m3fn (0) nor e5m2 (1)", format);
    APEXMI_REQUIRE(scale_count == 1 || scale_count == rows,
                   "dequant_fp8_scaled: scale_weight has %lld values for %lld output rows", (long long)scale_count,
                   (long long)rows);
    APEXMI_REQUIRE(rows < 65536 * 32767LL, "dequant_fp8_scaled: too many rows");
    ApexmiProfScope prof(5, stream, 0.0, 3.0 * rows * cols);
    const int64_t gx64 = (cols + 2047) / 2048;
    const unsigned gx = (unsigned)(gx64 < 64 ? gx64 : 64);
    // blockIdx.y is limited to 65535: fold longer matrices into several launches
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
        hipLaunchKernelGGL(dequant_fp8_scaled_kernel, dim3(gx, (unsigned)nr), dim3(256), 0, stream,
                           (const uint8_t*)w + r0 * cols, format,
                           (const bf16_t*)scale + (scale_count == 1 ? 0 : r0), (int)(scale_count != 1), nr, cols,
                           (bf16_t*)out + r0 * ldo, ldo);
    }
    return apexmi_check_launch("dequant_fp8_scaled");
}

extern "C" int apexmi_add_rowvec_bf16(const void* x, int64_t ldx, const void* v, void* out, int64_t ldo, int64_t rows,
                                      int cols, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(x && v && out && rows > 0 && cols > 0, "add_rowvec_bf16: bad arguments");
    APEXMI_REQUIRE(cols % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)v % 16) == 0 &&
                       ((uintptr_t)out % 16) == 0,
                   "add_rowvec_bf16: rows must be 16-byte aligned and cols a multiple of 8");
    ApexmiProfScope prof(5, stream, 0.0, 4.0 * rows * cols);
    const int64_t n = rows * (cols / 8);
    hipLaunchKernelGGL(add_rowvec_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x,
                       ldx, (const bf16_t*)v, (bf16_t*)out, ldo, rows, cols);
    return apexmi_check_launch("add_rowvec_bf16");
}

extern "C" int apexmi_add_rowvec_f32(const void* x, int64_t ldx, const void* v, void* out, int64_t ldo, int64_t rows,
                                     int cols, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(x && v && out && rows > 0 && cols > 0, "add_rowvec_f32: bad arguments");
    APEXMI_REQUIRE(cols % 8 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)v % 16) == 0 &&
                       ((uintptr_t)out % 16) == 0,
                   "add_rowvec_f32: rows must be 16-byte aligned and cols a multiple of 8");
    const int64_t n = rows * (cols / 8);
    hipLaunchKernelGGL(add_rowvec_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)x, ldx,
                       (const bf16_t*)v, (float*)out, ldo, rows, cols);
    return apexmi_check_launch("add_rowvec_f32");
}

template <typename TS>
__global__ __launch_bounds__(256) void group_mean_kernel(const TS* __restrict__ x, TS* __restrict__ out, int64_t n, int gs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // one output element: position * C + channel
    if (i >= n) return;
    const TS* p = x + i * gs;
    float s = 0.0f;
    for (int g = 0; g < gs; ++g) s += load1<TS>(p + g);
    store1<TS>(out + i, s / (float)gs);
}

extern "C" int apexmi_group_mean_bf16(const void* x, void* out, int64_t P, int C, int gs, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(x && out && P > 0 && C > 0 && gs >= 1 && gs <= 64, "group_mean_bf16: bad arguments (gs=%d)", gs);
    const int64_t n = P * C;
    ApexmiProfScope prof(5, stream, 0.0, 2.0 * n * (gs + 1));
    hipLaunchKernelGGL(group_mean_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x,
                       (bf16_t*)out, n, gs);
    return apexmi_check_launch("group_mean_bf16");
}

// f32-storage verification mode
extern "C" int apexmi_group_mean_f32(const void* x, void* out, int64_t P, int C, int gs, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(x && out && P > 0 && C > 0 && gs >= 1 && gs <= 64, "group_mean_f32: bad arguments (gs=%d)", gs);
    const int64_t n = P * C;
    hipLaunchKernelGGL(group_mean_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)x,
                       (float*)out, n, gs);
    return apexmi_check_launch("group_mean_f32");
}

extern "C" int apexmi_add_bf16(const void* a, const void* b, void* out, int64_t n, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(a && b && out && n > 0 && n % 8 == 0, "add_bf16: n=%lld must be a positive multiple of 8", (long long)n);
    APEXMI_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && ((uintptr_t)out % 16) == 0,
                   "add_bf16: operands must be 16-byte aligned");
    ApexmiProfScope prof(5, stream, 0.0, 6.0 * n);
    hipLaunchKernelGGL(add_kernel<bf16_t>, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)a,
                       (const bf16_t*)b, (bf16_t*)out, n / 8);
    return apexmi_check_launch("add_bf16");
}

extern "C" int apexmi_add_f32(const void* a, const void* b, void* out, int64_t n, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(a && b && out && n > 0 && n % 8 == 0, "add_f32: n=%lld must be a positive multiple of 8", (long long)n);
    APEXMI_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && ((uintptr_t)out % 16) == 0,
                   "add_f32: operands must be 16-byte aligned");
    hipLaunchKernelGGL(add_kernel<float>, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, stream, (const float*)a,
                       (const float*)b, (float*)out, n / 8);
    return apexmi_check_launch("add_f32");
}

extern "C" int apexmi_rope_half(void* x, int64_t ldx, int64_t rows, int heads, int head_stride, int D, const float* cos_,
                                const float* sin_, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(x && cos_ && sin_ && rows > 0 && heads > 0, "rope_half: bad arguments");
    APEXMI_REQUIRE(D > 0 && D % 2 == 0 && D <= head_stride, "rope_half: D=%d must be even and <= head_stride=%d", D, head_stride);
    const int64_t n = rows * heads * (D / 2);
    ApexmiProfScope prof(4, stream, 0.0, 4.0 * (double)rows * heads * D);
    hipLaunchKernelGGL(rope_half_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (bf16_t*)x, ldx, rows, heads,
                       head_stride, D, cos_, sin_);
    return apexmi_check_launch("rope_half");
}

// f32-storage verification mode
extern "C" int apexmi_rope_half_f32(void* x, int64_t ldx, int64_t rows, int heads, int head_stride, int D, const float* cos_,
                                    const float* sin_, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(x && cos_ && sin_ && rows > 0 && heads > 0, "rope_half_f32: bad arguments");
    APEXMI_REQUIRE(D > 0 && D % 2 == 0 && D <= head_stride, "rope_half_f32: D=%d must be even and <= head_stride=%d", D, head_stride);
    const int64_t n = rows * heads * (D / 2);
    hipLaunchKernelGGL(rope_half_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (float*)x, ldx, rows, heads,
                       head_stride, D, cos_, sin_);
    return apexmi_check_launch("rope_half_f32");
}

extern "C" int apexmi_frames_to_u8(const void* video, int64_t stride_c, int64_t stride_t, int64_t stride_h,
                                   int64_t stride_w, int C, int T, int H, int W, void* out, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(video && out && C > 0 && C <= 4 && T > 0 && H > 0 && W > 0, "frames_to_u8: bad arguments (C=%d)", C);
    const int64_t n = (int64_t)T * H * W;
    ApexmiProfScope prof(5, stream, 0.0, 3.0 * (double)n * C);
    hipLaunchKernelGGL(frames_to_u8_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)video,
                       stride_c, stride_t, stride_h, stride_w, C, T, H, W, (uint8_t*)out);
    return apexmi_check_launch("frames_to_u8");
}

// f32-storage verification mode: float video, the reference's fp32 chain (no intermediate bf16 roundings)
extern "C" int apexmi_frames_to_u8_f32(const void* video, int64_t stride_c, int64_t stride_t, int64_t stride_h,
                                       int64_t stride_w, int C, int T, int H, int W, void* out, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(video && out && C > 0 && C <= 4 && T > 0 && H > 0 && W > 0, "frames_to_u8: bad arguments (C=%d)", C);
    const int64_t n = (int64_t)T * H * W;
    ApexmiProfScope prof(5, stream, 0.0, 5.0 * (double)n * C);
    hipLaunchKernelGGL(frames_to_u8_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)video,
                       stride_c, stride_t, stride_h, stride_w, C, T, H, W, (uint8_t*)out);
    return apexmi_check_launch("frames_to_u8");
}

extern "C" int apexmi_mul_bf16(const void* a, const void* b, void* out, int64_t n, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(a && b && out && n > 0 && n % 8 == 0, "mul_bf16: n=%lld must be a positive multiple of 8", (long long)n);
    APEXMI_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && ((uintptr_t)out % 16) == 0,
                   "mul_bf16: operands must be 16-byte aligned");
    ApexmiProfScope prof(5, stream, 0.0, 6.0 * n);
    hipLaunchKernelGGL(mul_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)a,
                       (const bf16_t*)b, (bf16_t*)out, n / 8);
    return apexmi_check_launch("mul_bf16");
}

extern "C" int apexmi_mul_f32(const float* a, const float* b, float* out, int64_t n, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(a && b && out && n > 0, "mul_f32: bad arguments");
    hipLaunchKernelGGL(mul_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, b, out, n);
    return apexmi_check_launch("mul_f32");
}

extern "C" int apexmi_gather_rows_f32(const void* table, int64_t ldt, int64_t vocab, const int64_t* ids, const void* pos, int64_t ldp,
                                      int period, float* out, int64_t ldo, int64_t rows, int C, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(table && ids && out && rows > 0 && vocab > 0 && C > 0 && (!pos || period > 0), "gather_rows_f32: bad arguments");
    const int64_t n = rows * C;
    hipLaunchKernelGGL(gather_rows_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)table, ldt, ids,
                       vocab, (const bf16_t*)pos, ldp, pos ? period : 1, out, ldo, rows, C);
    return apexmi_check_launch("gather_rows_f32");
}

extern "C" int apexmi_gather_rows_bf16(const void* table, int64_t ldt, int64_t vocab, const int64_t* ids, const void* pos,
                                       int64_t ldp, int period, void* out, int64_t ldo, int64_t rows, int C,
                                       apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(table && ids && out && rows > 0 && vocab > 0, "gather_rows_bf16: bad arguments");
    APEXMI_REQUIRE(C > 0 && C % 8 == 0 && ldt % 8 == 0 && ldo % 8 == 0 && (!pos || (ldp % 8 == 0 && period > 0)),
                   "gather_rows_bf16: C=%d and the row strides must be multiples of 8", C);
    APEXMI_REQUIRE(((uintptr_t)table % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)pos % 16) == 0,
                   "gather_rows_bf16: operands must be 16-byte aligned");
    const int64_t n = rows * (C / 8);
    ApexmiProfScope prof(5, stream, 0.0, 4.0 * (double)rows * C);
    hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       (const bf16_t*)table, ldt, ids, vocab, (const bf16_t*)pos, ldp, pos ? period : 1, (bf16_t*)out, ldo,
                       rows, C / 8);
    return apexmi_check_launch("gather_rows_bf16");
}

extern "C" int apexmi_relpos_bias(const void* weight, int num_buckets, int H, const int* bucket, int Sq, int Sk,
                                  float* out, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(weight && bucket && out && H > 0 && Sq > 0 && Sk > 0 && num_buckets > 0, "relpos_bias: bad arguments");
    const int64_t n = (int64_t)H * Sq * Sk;
    ApexmiProfScope prof(5, stream, 0.0, 4.0 * (double)n);
    hipLaunchKernelGGL(relpos_bias_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)weight, H,
                       bucket, Sq, Sk, out);
    return apexmi_check_launch("relpos_bias");
}

extern "C" int apexmi_cast_f32_to_bf16(const float* x, void* out, int64_t n, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(x && out && n > 0, "cast_f32_to_bf16: bad arguments");
    ApexmiProfScope prof(5, stream, 0.0, 6.0 * n);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x,
                       (bf16_t*)out, n);
    return apexmi_check_launch("cast_f32_to_bf16");
}

extern "C" int apexmi_cast_bf16_to_f32(const void* x, float* out, int64_t n, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(x && out && n > 0, "cast_bf16_to_f32: bad arguments");
    ApexmiProfScope prof(5, stream, 0.0, 6.0 * n);
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       (const bf16_t*)x, out, n);
    return apexmi_check_launch("cast_bf16_to_f32");
}

extern "C" int apexmi_euler_step(const void* sample, const void* model_out, void* out, int64_t n,
                                 float dt, int sample_dtype, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(sample && model_out && out && n > 0, "euler_step: bad arguments");
    ApexmiProfScope prof(5, stream, 0.0, 6.0 * n);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (sample_dtype == APEXMI_F32)
        hipLaunchKernelGGL(euler_step_kernel<1>, grid, dim3(256), 0, stream, sample,
                           (const bf16_t*)model_out, out, n, dt);
    else if (sample_dtype == APEXMI_BF16)
        hipLaunchKernelGGL(euler_step_kernel<0>, grid, dim3(256), 0, stream, sample,
                           (const bf16_t*)model_out, out, n, dt);
    else {
        apexmi_set_error("euler_step: unsupported sample dtype %d", sample_dtype);
        return 1;
    }
    return apexmi_check_launch("euler_step");
}
