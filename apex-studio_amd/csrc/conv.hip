// Causal 3-D convolution (and its 2-D / 1x1 special cases) of the Wan / QwenImage VAE decoder as an
// implicit GEMM on MFMA, channels-last.
//
//   out[t, y, x, co] = bias[co] + sum_{tap, ci} W[co, tap, ci] * in[t + dt - (kT-1), y + dy - pH, x + dx - pW, ci]
//   (+ residual[t, y, x, co])
//
// Replaces WanCausalConv3d.forward (reference vae/wan/model.py:178-185: time padding is all on the left,
// 2*pad frames, i.e. causal; the per-frame feat_cache streaming of WanDecoder3d.forward :972-1021 is the
// same arithmetic as one causal convolution over the whole tile sequence), nn.Conv2d 3x3 of WanResample
// (:264-273) and the 1x1 convs (post_quant_conv, conv_shortcut, to_qkv, proj).  Zero padding at the
// TILE borders is part of the reference's numerical contract (tiled_decode :1574-1596), so every tile is
// convolved on its own.
//
// GEMM view: M = T*H*W output positions, N = Cout, K = ntaps * Cin (padded to 64).  Activations are
// [T, H, W, Cin] bf16, so for one tap the Cin run of a position is contiguous and the A tile is staged
// with the same 16-byte global_load_lds + XOR-swizzled LDS image as gemm.hip — only the per-lane source
// address changes (tap shift + bounds test; out-of-range taps read a 16-byte zero line).
// Weights are pre-packed [Cout, Kpad] (tap-major, ci-minor).  128x128x64 tile, 4 waves, 2 workgroups/CU.
#include "common.h"

// Per-workgroup timeline of the direct-convolution kernels (tools/conv_tile_trace.py): compiled in only with -DAPEXMI_CONV_TRACE=1 into
// a side library; the host reads a device pointer from APEXMI_CONV_TRACE_PTR (hex) at every slab launch.  8 u64 per workgroup:
// {HW_ID, XCC_ID, t_entry, t_loop_begin, t_loop_end, t_stores_issued, t_stores_acknowledged, 0} (s_memrealtime, 10 ns).
#ifndef APEXMI_CONV_TRACE
#define APEXMI_CONV_TRACE 0
#endif
#if APEXMI_CONV_TRACE
#include <stdlib.h>
__device__ unsigned long long* d_conv_trace = nullptr;
#define CONV_TRACE_DECL() unsigned long long* const trace_ = d_conv_trace; unsigned long long tr_in = 0, tr_l0 = 0, tr_l1 = 0; \
    if (trace_) tr_in = __builtin_amdgcn_s_memrealtime()
#define CONV_TRACE_AT(x) do { if (trace_) { __builtin_amdgcn_sched_barrier(0); x = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define CONV_TRACE_END()                                                                                     \
    do {                                                                                                     \
        if (trace_) {                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                               \
            const unsigned long long t_st = __builtin_amdgcn_s_memrealtime();                                \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                 \
            const unsigned long long t_ack = __builtin_amdgcn_s_memrealtime();                               \
            if (threadIdx.x == 0) {                                                                          \
                unsigned long long* o = trace_ + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;          \
                o[0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);                                 \
                o[1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);                                \
                o[2] = tr_in; o[3] = tr_l0; o[4] = tr_l1; o[5] = t_st; o[6] = t_ack;                         \
            }                                                                                                \
        }                                                                                                    \
    } while (0)
static void conv_trace_arm(hipStream_t stream) {
    const char* e = getenv("APEXMI_CONV_TRACE_PTR");
    unsigned long long* p = e ? (unsigned long long*)strtoull(e, nullptr, 16) : nullptr;
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(d_conv_trace), &p, sizeof(p), 0, hipMemcpyHostToDevice, stream);
}
#else
#define CONV_TRACE_DECL() do { } while (0)
#define CONV_TRACE_AT(x) do { } while (0)
#define CONV_TRACE_END() do { } while (0)
#endif


namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int MAX_TAPS = 27;

struct ConvArgs {
    const bf16_t* in;     // [T, H, W, Cin]
    const bf16_t* w;      // [Cout, Kpad]
    const bf16_t* bias;   // [Cout] or null
    const bf16_t* res;    // [T, H, W, Cout] or null
    bf16_t* out;          // [T, H, W, Cout]
    const bf16_t* zeros;  // >= 16 bytes of zeros
    int T, H, W, Cin, Cout, Kpad;   // Kpad: row stride of w (elements)
    int Kext;                       // K extent actually iterated (a multiple of 64, <= Kpad)
    int kT, kH, kW, ntaps;
    int q64, r64;         // 64 / Cin, 64 % Cin
    int replicate;        // 0: out-of-range taps read zeros; 1: coordinates are clamped (replicate padding)
    int Ho, Wo, sy, sx;   // output height / width and spatial stride (Ho = H, Wo = W, 1, 1 for the "same" convolutions)
    int py, px;           // zero rows / columns assumed above / left of the input ((k-1)/2 for "same"; 0 for the
                          // ZeroPad2d((0,1,0,1)) + stride-2 downsampling convolution of the VAE encoders)
    int up;               // 1: the input is read through a nearest 2x spatial upsample (H, W are the UPSAMPLED extents,
    int Hin, Win;         //    the stored image is Hin x Win = H/2 x W/2 and tap (y, x) reads pixel (y >> 1, x >> 1))
    // fused RMS norm of the OUTPUT (v2 tiles that hold all Cout channels of a position): out_norm = [silu](y_bf16 /
    // max(||y||, 1e-12) * sqrt(Cout) * gamma) from the bf16-rounded y = conv + bias (+ residual); out may be null
    const bf16_t* norm_gamma;
    bf16_t* out_norm;
    int norm_silu;
    // activation applied to conv + bias (+ residual) in f32 before the one bf16 rounding: 0 none, 1 leaky ReLU with
    // `act_slope` (0.0 = ReLU) — the TAEHV blocks (conv, act) and act(conv + skip)
    int act;
    float act_slope;
    // temporal stride (128x128 kernel only): output frame j reads input frames j * st + t0 + dt - (kT - 1); To output frames
    int To, st, t0;
    unsigned long long* prof;   // cycle-stamp buffer of ONE workgroup (tools/conv_prof.py; null in production)
    int torder;                 // slab kernels: temporal taps by input frame mod 3 + frame-fastest tile order (see slab_tap_of_step)
    int Tc;                     // > 0: the T frames are T / Tc independent CLIPS of Tc frames stacked along T (the spatial tiles of a
                                // tiled VAE decode in one launch): the causal temporal taps stop at a clip's first frame
    int dbg;                    // timing experiments on the prefetch kernel (WRONG results; tools/conv_ablate.py): 1 no weight DMA,
                                // 2 no slab DMA inside the loop
};
APEXMI_DEVICE float conv_act(float v, int act, float slope) { return (act && v < 0.0f) ? v * slope : v; }

// TO = storage type of out / residual: bf16_t in production; float in the f32-storage verification mode, where `in` is the
// exact three-way bf16 split of the float activations (channels [hi | mid | lo], apexmi_split_bf16x3) and the packed weight
// repeats every tap's Cin run three times, so the MFMA products are exact.
template <typename TO>
__global__ __launch_bounds__(256, 2) void conv3d_cl_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int tap_off[MAX_TAPS];  // packed (dt, dy, dx) as signed offsets

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    if (tid < a.ntaps) {
        const int dt = tid / (a.kH * a.kW), r = tid % (a.kH * a.kW);
        const int dy = r / a.kW, dx = r % a.kW;
        // causal in time (all padding on the left), symmetric "same" padding in space
        tap_off[tid] = ((dt - (a.kT - 1)) & 0xff) | (((dy - a.py) & 0xff) << 8) | (((dx - a.px) & 0xff) << 16);
    }
    __syncthreads();

    const int M = a.To * a.Ho * a.Wo;
    const int nm = (M + BM - 1) / BM;
    const int nn = (a.Cout + BN - 1) / BN;
    const int s = xcd_remap(blockIdx.x, nm * nn);
    const int pm = s / nn, pn = s % nn;  // consecutive workgroups share the activation rows
    const int m0 = pm * BM, n0 = pn * BN;

    // ---- per-lane A rows (output positions) and running (tap, ci) of the lane's 16-byte chunk ----
    int pos_t[4], pos_y[4], pos_x[4], a_tap[4], a_ci[4], pos_lo[4];
    const char* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = (i * 4 + wave) * 64 + lane;
        const int row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
        const int m = min(m0 + row, M - 1);
        pos_t[i] = (m / (a.Ho * a.Wo)) * a.st + a.t0;
        pos_lo[i] = a.Tc > 0 ? (pos_t[i] / a.Tc) * a.Tc : 0;    // first frame of this position's clip
        const int r = m % (a.Ho * a.Wo);
        pos_y[i] = (r / a.Wo) * a.sy;   // input row / column of tap (0, 0) before the pad offset
        pos_x[i] = (r % a.Wo) * a.sx;
        const int k = c * 8;
        a_tap[i] = k / a.Cin;
        a_ci[i] = k % a.Cin;
        w_src[i] = (const char*)(a.w + (int64_t)min(n0 + row, a.Cout - 1) * a.Kpad + c * 8);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nkt = a.Kext / BK;

    // Source of channel 0 of piece i's CURRENT tap (nullptr: the tap falls into the padding and reads the zero line).  It changes only
    // when the lane's 64-channel chunk wraps into the next tap — every Cin / 64 K-tiles — so the tap decode, the bounds tests and the
    // 64-bit address arithmetic (~40 VALU instructions a piece, four pieces per 16 MFMAs: the kernel was VALU-bound) run once per
    // tap instead of once per K-tile; in between a piece costs one add.
    auto tap_base = [&](int i) -> const bf16_t* {
        if (a_tap[i] >= a.ntaps) return nullptr;
        const int o = tap_off[a_tap[i]];
        const int ti = pos_t[i] + (int)(int8_t)(o & 0xff);
        const int yi = pos_y[i] + (int)(int8_t)((o >> 8) & 0xff);
        const int xi = pos_x[i] + (int)(int8_t)((o >> 16) & 0xff);
        if (a.replicate) {   // HunyuanVideo15CausalConv3d: F.pad(..., mode="replicate") == clamped coordinates
            const int tc = max(ti, pos_lo[i]), yc = min(max(yi, 0), a.H - 1), xc = min(max(xi, 0), a.W - 1);
            return a.in + ((int64_t)(tc * a.Hin + (yc >> a.up)) * a.Win + (xc >> a.up)) * a.Cin;
        }
        if (ti >= pos_lo[i] && ti < a.T && (unsigned)yi < (unsigned)a.H && (unsigned)xi < (unsigned)a.W)
            return a.in + ((int64_t)(ti * a.Hin + (yi >> a.up)) * a.Win + (xi >> a.up)) * a.Cin;
        return nullptr;
    };
    const bf16_t* tbase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) tbase[i] = tap_base(i);
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t* src = tbase[i] != nullptr ? tbase[i] + a_ci[i] : a.zeros;
            glds16(src, base + i * 4096);
            // advance this lane's chunk by one K-tile (64 channels)
            const int tap_was = a_tap[i];
            a_tap[i] += a.q64;
            a_ci[i] += a.r64;
            if (a_ci[i] >= a.Cin) {
                a_ci[i] -= a.Cin;
                a_tap[i] += 1;
            }
            if (a_tap[i] != tap_was) tbase[i] = tap_base(i);
        }
        const int64_t koff = (int64_t)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(w_src[i] + koff, base + TILE_BYTES + i * 4096);
    };

    int a_off[2], w_off[2], a_sw[2], w_sw[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ra = wm * 64 + t * 32 + l31;
        const int rw = wn * 64 + t * 32 + l31;
        a_off[t] = ra * 128;
        a_sw[t] = (ra >> 1) & 7;
        w_off[t] = rw * 128;
        w_sw[t] = (rw >> 1) & 7;
    }

    stage(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA is not covered by __syncthreads()
        __syncthreads();
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
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
    }

    // ---- epilogue: bias (+ residual) -> bf16, lane holds out[m][n .. n+3] per group.  Bias once per column group, every residual
    // load issued before the first store (out may alias res: the compiler cannot hoist them itself) ----
    float bsf[2][4][4], rsf[2][2][4][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bsf[nt][g][j] = 0.0f;
            if (a.bias != nullptr) load4<bf16_t>(a.bias + min(n0 + wn * 64 + nt * 32 + 8 * g + 4 * hi, a.Cout - 4), bsf[nt][g]);
        }
    const bool has_res = a.res != nullptr;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int j = 0; j < 4; ++j) rsf[mt][nt][g][j] = 0.0f;
                if (has_res) {
                    const int m = min(m0 + wm * 64 + mt * 32 + l31, M - 1);
                    load4<TO>((const TO*)a.res + (int64_t)m * a.Cout + min(n0 + wn * 64 + nt * 32 + 8 * g + 4 * hi, a.Cout - 4), rsf[mt][nt][g]);
                }
            }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + wm * 64 + mt * 32 + l31;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + nt * 32 + 8 * g + 4 * hi;
                if (m >= M || n >= a.Cout) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = conv_act(acc[nt][mt][4 * g + j] + bsf[nt][g][j] + rsf[mt][nt][g][j], a.act, a.act_slope);
                store4<TO>((TO*)a.out + (int64_t)m * a.Cout + n, v);
            }
    }
}

// ---- shared epilogue of the conv-shaped kernels: acc[NT][MT] 32x32 tiles of one wave, mrow[mt] = output position (row
// of the [positions, Cout] matrix) of this lane's row in m-tile mt or -1, wave (wm, wn) of a WMW x WNW wave grid, n0 = first
// output channel of the workgroup tile ------------------------------------------------------------------------------------
template <int MT, int NT, int WNW, int NORM>
APEXMI_DEVICE void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[NT][MT], const int (&mrow_)[MT], int n0, int wm, int wn,
                                 int l31, int hi, char* smem) {
    // Opaque copies of everything lane-derived: the epilogue's addresses are pure functions of the lane and the kernel arguments, so
    // hipcc computes them at kernel entry and carries them — a dozen 64-bit values — across the chunk loop, where the slab kernels
    // have no register to spare (spills).  The volatile statements keep their order against the loop's own (waits), i.e. stay here.
    int mrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        mrow[mt] = mrow_[mt];
        asm volatile("" : "+v"(mrow[mt]));
    }
    asm volatile("" : "+v"(l31), "+v"(hi));
    // ---- epilogue: bias (+ residual) -> bf16.  A lane holds out[m][nbase + 8 g + 4 hi + (0..3)], g = 0..3, per 32-column
    // tile; pairs of groups are exchanged with the other half-wave (v_permlane32_swap) into 8 CONSECUTIVE columns, so
    // residual loads and stores are 16 bytes per lane (the GEMM's epilogue trick).
    auto swap_pair = [](u32x2& x, u32x2& y) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            auto r = __builtin_amdgcn_permlane32_swap(x[i], y[i], false, false);
            x[i] = r[0];
            y[i] = r[1];
        }
    };
    float ssq[MT];
    if (!NORM && (a.Cout & 7) != 0) {   // Cout = 4 (conv_out, 3 channels + pad): 8-byte accesses, no exchange
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = mrow[mt];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn * (NT * 32) + nt * 32 + 8 * g + 4 * hi;
                    if (m < 0 || n >= a.Cout) continue;
                    float v[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) v[jj] = acc[nt][mt][4 * g + jj];
                    if (a.bias != nullptr) {
                        const u32x2 b = *(const u32x2*)(a.bias + n);
                        v[0] += bf16_lo(b[0]);
                        v[1] += bf16_hi(b[0]);
                        v[2] += bf16_lo(b[1]);
                        v[3] += bf16_hi(b[1]);
                    }
                    if (a.res != nullptr) {
                        const u32x2 r = *(const u32x2*)(a.res + (int64_t)m * a.Cout + n);
                        v[0] += bf16_lo(r[0]);
                        v[1] += bf16_hi(r[0]);
                        v[2] += bf16_lo(r[1]);
                        v[3] += bf16_hi(r[1]);
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) v[jj] = conv_act(v[jj], a.act, a.act_slope);
                    *(u32x2*)(a.out + (int64_t)m * a.Cout + n) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
                }
        }
        return;
    }
    // Loads first, all of them, then arithmetic and stores: out may alias res, so hipcc cannot move a residual load above an earlier
    // store itself, and with the loads inside the (mt, nt, pr) loops every iteration paid its own memory round trip (and re-read the
    // same bias for every m-tile) while the matrix pipe of the CU sat idle.  Bias stays packed (two dwords per 4-column group).
    // FULL (every 32-column n-tile of the wave lies inside Cout — all the layers of the shipped decoders): no column clamps or
    // guards, and one 64-bit row base per m-tile with compile-time column offsets; before, every 16-byte access rebuilt its address
    // with a 64-bit multiply-add behind a clamp and a lane-mask branch (~200 of the epilogue's instructions per lane).
    const int nb0 = n0 + wn * (NT * 32);
    auto body = [&](auto FULL_) {
        constexpr bool FULL = decltype(FULL_)::value;
        const int cg = nb0 + 4 * hi;                        // this lane's first column of a 4-column group (+ 8 g + 32 nt)
        const int cs = nb0 + 8 * hi;                        // ... of an 8-column store / residual segment (+ 16 pr + 32 nt)
        auto gcol = [&](int nt, int g) { return FULL ? cg + nt * 32 + 8 * g : min(cg + nt * 32 + 8 * g, a.Cout - 4); };
        u32x2 bsp[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) bsp[nt][g] = u32x2{0u, 0u};
        if (a.bias != nullptr) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) bsp[nt][g] = *(const u32x2*)(a.bias + gcol(nt, g));
        }
        int64_t rowb[MT];                                   // element offset of this lane's first store column in row mrow[mt]
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) rowb[mt] = (int64_t)max(mrow[mt], 0) * a.Cout + cs;
        // residual rows: all m-tiles up front (one m-tile at a time costs the fused-norm epilogue a second round trip, +2.3 us)
        constexpr int RMT = MT;
        u32x4 rr[RMT][NT][2];
        const bool has_res = a.res != nullptr;
        auto load_res = [&](int mt, int slot) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    if constexpr (FULL) rr[slot][nt][pr] = *(const u32x4*)(a.res + rowb[mt] + (nt * 32 + 16 * pr));
                    else rr[slot][nt][pr] = *(const u32x4*)(a.res + (int64_t)max(mrow[mt], 0) * a.Cout + max(min(cs + nt * 32 + 16 * pr, a.Cout - 8), 0));
                }
        };
        if (has_res) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) load_res(mt, mt);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            ssq[mt] = 0.0f;
            const int m = mrow[mt];
            const int rs = mt;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int nb = nb0 + nt * 32;
                    const int nst = nb + 8 * (2 * pr + hi);           // first of the 8 columns this lane loads / stores
                    if constexpr (!FULL) {
                        if (nb + 16 * pr >= a.Cout) continue;         // wave-uniform; lanes past Cout inside the pair store nothing
                    }
                    u32x2 ra = {0u, 0u}, rb = {0u, 0u};
                    if (has_res) {
                        ra = u32x2{rr[rs][nt][pr][0], rr[rs][nt][pr][1]};
                        rb = u32x2{rr[rs][nt][pr][2], rr[rs][nt][pr][3]};
                        swap_pair(ra, rb);                            // 16-byte row segment -> accumulator layout
                    }
                    u32x2 o[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int g = 2 * pr + q;
                        float v[4];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) v[jj] = acc[nt][mt][4 * g + jj];
                        const u32x2 b = bsp[nt][g];
                        v[0] += bf16_lo(b[0]);
                        v[1] += bf16_hi(b[0]);
                        v[2] += bf16_lo(b[1]);
                        v[3] += bf16_hi(b[1]);
                        const u32x2 r2 = q ? rb : ra;
                        v[0] += bf16_lo(r2[0]);
                        v[1] += bf16_hi(r2[0]);
                        v[2] += bf16_lo(r2[1]);
                        v[3] += bf16_hi(r2[1]);
                        if (!NORM) {
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) v[jj] = conv_act(v[jj], a.act, a.act_slope);
                        }
                        o[q][0] = pack_bf16(v[0], v[1]);
                        o[q][1] = pack_bf16(v[2], v[3]);
                        if (NORM) {   // the norm is taken over the STORED (bf16) values, as the separate pass reads them back
                            const float r0 = bf16_lo(o[q][0]), r1 = bf16_hi(o[q][0]), r2f = bf16_lo(o[q][1]), r3 = bf16_hi(o[q][1]);
                            acc[nt][mt][4 * g + 0] = r0;
                            acc[nt][mt][4 * g + 1] = r1;
                            acc[nt][mt][4 * g + 2] = r2f;
                            acc[nt][mt][4 * g + 3] = r3;
                            if (FULL || nb + 8 * g + 4 * hi < a.Cout) ssq[mt] += r0 * r0 + r1 * r1 + r2f * r2f + r3 * r3;
                        }
                    }
                    if (!NORM || a.out != nullptr) {
                        swap_pair(o[0], o[1]);
                        const u32x4 ov = u32x4{o[0][0], o[0][1], o[1][0], o[1][1]};
                        if constexpr (FULL) {
                            if (m >= 0) *(u32x4*)(a.out + rowb[mt] + (nt * 32 + 16 * pr)) = ov;
                        } else {
                            if (m >= 0 && nst < a.Cout) *(u32x4*)(a.out + (int64_t)m * a.Cout + nst) = ov;
                        }
                    }
                    // (without the guards' basic-block boundaries the scheduler overlaps all twelve segments of a lane and spills)
                    if constexpr (FULL) __builtin_amdgcn_sched_barrier(0);
                }
        }
        if constexpr (NORM) {
            // a row's channels: this lane's groups + those of lane ^ 32, and (WNW > 1) the other waves of the row through LDS
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) ssq[mt] = sum_xor32(ssq[mt]);
            if constexpr (WNW > 1) {
                float* red = (float*)smem;          // the staging buffers are free: every wave is past its last fragment read
                __syncthreads();
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (hi == 0) red[(wm * (MT * 32) + mt * 32 + l31) * WNW + wn] = ssq[mt];
                __syncthreads();
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float t = 0.0f;
#pragma unroll
                    for (int w = 0; w < WNW; ++w) t += red[(wm * (MT * 32) + mt * 32 + l31) * WNW + w];
                    ssq[mt] = t;
                }
            }
            const float root_c = sqrtf((float)a.Cout);
            u32x2 gmp[NT][4];                      // gamma, packed, once per channel group (it was re-read for every m-tile)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) gmp[nt][g] = *(const u32x2*)(a.norm_gamma + gcol(nt, g));
            auto second = [&](auto SILU) {         // the SiLU switch is block-uniform: decided once, not per 4-column group
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = mrow[mt];
                    const float scale = root_c / fmaxf(sqrtf(ssq[mt]), 1e-12f);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            const int nb = nb0 + nt * 32;
                            const int nst = nb + 8 * (2 * pr + hi);
                            if constexpr (!FULL) {
                                if (nb + 16 * pr >= a.Cout) continue;
                            }
                            u32x2 o[2];
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const int g = 2 * pr + q;
                                const u32x2 gm = gmp[nt][g];
                                float y[4] = {acc[nt][mt][4 * g + 0] * scale * bf16_lo(gm[0]), acc[nt][mt][4 * g + 1] * scale * bf16_hi(gm[0]),
                                              acc[nt][mt][4 * g + 2] * scale * bf16_lo(gm[1]), acc[nt][mt][4 * g + 3] * scale * bf16_hi(gm[1])};
                                if constexpr (decltype(SILU)::value) {
#pragma unroll
                                    for (int jj = 0; jj < 4; ++jj) y[jj] = silu_f(y[jj]);
                                }
                                o[q][0] = pack_bf16(y[0], y[1]);
                                o[q][1] = pack_bf16(y[2], y[3]);
                            }
                            swap_pair(o[0], o[1]);
                            const u32x4 ov = u32x4{o[0][0], o[0][1], o[1][0], o[1][1]};
                            if constexpr (FULL) {
                                if (m >= 0) *(u32x4*)(a.out_norm + rowb[mt] + (nt * 32 + 16 * pr)) = ov;
                            } else {
                                if (m >= 0 && nst < a.Cout) *(u32x4*)(a.out_norm + (int64_t)m * a.Cout + nst) = ov;
                            }
                            if constexpr (FULL) __builtin_amdgcn_sched_barrier(0);
                        }
                }
            };
            if (a.norm_silu) second(std::true_type{});
            else second(std::false_type{});
        }
    };
    // (the fused-norm instantiations keep the guarded path: their FULL form compiled to 256 VGPRs + ~30 spilled values inside the
    // epilogue and measured 6-7 us SLOWER per workgroup, profiles/r04_conv_tile_trace_c.log)
    if constexpr (NORM) {
        body(std::false_type{});
    } else {
        if ((a.Cout & 31) == 0 && nb0 + NT * 32 <= a.Cout) body(std::true_type{});
        else body(std::false_type{});
    }
}

// ---- v2: conv-shaped tiles ---------------------------------------------------------------------------------------
// The 128x128 kernel above is co-limited by LDS bandwidth (64x64 wave tiles: 4 fragment reads per 4 MFMAs) and wastes
// a quarter of the matrix work on the 96- and 192-channel stages that hold 3/4 of the Wan decode's FLOPs.  v2 keeps the
// same LDS image, swizzle and tap gather but uses the GEMM's proportions: 512 threads (two waves per SIMD, ONE
// workgroup per CU), wave tiles of MT x NT 32x32 MFMA tiles with more MFMAs than fragment reads (64x96: 5 reads per 6
// MFMAs; 128x64: 6 per 8), and a workgroup tile whose N extent is the layer's channel count:
//     Cout  96 : 512 x  96  (8 x 1 waves of  64 x 96)        Cout 192 / 384 : 256 x 192  (4 x 2 waves of 64 x 96)
//     Cout 256k: 256 x 256  (2 x 4 waves of 128 x 64)        Cout <= 32 (conv_out): 512 x 32   Cout 64 / 128: 512 x 64, 256 x 128
// The gather state is per LANE, not per chunk: all pieces of a lane share the K-chunk column (the swizzle term does not
// depend on the piece), so one (tap, ci) pair advances per K-tile; per piece only the pixel index of its output position
// and a 27-bit tap-validity mask are kept (computed once — the positions do not change along K).  Scope: stride 1, zero
// padding, optionally read through the nearest 2x upsample; everything else stays on the kernel above.
template <int WMW_, int WNW_, int MT_, int NT_>
struct ConvCfg {
    static constexpr int WMW = WMW_, WNW = WNW_, MT = MT_, NT = NT_;
    static constexpr int NW = WMW * WNW, NTHR = NW * 64;
    static constexpr int BM = WMW * MT * 32, BN = WNW * NT * 32;
    static constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
    static constexpr int A_LD = BM * 8 / NTHR;                  // 1 KiB pieces (8 rows) per wave per K-tile
    static constexpr int W_PIECES = BN / 8;                     // 1 KiB pieces of the weight tile
    static constexpr int W_LD = (W_PIECES + NW - 1) / NW;
    static_assert(NW == 8 && BM % 64 == 0 && BN % 32 == 0, "v2 is written for 8 waves (a 4-wave, one-wave-per-SIMD 256x96 tile measured 30 % slower without a hand-rotated pipeline)");
};

template <typename CFG, int UP, int NORM = 0>
__global__ __launch_bounds__(CFG::NTHR, 2) void conv3d_v2_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int tap_off[MAX_TAPS];
    constexpr int BM2 = CFG::BM, BN2 = CFG::BN, MT = CFG::MT, NT = CFG::NT, NW = CFG::NW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / CFG::WNW, wn = wave % CFG::WNW;
    const int l31 = lane & 31, hi = lane >> 5;

    if (tid < a.ntaps) {
        const int dt = tid / (a.kH * a.kW), r = tid % (a.kH * a.kW);
        const int dy = r / a.kW, dx = r % a.kW;
        tap_off[tid] = ((dt - (a.kT - 1)) & 0xff) | (((dy - a.py) & 0xff) << 8) | (((dx - a.px) & 0xff) << 16);
    }

    const int M = a.T * a.H * a.W;
    const int nn = (a.Cout + BN2 - 1) / BN2;
    const int nm = (M + BM2 - 1) / BM2;
    const int s = xcd_remap(blockIdx.x, nm * nn);
    const int pm = s / nn, pn = s % nn;
    const int m0 = pm * BM2, n0 = pn * BN2;

    // ---- gather state: byte offset of the piece's pixel + tap-validity mask per piece; (tap, ci) per lane ----
    // Activation pieces go through buffer_load ... lds: a 32-bit byte offset per lane and no 64-bit address math; an
    // out-of-range tap is given an offset past the descriptor's range, which reads as zeros (the padding).
    uint32_t off[CFG::A_LD];
    uint32_t vmask[CFG::A_LD];
#pragma unroll
    for (int i = 0; i < CFG::A_LD; ++i) {
        const int row = (i * NW + wave) * 8 + (lane >> 3);
        const int m = min(m0 + row, M - 1);
        const int t = m / (a.H * a.W);
        const int r = m % (a.H * a.W);
        const int y = r / a.W, x = r % a.W;
        off[i] = (uint32_t)((t * a.Hin + (y >> UP)) * a.Win + (x >> UP)) * (uint32_t)(a.Cin * 2);
        uint32_t mk = 0;
        for (int tap = 0; tap < a.ntaps; ++tap) {
            const int dt = tap / (a.kH * a.kW) - (a.kT - 1), rr = tap % (a.kH * a.kW);
            const int dy = rr / a.kW - a.py, dx = rr % a.kW - a.px;
            const bool ok = (unsigned)(t + dt) < (unsigned)a.T && (unsigned)(y + dy) < (unsigned)a.H &&
                            (unsigned)(x + dx) < (unsigned)a.W;
            mk |= (ok ? 1u : 0u) << tap;
        }
        vmask[i] = mk | ((uint32_t)(y & 1) << 27) | ((uint32_t)(x & 1) << 28);
        if (a.replicate) {   // replicate padding clamps coordinates per tap: keep (t, y, x) instead of an offset and a mask
            off[i] = (uint32_t)t;
            vmask[i] = ((uint32_t)y << 16) | (uint32_t)x;
        }
    }
    auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((int64_t)a.T * a.Hin * a.Win * a.Cin * 2), 0x00020000);
    const int cchunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);   // (p & 7) ^ ((row >> 1) & 7), same for every piece
    int a_tap = (cchunk * 8) / a.Cin, a_ci = (cchunk * 8) % a.Cin;
    const char* w_src[CFG::W_LD];
#pragma unroll
    for (int i = 0; i < CFG::W_LD; ++i) {
        const int row = (i * NW + wave) * 8 + (lane >> 3);
        w_src[i] = (const char*)(a.w + (int64_t)min(n0 + row, a.Cout - 1) * a.Kpad + cchunk * 8);
    }
    __syncthreads();   // tap_off

    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nkt = a.Kext / BK;

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * CFG::STAGE + wave * 1024;
        int delta = 0, dy = 0, dx = 0;
        uint32_t bit = 0;
        if (a_tap < a.ntaps) {
            const int o = tap_off[a_tap];
            const int dt = (int)(int8_t)(o & 0xff);
            dy = (int)(int8_t)((o >> 8) & 0xff);
            dx = (int)(int8_t)((o >> 16) & 0xff);
            int dpix = dt * a.Hin * a.Win;
            if (!UP) dpix += dy * a.Win + dx;
            delta = (dpix * a.Cin + a_ci) * 2;
            bit = 1u << a_tap;
        }
        if (a.replicate) {   // wave-uniform: HunyuanVideo15CausalConv3d, F.pad(mode="replicate") == clamped tap coordinates
            int dt = 0;
            if (a_tap < a.ntaps) dt = (int)(int8_t)(tap_off[a_tap] & 0xff);
#pragma unroll
            for (int i = 0; i < CFG::A_LD; ++i) {
                const int tc = max((int)off[i] + dt, 0);
                const int yc = min(max((int)(vmask[i] >> 16) + dy, 0), a.H - 1);
                const int xc = min(max((int)(vmask[i] & 0xffffu) + dx, 0), a.W - 1);
                uint32_t o = (uint32_t)(((tc * a.H + yc) * a.W + xc) * a.Cin + a_ci) * 2u;
                if (a_tap >= a.ntaps) o = 0x80000000u;      // K padding past the last tap: zeros
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (__attribute__((address_space(3))) void*)(base + i * (NW * 1024)), 16,
                                                         (int)o, 0, 0, 0);
            }
        } else
#pragma unroll
        for (int i = 0; i < CFG::A_LD; ++i) {
            uint32_t o = off[i] + (uint32_t)delta;
            if (UP) {   // stored pixel of upsampled (y + dy, x + dx): ((y & 1) + dy) >> 1 rows below (y >> 1), same in x
                const int d = (((int)((vmask[i] >> 27) & 1u) + dy) >> 1) * a.Win + (((int)((vmask[i] >> 28) & 1u) + dx) >> 1);
                o += (uint32_t)(d * a.Cin * 2);
            }
            o = (vmask[i] & bit) ? o : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (__attribute__((address_space(3))) void*)(base + i * (NW * 1024)), 16,
                                                     (int)o, 0, 0, 0);
        }
        a_tap += a.q64;
        a_ci += a.r64;
        if (a_ci >= a.Cin) {
            a_ci -= a.Cin;
            a_tap += 1;
        }
        const int64_t koff = (int64_t)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < CFG::W_LD; ++i)
            if ((i + 1) * NW <= CFG::W_PIECES || i * NW + wave < CFG::W_PIECES)   // wave-uniform
                glds16(w_src[i] + koff, base + CFG::A_BYTES + i * (NW * 1024));
    };

    int a_off[MT], a_sw[MT], w_off[NT], w_sw[NT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int r = wm * (MT * 32) + t * 32 + l31;
        a_off[t] = r * 128;
        a_sw[t] = (r >> 1) & 7;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int r = wn * (NT * 32) + t * 32 + l31;
        w_off[t] = r * 128;
        w_sw[t] = (r >> 1) & 7;
    }

    stage(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
        const char* As = smem + (kt & 1) * CFG::STAGE;
        const char* Ws = As + CFG::A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = ks * 2 + hi;
            bf16x8 af[MT], wf[NT];
#pragma unroll
            for (int t = 0; t < MT; ++t) af[t] = *(const bf16x8*)(As + a_off[t] + ((c ^ a_sw[t]) << 4));
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[t] = *(const bf16x8*)(Ws + w_off[t] + ((c ^ w_sw[t]) << 4));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
    }

    int mrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + wm * (MT * 32) + mt * 32 + l31;
        mrow[mt] = m < M ? m : -1;
    }
    conv_epilogue<MT, NT, CFG::WNW, NORM>(a, acc, mrow, n0, wm, wn, l31, hi, smem);
}

// ---- direct ("slab") convolution, Cin = 96 ---------------------------------------------------------------------------------
// The implicit GEMM above gathers every input element once per TAP through the LDS-DMA path (27 times for 3x3x3), and on the
// 96-channel layers that gather, not the matrix pipe, is 3/4 of the time (conv_out, a tenth of the MFMA work on the same
// gather, takes 3.7 ms where 96 -> 96 takes 4.95).  Here a workgroup owns an 8 x 32 block of output positions of one frame
// and stages, per temporal tap, the haloed input slab (10 x 34 positions x 96 channels = 64 KiB) ONCE; the nine spatial taps
// are shifted fragment reads inside LDS.  Weights stream through a 3-deep ring of half-tap chunks (96 rows x 48 channels).
//   wave w = output row y0 + w, lane row l31 = column x0 + l31: a tap (dy, dx) reads slab position p = (w + 1 + dy) * 34 +
//   (l31 + 1 + dx) — consecutive in the lane, so with the chunk ROTATION s = (c + ((p >> 2) & 3)) mod 12 (192-byte position
//   pitch: 4-dword slot = 4 * ((-p) & 3) + s) every 16-lane group of a ds_read_b128 covers all 16 slots for every shift;
//   weight rows (96-byte pitch) use chunk ^ ((n >> 3) & 1).
// Scope: Cin == 96, 3x3 spatial taps with "same" zero padding, kT <= 3 causal, stride 1, Cout <= 96, no upsample.
constexpr int SLAB_BYTES = 65536, SLAB_WCH = 9216, SLAB_LDS = 2 * SLAB_BYTES + 3 * SLAB_WCH;

template <int NT, int NORM>
__global__ __launch_bounds__(512, 2) void conv3d_slab96_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TH = 8, TW = 32, SC = TW + 2, SPOS = (TH + 2) * SC;
    constexpr int WP = 3 * NT;                         // 1 KiB pieces of a weight chunk (NT * 32 rows x 96 B)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;

    const int ntx = (a.W + TW - 1) / TW, nty = (a.H + TH - 1) / TH;
    const int b = xcd_remap(blockIdx.x, a.T * nty * ntx);
    const int tx = b % ntx, ty = (b / ntx) % nty, t = b / (ntx * nty);
    const int y0 = ty * TH, x0 = tx * TW;
    const int nk = min(a.kT, t + 1);                   // temporal taps that touch data (frames t - nk + 1 .. t)
    const int kt_first = a.kT - nk;
    const uint32_t frame_bytes = (uint32_t)(a.H * a.W) * 192u;

    // ---- slab gather state: piece q = j * 8 + wave covers LDS bytes [q KiB, +1 KiB) = 64 slots of 16 B ----
    uint32_t soff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int L = (j * 8 + wave) * 64 + lane;
        const int p = L / 12, sl = L - p * 12;
        int c = sl - ((p >> 2) & 3);
        if (c < 0) c += 12;
        const int r = p / SC, cx = p - r * SC;
        const int yy = y0 - 1 + r, xx = x0 - 1 + cx;
        const bool ok = p < SPOS && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        soff[j] = ok ? (uint32_t)((yy * a.W + xx) * 192 + c * 16) : 0x80000000u;
    }
    auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((int64_t)a.T * a.H * a.W * 192), 0x00020000);
    auto slab_piece = [&](int sb, int f, int j) {
        uint32_t o = soff[j];
        if (!(o & 0x80000000u)) o += (uint32_t)f * frame_bytes;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (__attribute__((address_space(3))) void*)(smem + sb * SLAB_BYTES + (j * 8 + wave) * 1024),
                                                 16, (int)o, 0, 0, 0);
    };
    // ---- weight chunk pieces: piece q (row n = L / 6, slot L % 6 holds chunk slot ^ ((n >> 3) & 1)) ----
    const char* wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int L = (i == 0 ? wave : 8) * 64 + lane;
        const int n = L / 6, sl = L - n * 6;
        wsrc[i] = (const char*)(a.w + (int64_t)min(n, a.Cout - 1) * a.Kpad + ((sl ^ ((n >> 3) & 1)) * 8));
    }
    const int nchunks = nk * 18;
    auto w_chunk = [&](int ci) {      // chunk ci = (temporal tap ki, spatial tap sp, channel half h)
        const int ki = ci / 18, rem = ci - ki * 18;
        const int64_t kbase = (int64_t)(((kt_first + ki) * 9 + (rem >> 1)) * 96 + 48 * (rem & 1)) * 2;
        char* dst = smem + 2 * SLAB_BYTES + (ci % 3) * SLAB_WCH;
        if (wave < (WP < 8 ? WP : 8)) glds16(wsrc[0] + kbase, dst + wave * 1024);
        if (WP == 9 && wave == 0) glds16(wsrc[1] + kbase, dst + 8 * 1024);
    };
    const int nw = (wave < (WP < 8 ? WP : 8) ? 1 : 0) + ((WP == 9 && wave == 0) ? 1 : 0);   // this wave's loads per chunk

    f32x16 acc[NT][1];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.0f;

    // ---- prologue: first slab, chunks 0 and 1 ----
#pragma unroll
    for (int j = 0; j < 8; ++j) slab_piece(0, t - nk + 1, j);
    w_chunk(0);
    w_chunk(1);          // nchunks >= 18

    int ki = 0, jl = 0;  // temporal tap index and iteration inside it
    bool prev_slab = false;
    for (int it = 0; it < nchunks; ++it) {
        // loads issued after chunk `it`: the slab piece of the previous iteration and chunk it + 1
        const int k = (prev_slab ? 1 : 0) + (it + 1 < nchunks ? nw : 0);
        switch (k) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        }
        // a bare s_barrier: __syncthreads() carries a fence that drains vmcnt to 0, i.e. waits for the prefetches too
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        prev_slab = ki + 1 < nk && jl < 8;
        if (prev_slab) slab_piece((ki + 1) & 1, t - nk + 1 + ki + 1, jl);
        if (it + 2 < nchunks) w_chunk(it + 2);

        const int sp = jl >> 1, h = jl & 1;
        const int dy = sp / 3, dx = sp - dy * 3;            // 0..2 = tap offset + 1
        const int p = (wave + dy) * SC + (l31 + dx);
        const int rot = (p >> 2) & 3;
        const char* As = smem + (ki & 1) * SLAB_BYTES + p * 192;
        const char* Ws = smem + 2 * SLAB_BYTES + (it % 3) * SLAB_WCH;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            int sa = 6 * h + 2 * ks + hi + rot;
            if (sa >= 12) sa -= 12;
            const bf16x8 af = *(const bf16x8*)(As + sa * 16);
            bf16x8 wf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = nt * 32 + l31;
                wf[nt] = *(const bf16x8*)(Ws + n * 96 + (((2 * ks + hi) ^ ((n >> 3) & 1)) << 4));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], af, acc[nt][0], 0, 0, 0);
        }
        if (++jl == 18) {
            jl = 0;
            ++ki;
        }
    }

    int mrow[1];
    const int y = y0 + wave, x = x0 + l31;
    mrow[0] = (y < a.H && x < a.W) ? (t * a.H + y) * a.W + x : -1;
    conv_epilogue<1, NT, 1, NORM>(a, acc, mrow, 0, wave, 0, l31, hi, smem);
}


// General form of the slab kernel (shipped): Cin = 48 * S channels are walked in S slices of 48 (96-byte slab pitch, chunk ^
// ((p >> 3) & 1) — the same swizzle as the weight rows), the slab of one (temporal tap, slice) is double buffered and the
// loop runs temporal tap -> slice -> spatial tap, so the f32 accumulation order differs from the implicit GEMM's (tap ->
// channel): the same values up to f32 summation order (1 bf16 ulp on ~3e-4 of the outputs).  Two shapes:
//   MT = 2: 16 x 32 positions per workgroup, two rows per wave (every weight fragment feeds two MFMAs), Cout <= 96 (NT <= 3)
//   MT = 1:  8 x 32 positions, one row per wave, 192 output channels per workgroup (NT = 6; blockIdx.y = N tile for Cout 384)
// UP (runtime): the slab is read through the nearest 2x upsample (stored pixel (y >> 1, x >> 1)).
// SLW = channels per slice: 48 (96-byte pitch, 6 chunks, swizzle c ^ ((p >> 3) & 1), 3 k-steps) or 64 (128-byte pitch, 8 chunks,
// c ^ ((p >> 1) & 7): 4-dword slot = 8 (p & 1) + swizzled chunk, 4 k-steps) — the latter for the channel counts of the
// HunyuanVideo-1.5 / Flux / TAEHV decoders (64 .. 1024); a.replicate: clamped slab coordinates (F.pad(mode="replicate")).
// Temporal-tap order of the slab kernels (conv.torder = 1): the nk taps of output frame t are visited in the order of their
// INPUT frame index mod 3, not oldest first.  Workgroups of the same spatial tile and consecutive frames run side by side on one XCD
// (frame-fastest tile order below) and each needs the frames {t-2, t-1, t}: visiting "the frame = j mod 3" at step j makes the
// three of them read the SAME input slab at the same time, so two of the three reads are L2 hits instead of fabric fetches
// (rocprofv3: 3.37 GB fetched per 96->96 launch for a 1.02 GB input before).  Only the f32 summation order over the temporal taps
// changes (it now depends on t mod 3).  step j -> tap index k in [0, nk).
APEXMI_DEVICE int slab_tap_of_step(int j, int t, int nk, int torder) {
    if (!torder || nk != 3) return j;
    int k = (j + 2 - t) % 3;
    return k < 0 ? k + 3 : k;
}

template <int SLW>
APEXMI_DEVICE int slab_swz(int c, int p) { return SLW == 48 ? (c ^ ((p >> 3) & 1)) : (c ^ ((p >> 1) & 7)); }

template <int NT, int MT, int NORM, int SLW = 48>
__global__ __launch_bounds__(512, 2) void conv3d_slab_kernel(const ConvArgs a) {
    constexpr int CPP = SLW / 8, PITCH = SLW * 2, KS = SLW / 16;   // chunks per position, bytes per position, k-steps per chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TH = 8 * MT, TW = 32, SC = TW + 2, SPOS = (TH + 2) * SC;
    constexpr int NPJ = (SPOS * PITCH + 8191) / 8192;    // slab pieces per 8 waves (8 for MT = 2, 4 for MT = 1 at SLW 48)
    constexpr int SLABB = NPJ * 8192;
    constexpr int WCH = NT * 32 * PITCH, WP = WCH / 1024;   // weight chunk: NT * 32 rows x PITCH bytes
    constexpr int WLD = (WP + 3) / 4;                    // weight pieces per issuing wave (waves 0..3)
    CONV_TRACE_DECL();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;

    const int ntx = (a.W + TW - 1) / TW, nty = (a.H + TH - 1) / TH;
    const int b = xcd_remap(blockIdx.x, a.T * nty * ntx);
    int tx, ty, t;
    if (a.torder && a.kT == 3) {   // frame fastest: the same tile of consecutive frames runs side by side (they share input frames)
        t = b % a.T;
        const int tile = b / a.T;
        tx = tile % ntx, ty = tile / ntx;
    } else {
        tx = b % ntx, ty = (b / ntx) % nty, t = b / (ntx * nty);
    }
    const int y0 = ty * TH, x0 = tx * TW;
    const int n0 = blockIdx.y * (NT * 32);
    const int S = a.Cin / SLW;
    const int nk = a.replicate ? a.kT : min(a.kT, t + 1);   // replicate: frames before the first repeat it
    const int kt_first = a.kT - nk;
    const int nph = nk * S;                            // phases = (temporal tap, channel slice)
    const uint32_t pos_bytes = (uint32_t)a.Cin * 2u;
    const uint32_t frame_bytes = (uint32_t)(a.Hin * a.Win) * pos_bytes;

    // DMA roles are split by wave, because vmcnt counts a wave's loads IN ORDER: a wait for a (fast, L2-resident) weight
    // piece would also wait for every older (slow, HBM-latency) slab piece of the same wave.  Waves 0..3 issue only weight
    // pieces and wait for them every chunk; waves 4..7 issue only slab pieces (2 * NPJ each per phase) and wait ONCE per
    // phase, a whole phase after issuing them.
    const bool loader = wave >= 4;
    uint32_t soff[2 * NPJ];
#pragma unroll
    for (int j = 0; j < 2 * NPJ; ++j) {
        const int L = (j * 4 + (wave & 3)) * 64 + lane;
        const int p = L / CPP, sl = L - p * CPP;
        const int c = slab_swz<SLW>(sl, p);
        const int r = p / SC, cx = p - r * SC;
        int yy = y0 - 1 + r, xx = x0 - 1 + cx;
        if (a.replicate) {
            yy = min(max(yy, 0), a.H - 1);
            xx = min(max(xx, 0), a.W - 1);
        }
        const bool ok = p < SPOS && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        soff[j] = ok ? (uint32_t)((yy >> a.up) * a.Win + (xx >> a.up)) * pos_bytes + (uint32_t)(c * 16) : 0x80000000u;
    }
    auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((int64_t)a.T * a.Hin * a.Win * a.Cin * 2), 0x00020000);
    // (no runtime divisions in the loop: phase / chunk coordinates are carried as counters)
    auto slab_piece = [&](int buf, uint32_t phase_off, int j) {     // phase_off = frame * frame_bytes + slice * 96
        uint32_t o = soff[j];
        if (!(o & 0x80000000u)) o += phase_off;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (__attribute__((address_space(3))) void*)(smem + buf * SLABB + (j * 4 + (wave & 3)) * 1024), 16,
                                                 (int)o, 0, 0, 0);
    };
    const char* wsrc[WLD];
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
        const int L = (i * 4 + (wave & 3)) * 64 + lane;
        const int n = L / CPP, sl = L - n * CPP;
        wsrc[i] = (const char*)(a.w + (int64_t)min(n0 + n, a.Cout - 1) * a.Kpad + (slab_swz<SLW>(sl, n) * 8));
    }
    // A phase = (temporal tap, channel slice) = nine chunks, one per spatial tap, UNROLLED below: the tap (dy, dx), the weight-ring
    // slot (9 = 0 mod 3: chunk sp of every phase sits in slot sp % 3), which chunks carry slab pieces and the issue cursor's tap are
    // compile-time constants, so what is left per chunk is the work itself (round 3 counted 105 SALU + 58 VALU instructions and 15
    // branches of bookkeeping per chunk beside 18 MFMAs).  Same order of loads, waits, reads and MFMAs as before: same bits.
    auto w_issue = [&](int slot_, int64_t kbase) {       // the weight chunk at byte offset kbase of every row -> ring slot slot_
        char* dst = smem + 2 * SLABB + slot_ * WCH;
#pragma unroll
        for (int i = 0; i < WLD; ++i)
            if ((i + 1) * 4 <= WP || i * 4 + wave < WP) glds16(wsrc[i] + kbase, dst + (i * 4 + wave) * 1024);   // wave-uniform
    };
    int w_sl = 0, w_step = 0;                             // (slice, temporal step) of the phase AFTER the current one
    auto w_phase_base = [&]() -> int64_t {
        return (int64_t)(kt_first + slab_tap_of_step(min(w_step, nk - 1), t, nk, a.torder)) * 9 * a.Cin * 2 + w_sl * PITCH;
    };
    int64_t wk_cur = w_phase_base();
    auto w_phase_advance = [&]() {
        if (++w_sl == S) {
            w_sl = 0;
            ++w_step;
        }
    };
    w_phase_advance();
    int64_t wk_next = w_phase_base();
    int nw = 0;                                        // this wave's loads per weight chunk (0 for the slab loaders)
#pragma unroll
    for (int i = 0; i < WLD; ++i) nw += (!loader && ((i + 1) * 4 <= WP || i * 4 + wave < WP)) ? 1 : 0;

    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][m][r] = 0.0f;

    // next-phase cursor for the slab prefetch
    int np_step = 0;
    int np_f = t - nk + 1 + slab_tap_of_step(0, t, nk, a.torder);   // frame of the next phase (may be < 0 under replicate padding: clamped)
    uint32_t np_off = (uint32_t)max(np_f, 0) * frame_bytes;
    int np_sl = 0;
    if (loader) {
#pragma unroll
        for (int j = 0; j < 2 * NPJ; ++j) slab_piece(0, np_off, j);
    }
    auto advance_phase = [&]() {
        if (++np_sl == S) {
            np_sl = 0;
            ++np_step;
            np_f = t - nk + 1 + slab_tap_of_step(min(np_step, nk - 1), t, nk, a.torder);
            np_off = (uint32_t)max(np_f, 0) * frame_bytes;
        } else {
            np_off += (uint32_t)PITCH;
        }
    };
    advance_phase();
    if (!loader) {
        w_issue(0, wk_cur);
        w_issue(1, wk_cur + (int64_t)(a.Cin * 2));
    }
    auto wait_weights = [&](bool last) {               // at most this wave's pieces of the NEXT chunk may still be in flight
        switch (last ? 0 : nw) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        }
    };
    int pa0[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) pa0[m] = (MT * wave + m) * SC + l31;

    CONV_TRACE_AT(tr_l0);
    for (int ph = 0; ph < nph; ++ph) {
        const char* Sb = smem + (ph & 1) * SLABB;
        const bool last_ph = ph + 1 == nph;
        auto chunk = [&](auto SPC) {
            constexpr int sp = decltype(SPC)::value;
            constexpr int dy = sp / 3, dx = sp - dy * 3;
            if (loader) {
                // the slab of this phase was issued during the previous phase (or the prologue): wait for it once, at the phase's
                // first chunk; its pieces had a whole phase to land
                if constexpr (sp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                wait_weights(sp == 8 && last_ph);
            }
            // a bare s_barrier: __syncthreads() carries a fence that drains vmcnt to 0, i.e. waits for the prefetches too
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const char* Ws = smem + 2 * SLABB + (sp % 3) * WCH;
            int pa[MT];
            // (with a compile-time tap the fragment offsets of all nine taps are loop-invariant and hipcc hoists them out of the phase
            // loop: up to 256 VGPRs, no spill in any instantiation, and measured FASTER than recomputing them per chunk — HunyuanVideo-1.5
            // decode 2265 vs 2294 ms; the prefetch kernel below, which has no registers to spare, keeps them opaque instead)
            const int lrow = l31;
#pragma unroll
            for (int m = 0; m < MT; ++m) pa[m] = pa0[m] + (dy * SC + dx);
            // order inside a chunk: all fragment reads -> this wave's DMA issues (their ~150 cycles apiece hide the LDS latency)
            // -> the MFMAs, which then drain underneath the next chunk's wait / barrier / reads
            bf16x8 af[KS][MT], wf[KS][NT];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int m = 0; m < MT; ++m) af[ks][m] = *(const bf16x8*)(Sb + pa[m] * PITCH + (slab_swz<SLW>(2 * ks + hi, pa[m]) << 4));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    wf[ks][nt] = *(const bf16x8*)(Ws + (nt * 32 + lrow) * PITCH + (slab_swz<SLW>(2 * ks + hi, nt * 32 + lrow) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (loader) {     // next phase's slab: two pieces per chunk over the first NPJ chunks of this phase
                if constexpr (sp < NPJ) {
                    if (!last_ph) {
                        slab_piece((ph + 1) & 1, np_off, 2 * sp);
                        slab_piece((ph + 1) & 1, np_off, 2 * sp + 1);
                    }
                }
            } else if (!(last_ph && sp >= 7)) {         // the chunk two ahead: spatial tap (sp + 2) % 9 of this phase or of the next
                w_issue((sp + 2) % 3, (sp < 7 ? wk_cur : wk_next) + (int64_t)((sp + 2) % 9) * (a.Cin * 2));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][nt], af[ks][m], acc[nt][m], 0, 0, 0);
        };
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
        chunk(std::integral_constant<int, 2>{});
        chunk(std::integral_constant<int, 3>{});
        chunk(std::integral_constant<int, 4>{});
        chunk(std::integral_constant<int, 5>{});
        chunk(std::integral_constant<int, 6>{});
        chunk(std::integral_constant<int, 7>{});
        chunk(std::integral_constant<int, 8>{});
        advance_phase();
        wk_cur = wk_next;
        w_phase_advance();
        wk_next = w_phase_base();
    }

    CONV_TRACE_AT(tr_l1);
    int mrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int y = y0 + MT * wave + m, x = x0 + l31;
        mrow[m] = (y < a.H && x < a.W) ? (t * a.H + y) * a.W + x : -1;
    }
    conv_epilogue<MT, NT, 1, NORM>(a, acc, mrow, n0, wave, 0, l31, hi, smem);
    CONV_TRACE_END();
}

// Register-prefetch form of the slab kernel (shipped, conv.pp = 1).  Same slab / weight-chunk images, same (temporal tap ->
// slice -> spatial tap) accumulation order and therefore the same bits as conv3d_slab_kernel; what changes is WHEN things
// are issued.  Measured on the kernel above (s_memtime stamps, profiles/r03_vae_conv_*): a wave's chunk is a serial chain
// barrier -> 15 fragment reads (~520 cycles until the last returns) -> DMA issue (~190 cycles a piece) -> 18 MFMAs (576),
// ~2060 cycles per chunk against 1152 cycles of matrix work per SIMD — LDS latency and DMA issue are exposed once per
// chunk in every wave, and skewing the two waves of a SIMD against each other does not shorten either wave's chain.  Here
//   * the fragments of chunk c + 1 are read (into a second register set) at the START of interval c, so their latency runs
//     under the MFMAs of chunk c; the weight ring is 4 deep so that chunk c + 1 is already visible at barrier c;
//   * every wave carries an equal share of the DMA pieces (slab and weights) and issues them BETWEEN k-step groups of its
//     MFMAs — the two waves of a SIMD at different groups (wave < 4 after the first, wave >= 4 after the last but one), so
//     one of them always has matrix work while the other sits in the DMA issue;
//   * the wait before the barrier is vmcnt(pieces issued in this interval): in-order retirement then guarantees chunk c + 2's
//     weights (issued an interval ago) and, from spatial tap 7 on, the whole next slab.
// The 192-channel stages run as 4 row pairs x 2 channel halves (WNW = 2: NT = 3, MT = 2 per wave — 15 fragment reads per
// 18 MFMAs instead of the 21 of the one-row form), the 64-channel-slice stages as 4 row pairs x 2 halves of 64.
template <int NT, int MT, int WNW, int NORM, int SLW = 48, int NW = 8>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) void conv3d_slabp_kernel(const ConvArgs a) {
    constexpr int CPP = SLW / 8, PITCH = SLW * 2, KS = SLW / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RG = NW / WNW;                              // row groups (waves along M)
    constexpr int TH = RG * MT, TW = 32, SC = TW + 2, SPOS = (TH + 2) * SC;
    constexpr int NSP = (SPOS * PITCH + 1023) / 1024;         // 1 KiB slab pieces
    constexpr int NPJ = (NSP + NW - 1) / NW;                  // ... per wave
    constexpr int SPT = (NPJ + 6) / 7;                        // ... per wave and spatial tap (taps 0..6 carry the next slab)
    constexpr int SLABB = NSP * 1024;
    constexpr int WROWS = WNW * NT * 32;                      // weight rows per chunk
    constexpr int WCH = WROWS * PITCH, WP = WCH / 1024;       // weight chunk bytes / pieces
    constexpr int WLN = (WP + NW - 1) / NW;                   // weight pieces per wave and chunk
    static_assert(NPJ <= 16 && WLN <= 5 && SPT <= 3, "piece tables");
    CONV_TRACE_DECL();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WNW, wn = wave - wr * WNW;
    const int l31 = lane & 31, hi = lane >> 5;
    const bool late = NW == 8 && wave >= 4;                    // the SIMD partner of wave - 4: takes its DMA stall later

    const int ntx = (a.W + TW - 1) / TW, nty = (a.H + TH - 1) / TH;
    const int b = xcd_remap(blockIdx.x, a.T * nty * ntx);
    int tx, ty, t;
    if (a.torder && a.kT == 3) {   // frame fastest: the same tile of consecutive frames runs side by side (they share input frames)
        t = b % a.T;
        const int tile = b / a.T;
        tx = tile % ntx, ty = tile / ntx;
    } else {
        tx = b % ntx, ty = (b / ntx) % nty, t = b / (ntx * nty);
    }
    const int y0 = ty * TH, x0 = tx * TW;
    const int n0 = blockIdx.y * WROWS;
    const int S = a.Cin / SLW;
    const int nk = a.replicate ? a.kT : min(a.kT, t + 1);
    const int kt_first = a.kT - nk;
    const int nph = nk * S;
    const int nchunks = nph * 9;
    const uint32_t pos_bytes = (uint32_t)a.Cin * 2u;
    const uint32_t frame_bytes = (uint32_t)(a.Hin * a.Win) * pos_bytes;

    // ---- DMA tables: slab piece q = NW j + wave, weight piece q = NW i + wave (1 KiB = 64 lanes x 16 bytes each)
    uint32_t soff[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        soff[j] = 0x80000000u;
        if (j < NPJ) {
            const int L = (j * NW + wave) * 64 + lane;
            const int p = L / CPP, sl = L - p * CPP;
            const int c = slab_swz<SLW>(sl, p);
            const int r = p / SC, cx = p - r * SC;
            int yy = y0 - 1 + r, xx = x0 - 1 + cx;
            if (a.replicate) {
                yy = min(max(yy, 0), a.H - 1);
                xx = min(max(xx, 0), a.W - 1);
            }
            const bool ok = p < SPOS && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
            if (ok) soff[j] = (uint32_t)((yy >> a.up) * a.Win + (xx >> a.up)) * pos_bytes + (uint32_t)(c * 16);
        }
    }
    auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((int64_t)a.T * a.Hin * a.Win * a.Cin * 2), 0x00020000);
    auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)((int64_t)a.Cout * a.Kpad * 2), 0x00020000);
    int wvoff[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 5; ++i)
        if (i < WLN) {
            const int L = (i * NW + wave) * 64 + lane;
            const int n = L / CPP, sl = L - n * CPP;
            wvoff[i] = min(n0 + n, a.Cout - 1) * (a.Kpad * 2) + slab_swz<SLW>(sl, n) * 16;
        }

    auto slab_piece = [&](int buf, uint32_t phase_off, uint32_t o, int j) {
        if (!(o & 0x80000000u)) o += phase_off;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (__attribute__((address_space(3))) void*)(smem + buf * SLABB + (j * NW + wave) * 1024), 16,
                                                 (int)o, 0, 0, 0);
    };
    // Weight chunks: byte offset of a PHASE's (temporal tap, slice) inside a weight row; the nine spatial taps of the phase follow at
    // Cin * 2 bytes each.  wk_cur = the phase whose MFMAs run, wk_next = the one after it.
    int w_sl = 0, w_step = 0;
    auto w_phase_base = [&]() -> int {
        return (kt_first + slab_tap_of_step(min(w_step, nk - 1), t, nk, a.torder)) * 9 * a.Cin * 2 + w_sl * PITCH;
    };
    auto w_phase_advance = [&]() {
        if (++w_sl == S) {
            w_sl = 0;
            ++w_step;
        }
    };
    int wk_cur = w_phase_base();
    w_phase_advance();
    int wk_next = w_phase_base();
    // which weight pieces (bit i: piece i NW + wave exists) and which slab pieces (bit j) this wave carries, and whether it is of the
    // early half: plain scalar integers tested with one compare each (as `bool`s combined with && they became lane masks that hipcc
    // rebuilt through v_cndmask / v_cmp at every piece of every interval)
    int wmask_ = 0, smask_ = 0;
#pragma unroll
    for (int i = 0; i < WLN; ++i) wmask_ |= (i * NW + wave < WP) ? (1 << i) : 0;
#pragma unroll
    for (int j = 0; j < NPJ; ++j) smask_ |= (j * NW + wave < NSP) ? (1 << j) : 0;
#ifdef APEXMI_DEBUG
    if (a.dbg & 1) wmask_ = 0;        // conv.dbg: timing-only ablation of the DMA (tools/conv_ablate.py)
    if (a.dbg & 2) smask_ = 0;
#endif
    const int wmask = __builtin_amdgcn_readfirstlane(wmask_), smask = __builtin_amdgcn_readfirstlane(smask_);
    const int early_i = __builtin_amdgcn_readfirstlane(late ? 0 : 1);
    auto w_piece = [&](auto I, int slot_, int kbase) -> int {   // piece i of the chunk at kbase -> ring slot slot_; 1 if this wave has it
        constexpr int i = decltype(I)::value;
        if (!(wmask & (1 << i))) return 0;
        char* dst = smem + 2 * SLABB + slot_ * WCH;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(dst + (i * NW + wave) * 1024), 16, wvoff[i], kbase, 0,
                                                 0);
        return 1;
    };
    auto w_chunk_all = [&](int slot_, int kbase) {
        (void)w_piece(std::integral_constant<int, 0>{}, slot_, kbase);
        if constexpr (WLN > 1) (void)w_piece(std::integral_constant<int, 1>{}, slot_, kbase);
        if constexpr (WLN > 2) (void)w_piece(std::integral_constant<int, 2>{}, slot_, kbase);
        if constexpr (WLN > 3) (void)w_piece(std::integral_constant<int, 3>{}, slot_, kbase);
        if constexpr (WLN > 4) (void)w_piece(std::integral_constant<int, 4>{}, slot_, kbase);
    };

    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][m][r] = 0.0f;

    // next-phase cursor of the slab prefetch
    int np_step = 0;
    int np_f = t - nk + 1 + slab_tap_of_step(0, t, nk, a.torder);
    uint32_t np_off = (uint32_t)max(np_f, 0) * frame_bytes;
    int np_sl = 0;
    auto advance_phase = [&]() {
        if (++np_sl == S) {
            np_sl = 0;
            ++np_step;
            np_f = t - nk + 1 + slab_tap_of_step(min(np_step, nk - 1), t, nk, a.torder);
            np_off = (uint32_t)max(np_f, 0) * frame_bytes;
        } else {
            np_off += (uint32_t)PITCH;
        }
    };
#pragma unroll
    for (int j = 0; j < NPJ; ++j)
        if (j * NW + wave < NSP) slab_piece(0, np_off, soff[j], j);
    advance_phase();
    w_chunk_all(0, wk_cur);                                   // chunks 0, 1, 2 of phase 0 (a phase has nine: they exist)
    w_chunk_all(1, wk_cur + a.Cin * 2);
    w_chunk_all(2, wk_cur + 2 * a.Cin * 2);

    // ---- fragment reads ----
    const int wrow = wn * (NT * 32) + l31;                     // this lane's first weight row inside the chunk
    bf16x8 af[2][KS][MT], wf[2][KS][NT];
    constexpr int NR = KS * (MT + NT), NM = KS * NT * MT;      // fragment reads / MFMAs per chunk
    const char* rbase[MT + NT];                                // LDS row addresses of the chunk being read
    int rkey[MT + NT];                                         // ... and their swizzle keys (position / weight row)
    int pa0[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) pa0[m] = (MT * wr + m) * SC + l31;
    // address arithmetic of the reads of the chunk with spatial tap SPR in slab buffer buf_ and weight-ring slot slot_
    auto read_setup = [&](auto SPR, int buf_, int slot_) {
        constexpr int spr = decltype(SPR)::value;
        constexpr int dy = spr / 3, dx = spr - dy * 3;
        const char* Sb = smem + buf_ * SLABB;
        const char* Ws = smem + 2 * SLABB + slot_ * WCH;
        // opaque copies: with the tap a compile-time constant everything derived from the lane's position is loop-invariant, and
        // hipcc hoists nine taps' worth of swizzled offsets out of the chunk loop (+60 VGPRs: spills inside the loop)
        int w0 = wrow;
        asm volatile("" : "+v"(w0));
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            int p0 = pa0[m];
            asm volatile("" : "+v"(p0));
            rkey[m] = p0 + (dy * SC + dx);
            rbase[m] = Sb + rkey[m] * PITCH;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            rkey[MT + nt] = w0 + nt * 32;
            rbase[MT + nt] = Ws + rkey[MT + nt] * PITCH;
        }
    };
    auto read_one = [&](auto SET, auto R) {                   // read r of the chunk: k-step r / (MT + NT), operand r % (MT + NT)
        constexpr int s_ = decltype(SET)::value, r = decltype(R)::value;
        constexpr int ks = r / (MT + NT), o = r % (MT + NT);
        const bf16x8 v = *(const bf16x8*)(rbase[o] + (slab_swz<SLW>(2 * ks + hi, rkey[o]) << 4));
        if constexpr (o < MT) af[s_][ks][o] = v;
        else wf[s_][ks][o - MT] = v;
    };
    auto read_range = [&](auto SET, auto LO, auto HI, auto&& self) {
        constexpr int lo = decltype(LO)::value, hi_ = decltype(HI)::value;
        if constexpr (lo < hi_) {
            read_one(SET, LO);
            self(SET, std::integral_constant<int, lo + 1>{}, HI, self);
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_setup(std::integral_constant<int, 0>{}, 0, 0);
    read_range(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, NR>{}, read_range);
    // (the builtin, not inline asm: the compiler's own waitcnt bookkeeping must SEE that no fragment read is pending at the
    // top of an interval — otherwise it guards the first MFMA with lgkmcnt(0), i.e. waits for the prefetch just issued)
    __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0)

    // One interval = the NM MFMAs of a chunk with everything else issued in their shadow: after MFMA i, the wave issues its share of
    // the NEXT chunk's fragment reads (into the other register set; past the last chunk they read stale LDS, unused) and, at slots
    // that differ between the two waves of a SIMD, one DMA piece.  Round 4: the interval is instantiated per (spatial tap, register
    // set) — eighteen chunks = two phases per trip of the loop — so which slot carries which piece, the next chunk's tap and the
    // issue cursor's tap are compile-time; what stays run-time is the ring slot, the slab buffer parity and the end-of-clip guards.
    // Before, that bookkeeping was ~226 scalar instructions and ~39 branches per interval beside 18 MFMAs: 14 issues per MFMA gap
    // where ~5 fit (MI355X_MICROARCH.md, per-instruction constants).
    constexpr int DSTR = NM >= 16 ? 2 : 1;                     // MFMA slots between two DMA pieces
    // per-interval cycle stamps (tools/conv_prof.py): debug builds only — in the shipped kernel they were six uniform branches and a
    // dozen live SGPRs per interval
#ifdef APEXMI_DEBUG
    const bool prof = a.prof != nullptr && blockIdx.x == 300 && blockIdx.y == 0;
    unsigned long long pf_issue = 0, pf_vm = 0, pf_lgkm = 0, pf_bar = 0, pf_t3 = 0;
#define STAMP(x) do { if (prof) { __builtin_amdgcn_sched_barrier(0); x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define STAMP(x) do { } while (0)
#endif
    int ph = 0;                                                // phase of the chunk whose MFMAs run
    // DMA piece d of the interval of spatial tap SP: weights of the chunk three ahead first, then the next slab's share of this tap
    auto dma_piece = [&](auto SP, auto D, int it, int& issued) {
        constexpr int sp = decltype(SP)::value, d = decltype(D)::value;
        if constexpr (d < WLN) {
            if (!(ph + 1 == nph && sp >= 6)) {                 // chunk it + 3 exists
                const int kb = (sp < 6 ? wk_cur : wk_next) + ((sp + 3) % 9) * (a.Cin * 2);
                issued += w_piece(std::integral_constant<int, d>{}, (it + 3) & 3, kb);
            }
        } else if constexpr (sp < 7) {
            if (ph + 1 < nph) {
#pragma unroll
                for (int j = 0; j < NPJ; ++j)
                    if (j % SPT == d - WLN && j * 7 / NPJ == sp) {
                        if (smask & (1 << j)) {
                            slab_piece((ph + 1) & 1, np_off, soff[j], j);
                            ++issued;
                        }
                    }
            }
        }
    };
    auto dma_at = [&](auto SP, auto I, auto D, int it, int& issued, auto&& self) {   // pieces whose slot is MFMA i (early / late half)
        constexpr int i = decltype(I)::value, d = decltype(D)::value;
        if constexpr (d < WLN + SPT) {
            if constexpr (i == 1 + DSTR * d) {
                if (early_i) dma_piece(SP, D, it, issued);
            }
            if constexpr (NW == 8 && i == NM / 2 + DSTR * d) {
                if (!early_i) dma_piece(SP, D, it, issued);
            }
            self(SP, I, std::integral_constant<int, d + 1>{}, it, issued, self);
        }
    };
    auto slots = [&](auto SET, auto SP, auto I, int it, int& issued, auto&& self) {
        constexpr int s_ = decltype(SET)::value, sp = decltype(SP)::value, i = decltype(I)::value;
        if constexpr (i < NM) {
            constexpr int ks = i / (NT * MT), nt = (i / MT) % NT, m = i % MT;
            acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s_][ks][nt], af[s_][ks][m], acc[nt][m], 0, 0, 0);
            if constexpr (i == 0) {
                // address arithmetic of chunk it + 1 under the first MFMA: its tap is compile-time, its slab buffer is the next
                // phase's after tap 8
                read_setup(std::integral_constant<int, (sp + 1) % 9>{}, (sp == 8 ? ph + 1 : ph) & 1, (it + 1) & 3);
            } else {
                read_range(std::integral_constant<int, s_ ^ 1>{}, std::integral_constant<int, (i - 1) * NR / (NM - 1)>{},
                           std::integral_constant<int, i * NR / (NM - 1)>{}, read_range);
            }
            dma_at(SP, I, std::integral_constant<int, 0>{}, it, issued, dma_at);
            __builtin_amdgcn_sched_barrier(0);
            self(SET, SP, std::integral_constant<int, i + 1>{}, it, issued, self);
        }
    };
    auto interval = [&](auto IDX, int it) {
        constexpr int idx = decltype(IDX)::value;
        constexpr int sp = idx % 9;
#ifdef APEXMI_DEBUG
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#endif
        STAMP(t0);
        __builtin_amdgcn_sched_barrier(0);
        int issued = 0;
        slots(std::integral_constant<int, idx & 1>{}, std::integral_constant<int, sp>{}, std::integral_constant<int, 0>{}, it, issued, slots);
        STAMP(t1);
        switch (issued) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        }
        STAMP(t2);
        __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): the prefetched set (long complete by now)
#ifdef APEXMI_DEBUG
        if (prof) {
            if (it > 0) pf_bar += t0 - pf_t3;
            pf_issue += t1 - t0;
            pf_vm += t2 - t1;
        }
        STAMP(t3);
        if (prof) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            pf_lgkm += t3 - t2;
            pf_t3 = t3;
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (sp == 8) {
            ++ph;
            advance_phase();
            wk_cur = wk_next;
            w_phase_advance();
            wk_next = w_phase_base();
        }
    };
    auto phase9 = [&](auto BASE, int it) {                     // the nine intervals of one phase; BASE = 0 / 9: register-set parity
        constexpr int b0 = decltype(BASE)::value;
        interval(std::integral_constant<int, b0 + 0>{}, it + 0);
        interval(std::integral_constant<int, b0 + 1>{}, it + 1);
        interval(std::integral_constant<int, b0 + 2>{}, it + 2);
        interval(std::integral_constant<int, b0 + 3>{}, it + 3);
        interval(std::integral_constant<int, b0 + 4>{}, it + 4);
        interval(std::integral_constant<int, b0 + 5>{}, it + 5);
        interval(std::integral_constant<int, b0 + 6>{}, it + 6);
        interval(std::integral_constant<int, b0 + 7>{}, it + 7);
        interval(std::integral_constant<int, b0 + 8>{}, it + 8);
    };
    CONV_TRACE_AT(tr_l0);
    for (int it = 0; it < nchunks; it += 18) {
        phase9(std::integral_constant<int, 0>{}, it);
        if (it + 9 < nchunks) phase9(std::integral_constant<int, 9>{}, it + 9);
    }
    CONV_TRACE_AT(tr_l1);

#undef STAMP
#ifdef APEXMI_DEBUG
    if (prof && lane == 0) {
        unsigned long long* o = a.prof + wave * 8;
        o[0] = pf_bar;
        o[1] = pf_issue;
        o[2] = pf_vm;
        o[3] = pf_lgkm;
        o[5] = (unsigned long long)nchunks;
    }
#endif
    int mrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int y = y0 + MT * wr + m, x = x0 + l31;
        mrow[m] = (y < a.H && x < a.W) ? (t * a.H + y) * a.W + x : -1;
    }
    conv_epilogue<MT, NT, WNW, NORM>(a, acc, mrow, n0, wr, wn, l31, hi, smem);
    CONV_TRACE_END();
}

using CV_N32 = ConvCfg<8, 1, 2, 1>;    // 512 x 32
using CV_N64 = ConvCfg<8, 1, 2, 2>;    // 512 x 64
using CV_N96 = ConvCfg<8, 1, 2, 3>;    // 512 x 96
using CV_N128 = ConvCfg<4, 2, 2, 2>;   // 256 x 128
using CV_N192 = ConvCfg<4, 2, 2, 3>;   // 256 x 192
using CV_N256 = ConvCfg<2, 4, 4, 2>;   // 256 x 256

int g_conv_v2 = 1;   // apexmi_tune_set("conv.v2", 0/1)
int g_conv_slab = 2; // apexmi_tune_set("conv.slab", 0 | 1 | 2): direct convolution — 2 (shipped) the sliced kernel for every
                     // eligible layer, 1 the order-preserving 8 x 32 form for Cin = 96 and the sliced kernel elsewhere, 0 off

template <int NT, int NORM>
int launch_slab96_inst(const ConvArgs& a, hipStream_t stream) {     // conv.slab = 1: the order-preserving 8 x 32 form, Cin = 96
    static uint64_t attr = 0;
    APEXMI_SET_ATTR_ONCE(attr,
        (void)hipFuncSetAttribute((const void*)conv3d_slab96_kernel<NT, NORM>, hipFuncAttributeMaxDynamicSharedMemorySize, SLAB_LDS));
    const int grid = a.T * ((a.H + 7) / 8) * ((a.W + 31) / 32);
    hipLaunchKernelGGL((conv3d_slab96_kernel<NT, NORM>), dim3(grid), dim3(512), SLAB_LDS, stream, a);
    return apexmi_check_launch("conv3d_cl (slab 8x32)");
}

int g_conv_dbg = 0;
int g_conv_torder = 1;   // apexmi_tune_set("conv.torder", 0/1): see slab_tap_of_step (1 shipped)
uintptr_t g_conv_prof = 0;   // apexmi_tune_set("conv.prof_lo" / "conv.prof_hi", halves of a device pointer): 8 waves x 8 counters
int g_conv_pp = 1;   // apexmi_tune_set("conv.pp", 0..3): see launch_slab

template <int NT, int MT, int NORM, int SLW = 48>
int launch_slab_inst(const ConvArgs& a, hipStream_t stream) {
    constexpr int SPOS = (8 * MT + 2) * 34, NPJ = (SPOS * SLW * 2 + 8191) / 8192;
    constexpr int LDS = 2 * NPJ * 8192 + 3 * NT * 32 * SLW * 2;
    static_assert(LDS <= 160 * 1024, "slab kernel LDS");
    static uint64_t attr = 0;
    APEXMI_SET_ATTR_ONCE(attr,
        (void)hipFuncSetAttribute((const void*)conv3d_slab_kernel<NT, MT, NORM, SLW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    const int gx = a.T * ((a.H + 8 * MT - 1) / (8 * MT)) * ((a.W + 31) / 32), gy = (a.Cout + NT * 32 - 1) / (NT * 32);
#if APEXMI_CONV_TRACE
    conv_trace_arm(stream);
#endif
    hipLaunchKernelGGL((conv3d_slab_kernel<NT, MT, NORM, SLW>), dim3(gx, gy), dim3(512), LDS, stream, a);
    return apexmi_check_launch("conv3d_cl (slab)");
}

template <int NT, int MT, int WNW, int NORM, int SLW = 48, int NW = 8>
int launch_slabp_inst(const ConvArgs& a, hipStream_t stream) {
    constexpr int TH = (NW / WNW) * MT, SPOS = (TH + 2) * 34, NSP = (SPOS * SLW * 2 + 1023) / 1024;
    constexpr int WROWS = WNW * NT * 32;
    constexpr int LDS = 2 * NSP * 1024 + 4 * WROWS * SLW * 2;
    static_assert(LDS <= 160 * 1024, "slab kernel LDS");
    static uint64_t attr = 0;
    APEXMI_SET_ATTR_ONCE(attr, (void)hipFuncSetAttribute((const void*)conv3d_slabp_kernel<NT, MT, WNW, NORM, SLW, NW>,
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    const int gx = a.T * ((a.H + TH - 1) / TH) * ((a.W + 31) / 32), gy = (a.Cout + WROWS - 1) / WROWS;
#if APEXMI_CONV_TRACE
    conv_trace_arm(stream);
#endif
    hipLaunchKernelGGL((conv3d_slabp_kernel<NT, MT, WNW, NORM, SLW, NW>), dim3(gx, gy), dim3(NW * 64), LDS, stream, a);
    return apexmi_check_launch("conv3d_cl (slab, prefetch)");
}

// which convolutions the slab kernels take: Cin a multiple of 48, 3x3 "same" spatial taps, kT <= 3, stride 1, zero padding
// 48-channel slices (16x32 / 8x32 tiles) where Cin is a multiple of 48; otherwise 64-channel slices on 8 x 32 tiles with 128
// output channels per workgroup (Cout a multiple of 128: the HunyuanVideo-1.5 / Flux / TAEHV stages), replicate padding included
bool slab48(const ConvArgs& a) { return a.Cin % 48 == 0 && !a.replicate && (a.Cout <= 192 || a.Cout % 192 == 0); }
bool slab64(const ConvArgs& a) { return a.Cin % 64 == 0 && a.Cout % 128 == 0 && !a.up && a.H < 32768 && a.W < 32768; }
bool slab_eligible(const ConvArgs& a) {
    return g_conv_slab && a.kH == 3 && a.kW == 3 && a.py == 1 && a.px == 1 && a.kT <= 3 && (slab48(a) || slab64(a));
}

int launch_slab(const ConvArgs& a, hipStream_t stream) {
    const bool norm = a.out_norm != nullptr;
    // conv.pp: 1 (shipped) = per shape, whichever schedule measured faster (profiles/r03_vae_conv_schedules.md): the register-
    // prefetch kernel on the 192-channel-wide tiles (+4..10 %), the serial-chunk kernel elsewhere; 0 / 2 = one of them
    // everywhere it exists; 3 = the prefetch kernel as 4 waves of twice the tile (A/B only, no fused norm)
    if (!slab48(a)) {     // 64-channel slices
        if (norm && a.Cout > 128) {
            apexmi_set_error("conv3d_cl_norm: Cout=%d does not fit one N tile of the slab kernel", a.Cout);
            return 1;
        }
        if (g_conv_pp == 2) return norm ? launch_slabp_inst<2, 2, 2, 1, 64>(a, stream) : launch_slabp_inst<2, 2, 2, 0, 64>(a, stream);
        return norm ? launch_slab_inst<4, 1, 1, 64>(a, stream) : launch_slab_inst<4, 1, 0, 64>(a, stream);
    }
    if (g_conv_slab == 1 && a.Cin == 96 && a.Cout <= 96 && !a.up) {
        const int nt = (a.Cout + 31) / 32;
        if (norm) return nt == 1 ? launch_slab96_inst<1, 1>(a, stream) : nt == 2 ? launch_slab96_inst<2, 1>(a, stream) : launch_slab96_inst<3, 1>(a, stream);
        return nt == 1 ? launch_slab96_inst<1, 0>(a, stream) : nt == 2 ? launch_slab96_inst<2, 0>(a, stream) : launch_slab96_inst<3, 0>(a, stream);
    }
    if (a.Cout <= 96) {
        const int nt = (a.Cout + 31) / 32;
        if (g_conv_pp == 3 && nt == 3 && !norm) return launch_slabp_inst<3, 4, 1, 0, 48, 4>(a, stream);
        if (g_conv_pp >= 2) {
            if (norm) return nt == 1 ? launch_slabp_inst<1, 2, 1, 1>(a, stream) : nt == 2 ? launch_slabp_inst<2, 2, 1, 1>(a, stream) : launch_slabp_inst<3, 2, 1, 1>(a, stream);
            return nt == 1 ? launch_slabp_inst<1, 2, 1, 0>(a, stream) : nt == 2 ? launch_slabp_inst<2, 2, 1, 0>(a, stream) : launch_slabp_inst<3, 2, 1, 0>(a, stream);
        }
        if (norm) return nt == 1 ? launch_slab_inst<1, 2, 1>(a, stream) : nt == 2 ? launch_slab_inst<2, 2, 1>(a, stream) : launch_slab_inst<3, 2, 1>(a, stream);
        return nt == 1 ? launch_slab_inst<1, 2, 0>(a, stream) : nt == 2 ? launch_slab_inst<2, 2, 0>(a, stream) : launch_slab_inst<3, 2, 0>(a, stream);
    }
    if (norm) {
        if (a.Cout > 192) {
            apexmi_set_error("conv3d_cl_norm: Cout=%d does not fit one N tile of the slab kernel", a.Cout);
            return 1;
        }
        return g_conv_pp ? launch_slabp_inst<3, 2, 2, 1>(a, stream) : launch_slab_inst<6, 1, 1>(a, stream);
    }
    if (g_conv_pp == 3) return launch_slabp_inst<3, 4, 2, 0, 48, 4>(a, stream);
    return g_conv_pp ? launch_slabp_inst<3, 2, 2, 0>(a, stream) : launch_slab_inst<6, 1, 0>(a, stream);
}

template <typename CFG, int UP, int NORM>
int launch_v2_inst(const ConvArgs& a, hipStream_t stream, int grid) {
    static uint64_t attr = 0;
    APEXMI_SET_ATTR_ONCE(attr,
        (void)hipFuncSetAttribute((const void*)conv3d_v2_kernel<CFG, UP, NORM>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CFG::STAGE));
    hipLaunchKernelGGL((conv3d_v2_kernel<CFG, UP, NORM>), dim3(grid), dim3(CFG::NTHR), 2 * CFG::STAGE, stream, a);
    return apexmi_check_launch("conv3d_cl (v2)");
}

template <typename CFG, bool CAN_NORM = false>
int launch_v2(const ConvArgs& a, hipStream_t stream) {
    const int64_t M = (int64_t)a.T * a.H * a.W;
    const int nm = (int)((M + CFG::BM - 1) / CFG::BM), nn = (a.Cout + CFG::BN - 1) / CFG::BN;
    if (a.out_norm != nullptr) {
        if constexpr (CAN_NORM) {
            if (nn == 1) return a.up ? launch_v2_inst<CFG, 1, 1>(a, stream, nm) : launch_v2_inst<CFG, 0, 1>(a, stream, nm);
        }
        apexmi_set_error("conv3d_cl_norm: Cout=%d does not fit one N tile of this tiling (see apexmi_conv3d_cl_norm_fusable)", a.Cout);
        return 1;
    }
    return a.up ? launch_v2_inst<CFG, 1, 0>(a, stream, nm * nn) : launch_v2_inst<CFG, 0, 0>(a, stream, nm * nn);
}

// tile choice: the N extent that wastes the fewest matrix columns; v2 only where it fills the chip
int launch_v2_for(const ConvArgs& a, hipStream_t stream, bool* taken) {
    *taken = false;
    const int64_t M = (int64_t)a.T * a.H * a.W;
    if (!g_conv_v2 || (a.replicate && (a.up || a.H > 65535 || a.W > 65535)) || a.sy != 1 || a.sx != 1 || a.st != 1 || a.t0 != 0 || a.To != a.T || a.Ho != a.H || a.Wo != a.W || a.ntaps > 27 ||
        (int64_t)a.T * a.Hin * a.Win * a.Cin * 2 >= ((int64_t)1 << 31))   // 32-bit byte offsets in the gather
        return 0;
    const int c = a.Cout;
    *taken = true;
    if (slab_eligible(a) && !(a.out_norm != nullptr && c > (slab48(a) ? 192 : 128))) {
        // the slab kernels need a round of workgroups and mostly full tiles, not 65536 positions: e.g. the 32 x 32 x 61-frame
        // stages of the HunyuanVideo-1.5 decoder (62464 positions, 1024 channels = 1952 workgroups)
        const bool two_rows = slab48(a) && c <= 96;
        const int th = two_rows ? 16 : 8;
        const int nty = (a.H + th - 1) / th, ntx = (a.W + 31) / 32;
        const int64_t wgs = (int64_t)a.T * nty * ntx * (two_rows ? 1 : (c + (slab48(a) ? 191 : 127)) / (slab48(a) ? 192 : 128));
        const double fill = (double)a.H * a.W / ((double)nty * th * ntx * 32);
        if (M >= 65536 || (wgs >= 256 && fill >= 0.7)) return launch_slab(a, stream);
    }
    if (M < 65536) {
        *taken = false;
        return 0;
    }
    if (c <= 32) return launch_v2<CV_N32, true>(a, stream);
    if (c <= 64) return launch_v2<CV_N64, true>(a, stream);
    if (c <= 96) return launch_v2<CV_N96, true>(a, stream);
    if (c <= 128) {           // 256 x 128 (64 x 64 wave tiles) measured 9 % slower than the 128 x 128 kernel: stay on it
        *taken = false;
        return 0;
    }
    if (c % 192 == 0 || (c > 128 && c <= 192)) return launch_v2<CV_N192, true>(a, stream);
    if (c % 256 == 0) return launch_v2<CV_N256>(a, stream);
    if (c % 128 == 0) return launch_v2<CV_N128>(a, stream);
    return launch_v2<CV_N192>(a, stream);
}

// ---- channels-last elementwise companions --------------------------------------------------------

// y = silu?( x / max(||x||_2, 1e-12) * sqrt(C) * gamma[c] ) per position  (WanRMS_norm.forward, reference
// vae/wan/model.py:216-222, followed by the SiLU of WanResidualBlock.forward :398-399).  G lanes per
// position, 8 channels per lane.
template <typename T, int G>
__global__ __launch_bounds__(256) void rmsnorm_cl_kernel(const T* __restrict__ x,
                                                         T* __restrict__ y,
                                                         const bf16_t* __restrict__ gamma, int64_t P, int C,
                                                         int silu) {
    const int64_t pos = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
    const int l = threadIdx.x % G;
    const bool live = pos < P && l * 8 < C;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.0f;
    if (live) load8<T>(x + pos * C + l * 8, v);
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sq += v[j] * v[j];
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float scale = sqrtf((float)C) / fmaxf(sqrtf(sq), 1e-12f);
    if (live) {
        float g[8];
        unpack8(*(const u32x4*)(gamma + l * 8), g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float r = v[j] * scale * g[j];
            if (silu) r = silu_f(r);
            v[j] = r;
        }
        store8<T>(y + pos * C + l * 8, v);
    }
}

// C in (512, 1024] (HunyuanVideo-1.5 VAE, 1024 channels): one wave per position, two 8-channel chunks per lane.
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_cl_wide_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                              const bf16_t* __restrict__ gamma, int64_t P, int C,
                                                              int silu) {
    const int64_t pos = (int64_t)blockIdx.x * 4 + threadIdx.x / 64;
    const int l = threadIdx.x % 64;
    float v[2][8];
    bool live[2];
    float sq = 0.0f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        live[h] = pos < P && (h * 64 + l) * 8 < C;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[h][j] = 0.0f;
        if (live[h]) load8<T>(x + pos * C + (h * 64 + l) * 8, v[h]);
#pragma unroll
        for (int j = 0; j < 8; ++j) sq += v[h][j] * v[h][j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float scale = sqrtf((float)C) / fmaxf(sqrtf(sq), 1e-12f);
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (live[h]) {
            float g[8];
            unpack8(*(const u32x4*)(gamma + (h * 64 + l) * 8), g);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float r = v[h][j] * scale * g[j];
                if (silu) r = silu_f(r);
                v[h][j] = r;
            }
            store8<T>(y + pos * C + (h * 64 + l) * 8, v[h]);
        }
}

// nearest(-exact) 2x spatial upsample, [T, H, W, C] -> [T, 2H, 2W, C]  (WanUpsample, model.py:225-237)
template <typename TS>
__global__ __launch_bounds__(256) void upsample2x_cl_kernel(const TS* __restrict__ x,
                                                            TS* __restrict__ y, int T, int H, int W,
                                                            int C) {
    const int cc = C >> 3;
    const int64_t n = (int64_t)T * 2 * H * 2 * W * cc;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int c = (int)(idx % cc);
    int64_t r = idx / cc;
    const int xo = (int)(r % (2 * W));
    r /= 2 * W;
    const int yo = (int)(r % (2 * H));
    const int t = (int)(r / (2 * H));
    straw8(y + idx * 8, ldraw8(x + (((int64_t)t * H + (yo >> 1)) * W + (xo >> 1)) * C + c * 8));
}

// time_conv output [T, H, W, 2C] -> frames interleaved [2T, H, W, C]: channel half h of frame t becomes
// frame 2t + h  (WanResample.forward, model.py:332-336)
template <typename TS>
__global__ __launch_bounds__(256) void time_interleave_cl_kernel(const TS* __restrict__ x,
                                                                 TS* __restrict__ y, int T, int64_t HW,
                                                                 int C) {
    const int cc = C >> 3;
    const int64_t n = (int64_t)2 * T * HW * cc;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int c = (int)(idx % cc);
    int64_t r = idx / cc;
    const int64_t p = r % HW;
    const int f = (int)(r / HW);
    straw8(y + idx * 8, ldraw8(x + (((int64_t)(f >> 1) * HW + p) * 2 + (f & 1)) * C + c * 8));
}

// TAEHV input clamp (tae/model.py:24-26) behind the light VAE's 1/scaling_factor (hunyuanvideo15/model.py:1225-1226):
// y = 3 tanh(x * inv / 3), f32 inside, one bf16 rounding
__global__ __launch_bounds__(256) void tanh_clamp_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t n8,
                                                         float inv) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n8) return;
    float v[8];
    unpack8(*(const u32x4*)(x + idx * 8), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 3.0f * tanhf(v[j] * inv * (1.0f / 3.0f));
    *(u32x4*)(y + idx * 8) = pack8(v);
}

// f32-storage verification mode: the same function without the bf16 rounding
__global__ __launch_bounds__(256) void tanh_clamp_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float inv) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < n) y[idx] = 3.0f * tanhf(x[idx] * inv * (1.0f / 3.0f));
}

// TAEHV output (tae/model.py:318-333): clamp to [lo, hi], pixel-shuffle by r (channel c r^2 + i r + j -> pixel
// (h r + i, w r + j) of image channel c) and drop the first t0 frames: x [T, H, W, Cs] -> y [C, T - t0, H r, W r]
template <int R, typename TS = bf16_t>
__global__ __launch_bounds__(256) void pixel_shuffle_clamp_kernel(const TS* __restrict__ x, TS* __restrict__ y, int T,
                                                                  int H, int W, int Cs, int C, int t0, float lo, float hi) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = (int64_t)C * (T - t0) * H * W;
    if (idx >= n) return;
    const int w = (int)(idx % W);
    int64_t r = idx / W;
    const int h = (int)(r % H);
    r /= H;
    const int t = (int)(r % (T - t0));
    const int c = (int)(r / (T - t0));
    const TS* src = x + ((((int64_t)(t + t0) * H + h) * W + w) * Cs + c * (R * R));
    TS* dst = y + (((int64_t)c * (T - t0) + t) * (H * R) + (int64_t)h * R) * ((int64_t)W * R) + (int64_t)w * R;
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const float v = fminf(fmaxf(load1<TS>(src + i * R + j), lo), hi);
            store1<TS>(dst + (int64_t)i * W * R + j, v);
        }
}

// b[o, e, i] = a[o, e, i] * (1 - e/E) + b[o, e, i] * (e/E)   (blend_v / blend_h, model.py:1404-1422)
template <typename TS>
__global__ __launch_bounds__(256) void crossfade_kernel(const TS* __restrict__ a, TS* __restrict__ b,
                                                        int64_t outer, int E, int64_t inner, int64_t a_so,
                                                        int64_t a_se, int64_t b_so, int64_t b_se) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= outer * E * inner) return;
    const int64_t i = idx % inner;
    const int64_t r = idx / inner;
    const int e = (int)(r % E);
    const int64_t o = r / E;
    const float w = (float)e / (float)E;
    TS* bp = b + o * b_so + e * b_se + i;
    const float av = load1<TS>(a + o * a_so + e * a_se + i);
    store1<TS>(bp, av * (1.0f - w) + load1<TS>(bp) * w);
}

// ---- GroupNorm (channels-last) for the Flux 2-D VAE: diffusers ResnetBlock2D / Decoder use
// GroupNorm(32, C, eps=1e-6) (+ SiLU) (reference vae/auto/model.py:35-41 takes the
// Decoder from diffusers).  Three deterministic passes: per-block per-channel partial sums, a f64
// combine into per-group mean / rstd, and the apply (+ affine, + optional SiLU).
template <typename TS>
__global__ __launch_bounds__(256) void gn_partial_kernel(const TS* __restrict__ x, float* __restrict__ part,
                                                         int64_t P, int C, int rows_per_block) {
    __shared__ float red[16][256];           // [statistic][thread]: lane-contiguous, no bank conflicts either way
    const int ncol = C >> 3;                 // 16-byte chunks per position (<= 64)
    const int col = threadIdx.x % ncol, r0 = threadIdx.x / ncol, rstep = 256 / ncol;
    const int64_t p0 = (int64_t)blockIdx.x * rows_per_block;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.0f;
    if (r0 < rstep) {
        for (int r = r0; r < rows_per_block; r += rstep) {
            const int64_t p = p0 + r;
            if (p >= P) break;
            float v[8];
            load8<TS>(x + p * C + col * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[j] += v[j];
                q[j] += v[j] * v[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[j][threadIdx.x] = s[j];
        red[8 + j][threadIdx.x] = q[j];
    }
    __syncthreads();
    if (threadIdx.x < ncol) {                // fixed-order sum over the threads of this column
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
        for (int t = threadIdx.x; t < rstep * ncol; t += ncol)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] += red[j][t];
        float* o = part + ((int64_t)blockIdx.x * C + threadIdx.x * 8) * 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o[2 * j] = acc[j];
            o[2 * j + 1] = acc[8 + j];
        }
    }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats,
                                                          int nblk, int C, int G, int64_t P, float eps) {
    // one workgroup per group; thread t sums items t, t + 256, ... of the group's nblk x cpg partials in f64, then a
    // fixed-shape tree in LDS: the result does not depend on scheduling (run-to-run bit-identical)
    __shared__ double rs[256], rq[256];
    const int g = blockIdx.x, t = threadIdx.x;
    const int cpg = C / G;
    const int64_t items = (int64_t)nblk * cpg;
    double s = 0.0, q = 0.0;
    for (int64_t i = t; i < items; i += 256) {
        const int64_t b = i / cpg;
        const int c = g * cpg + (int)(i % cpg);
        s += part[(b * C + c) * 2];
        q += part[(b * C + c) * 2 + 1];
    }
    rs[t] = s;
    rq[t] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) {
            rs[t] += rs[t + w];
            rq[t] += rq[t + w];
        }
        __syncthreads();
    }
    if (t == 0) {
        const double n = (double)P * cpg;
        const double mean = rs[0] / n;
        const double var = fmax(rq[0] / n - mean * mean, 0.0);
        stats[2 * g] = (float)mean;
        stats[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

template <typename TS>
__global__ __launch_bounds__(256) void gn_apply_kernel(const TS* __restrict__ x, TS* __restrict__ y,
                                                       const float* __restrict__ stats,
                                                       const bf16_t* __restrict__ gamma,
                                                       const bf16_t* __restrict__ beta, int64_t P, int C, int G,
                                                       int silu) {
    const int ncol = C >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= P * ncol) return;
    const int col = (int)(idx % ncol);
    const int cpg = C / G;
    float v[8], gm[8], bt[8];
    load8<TS>(x + idx * 8, v);
    unpack8(*(const u32x4*)(gamma + col * 8), gm);
    unpack8(*(const u32x4*)(beta + col * 8), bt);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = (col * 8 + j) / cpg;
        float r = (v[j] - stats[2 * g]) * stats[2 * g + 1] * gm[j] + bt[j];
        if (silu) r = silu_f(r);
        v[j] = r;
    }
    store8<TS>(y + idx * 8, v);
}

}  // namespace
void apexmi_set_conv_v2(int v) { g_conv_v2 = v; }
void apexmi_set_conv_slab(int v) { g_conv_slab = v; }
void apexmi_set_conv_pp(int v) { g_conv_pp = v; }
void apexmi_set_conv_dbg(int v) { g_conv_dbg = v; }
void apexmi_set_conv_torder(int v) { g_conv_torder = v; }
void apexmi_set_conv_prof(int half, int v) {
    if (half) g_conv_prof = (g_conv_prof & 0xffffffffull) | ((uintptr_t)(uint32_t)v << 32);
    else g_conv_prof = (g_conv_prof & ~(uintptr_t)0xffffffffull) | (uint32_t)v;
}

extern "C" size_t apexmi_groupnorm_workspace_bytes(int64_t P, int C) {
    const int nblk = (int)((P + 1023) / 1024);
    return ((size_t)nblk * C * 2 + 2 * 64) * sizeof(float);
}

template <typename TS>
static int groupnorm_cl_impl(const void* x, void* y, const void* gamma, const void* beta, int64_t P,
                             int C, int G, float eps, int silu, void* workspace, size_t workspace_bytes,
                             apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(x && y && gamma && beta && workspace && P > 0, "groupnorm_cl: bad arguments");
    APEXMI_REQUIRE(C % 8 == 0 && C <= 512 && G > 0 && G <= 64 && C % G == 0 && 256 % (C / 8) == 0,
                   "groupnorm_cl: C=%d G=%d unsupported", C, G);
    APEXMI_REQUIRE(workspace_bytes >= apexmi_groupnorm_workspace_bytes(P, C), "groupnorm_cl: workspace too small");
    const int rows = 1024;
    const int nblk = (int)((P + rows - 1) / rows);
    float* part = (float*)workspace;
    float* stats = part + (size_t)nblk * C * 2;
    ApexmiProfScope prof(3, stream, 0.0, 6.0 * (double)P * C);
    hipLaunchKernelGGL(gn_partial_kernel<TS>, dim3(nblk), dim3(256), 0, stream, (const TS*)x, part, P, C, rows);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(G), dim3(256), 0, stream, part, stats, nblk, C, G, P, eps);
    const int64_t n = P * (C / 8);
    hipLaunchKernelGGL(gn_apply_kernel<TS>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const TS*)x,
                       (TS*)y, stats, (const bf16_t*)gamma, (const bf16_t*)beta, P, C, G, silu);
    return apexmi_check_launch("groupnorm_cl");
}

extern "C" int apexmi_groupnorm_cl(const void* x, void* y, const void* gamma, const void* beta, int64_t P,
                                   int C, int G, float eps, int silu, void* workspace, size_t workspace_bytes,
                                   apexmi_stream_t stream_) {
    return groupnorm_cl_impl<bf16_t>(x, y, gamma, beta, P, C, G, eps, silu, workspace, workspace_bytes, stream_);
}
// f32-storage verification mode: x, y float; gamma / beta stay bf16 weights
extern "C" int apexmi_groupnorm_cl_f32(const void* x, void* y, const void* gamma, const void* beta, int64_t P,
                                       int C, int G, float eps, int silu, void* workspace, size_t workspace_bytes,
                                       apexmi_stream_t stream_) {
    return groupnorm_cl_impl<float>(x, y, gamma, beta, P, C, G, eps, silu, workspace, workspace_bytes, stream_);
}

static int conv3d_cl_impl(const void* in, const void* w, const void* bias, const void* residual, void* out,
                          const void* zeros, int T, int H, int W, int Cin, int Cout, int Kpad, int kT, int kH, int kW,
                          int replicate, apexmi_stream_t stream_, int sy = 1, int sx = 1, int py = -1, int px = -1,
                          int Ho = 0, int Wo = 0, int independent = 0, int up = 0, const void* norm_gamma = nullptr,
                          void* out_norm = nullptr, int norm_silu = 0, int act = 0, float act_slope = 0.0f, int st = 1,
                          int t0 = 0, int To = 0, int f32io = 0, int clip_frames = 0) {
    const int Hin = H, Win = W;
    if (up) {          // H, W arrive as the STORED extents; the convolution runs over the 2x upsampled image
        H *= 2;
        W *= 2;
    }
    if (py < 0) py = (kH - 1) / 2;   // "same" convolution
    if (px < 0) px = (kW - 1) / 2;
    if (Ho <= 0) Ho = H;
    if (Wo <= 0) Wo = W;
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(in && w && (out || out_norm) && zeros, "conv3d_cl: null operand");
    APEXMI_REQUIRE(T > 0 && H > 0 && W > 0, "conv3d_cl: empty volume");
    APEXMI_REQUIRE(Cin % 8 == 0 && Cout % 4 == 0, "conv3d_cl: Cin=%d must be a multiple of 8, Cout=%d of 4", Cin, Cout);
    const int ntaps = kT * kH * kW;
    APEXMI_REQUIRE(ntaps >= 1 && ntaps <= MAX_TAPS && (kH & 1) && (kW & 1), "conv3d_cl: kernel %dx%dx%d unsupported", kT, kH, kW);
    APEXMI_REQUIRE(sy >= 1 && sx >= 1 && py >= 0 && px >= 0 && py < kH && px < kW && (Ho - 1) * sy - py < H &&
                       (Wo - 1) * sx - px < W,
                   "conv3d_cl: stride %dx%d / pad %d,%d / output %dx%d do not fit the %dx%d input", sy, sx, py, px, Ho, Wo, H, W);
    APEXMI_REQUIRE(Kpad % BK == 0 && Kpad >= ntaps * Cin, "conv3d_cl: Kpad=%d must be >= taps*Cin rounded up to 64", Kpad);
    APEXMI_REQUIRE((int64_t)T * H * W < (int64_t)2147483647 - BM, "conv3d_cl: volume too large for one call");
    APEXMI_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)out % 8) == 0 &&
                       ((uintptr_t)zeros % 16) == 0,
                   "conv3d_cl: operands must be 16-byte aligned");
    static uint64_t attr_set = 0;
    APEXMI_SET_ATTR_ONCE(attr_set, {
        (void)hipFuncSetAttribute((const void*)conv3d_cl_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  2 * STAGE_BYTES);
        (void)hipFuncSetAttribute((const void*)conv3d_cl_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  2 * STAGE_BYTES);
    });
    ConvArgs a;
    a.in = (const bf16_t*)in;
    a.w = (const bf16_t*)w;
    a.bias = (const bf16_t*)bias;
    a.res = (const bf16_t*)residual;
    a.out = (bf16_t*)out;
    a.zeros = (const bf16_t*)zeros;
    // A single frame sees only the LAST temporal tap: the kT - 1 earlier ones fall in the causal zero padding
    // (QwenImage's image VAE and every T = 1 tile run 3x3x3 kernels on one frame).  Skip them: start at the last
    // temporal slice of the packed weight (k = tap * Cin + ci, taps time-major) and iterate kH*kW taps — a third of the
    // work, the same sum (the skipped products are exact zeros).
    int Kext = Kpad, kT_eff = kT, ntaps_eff = ntaps;
    const int skip = (kT - 1) * kH * kW * Cin, kspatial = ((kH * kW * Cin + BK - 1) / BK) * BK;
    APEXMI_REQUIRE(!independent || kT == 1 || (!replicate && skip % 8 == 0 && skip + kspatial <= Kpad),
                   "conv3d_cl_frames: Kpad=%d leaves no room to address the last temporal slice (need >= %d; pack the "
                   "weight with apexmi's packing rule)", Kpad, skip + kspatial);
    if ((T == 1 || independent) && kT > 1 && !replicate && skip % 8 == 0 && skip + kspatial <= Kpad) {
        a.w += skip;
        Kext = kspatial;
        kT_eff = 1;
        ntaps_eff = kH * kW;
    }
    a.T = T; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Kpad = Kpad; a.Kext = Kext;
    a.kT = kT_eff; a.kH = kH; a.kW = kW; a.ntaps = ntaps_eff;
    a.q64 = 64 / Cin;
    a.r64 = 64 % Cin;
    a.replicate = replicate;
    a.Ho = Ho; a.Wo = Wo; a.sy = sy; a.sx = sx; a.py = py; a.px = px;
    a.up = up ? 1 : 0; a.Hin = Hin; a.Win = Win;
    a.norm_gamma = (const bf16_t*)norm_gamma; a.out_norm = (bf16_t*)out_norm; a.norm_silu = norm_silu;
    APEXMI_REQUIRE(act == 0 || (act == 1 && out_norm == nullptr), "conv3d_cl: activation %d unsupported (0 none, 1 leaky ReLU; "
                                                                  "not together with the fused norm)", act);
    a.act = act; a.act_slope = act_slope;
    if (To <= 0) To = T;
    APEXMI_REQUIRE(st >= 1 && t0 >= 0 && (To - 1) * st + t0 < T, "conv3d_cl: temporal stride %d / first frame %d / %d output frames "
                                                                 "do not fit %d input frames", st, t0, To, T);
    APEXMI_REQUIRE((st == 1 && t0 == 0 && To == T) || (!up && !independent && out_norm == nullptr && T > 1),
                   "conv3d_cl: a temporal stride excludes the upsample / independent-frame / fused-norm modes");
    a.To = To; a.st = st; a.t0 = t0;
    APEXMI_REQUIRE(clip_frames >= 0 && (clip_frames == 0 || (T % clip_frames == 0 && st == 1 && t0 == 0 && To == T && !up &&
                                                             !independent && out_norm == nullptr)),
                   "conv3d_cl: clip_frames=%d must divide T=%d (stride-1 plain convolutions only)", clip_frames, T);
    a.Tc = clip_frames == T ? 0 : clip_frames;
    a.prof = (unsigned long long*)g_conv_prof;
    a.dbg = g_conv_dbg;
    a.torder = g_conv_torder;
    const int64_t M = (int64_t)To * Ho * Wo;
    const int nm = (int)((M + BM - 1) / BM), nn = (Cout + BN - 1) / BN;
    ApexmiProfScope prof(0, stream, 2.0 * M * Cout * (double)ntaps_eff * Cin,
                         2.0 * ((double)M * Cin + (double)Cout * Kext + (double)M * Cout));
    if (f32io) {   // verification mode: float out / residual, the 128x128 implicit GEMM for every shape
        APEXMI_REQUIRE(out && out_norm == nullptr && ((uintptr_t)out % 16) == 0 && ((uintptr_t)residual % 16) == 0,
                       "conv3d_cl_f32: float out / residual must be 16-byte aligned (no fused norm)");
        hipLaunchKernelGGL(conv3d_cl_kernel<float>, dim3(nm * nn), dim3(256), 2 * STAGE_BYTES, stream, a);
        return apexmi_check_launch("conv3d_cl_f32");
    }
    bool taken = false;
    // stacked clips: the clip boundary lives in the 128x128 kernel's gather only (the conv-shaped / slab tiles walk frames)
    const int rc2 = a.Tc > 0 ? 0 : launch_v2_for(a, stream, &taken);
    if (taken) return rc2;
    APEXMI_REQUIRE(out_norm == nullptr, "conv3d_cl_norm: this convolution does not run on the fused-norm tiles "
                                        "(ask apexmi_conv3d_cl_norm_fusable first)");
    hipLaunchKernelGGL(conv3d_cl_kernel<bf16_t>, dim3(nm * nn), dim3(256), 2 * STAGE_BYTES, stream, a);
    return apexmi_check_launch("conv3d_cl");
}

extern "C" int apexmi_conv3d_cl(const void* in, const void* w, const void* bias, const void* residual,
                                void* out, const void* zeros, int T, int H, int W, int Cin, int Cout,
                                int Kpad, int kT, int kH, int kW, apexmi_stream_t stream_) {
    return conv3d_cl_impl(in, w, bias, residual, out, zeros, T, H, W, Cin, Cout, Kpad, kT, kH, kW, 0, stream_);
}

extern "C" int apexmi_conv3d_cl_clips(const void* in, const void* w, const void* bias, const void* residual, void* out,
                                      const void* zeros, int T, int H, int W, int Cin, int Cout, int Kpad, int kT, int kH,
                                      int kW, int replicate, int clip_frames, apexmi_stream_t stream_) {
    return conv3d_cl_impl(in, w, bias, residual, out, zeros, T, H, W, Cin, Cout, Kpad, kT, kH, kW, replicate, stream_, 1, 1, -1,
                          -1, 0, 0, 0, 0, nullptr, nullptr, 0, 0, 0.0f, 1, 0, 0, 0, clip_frames);
}

extern "C" int apexmi_conv3d_cl_up2(const void* in, const void* w, const void* bias, const void* residual, void* out,
                                    const void* zeros, int T, int H, int W, int Cin, int Cout, int Kpad, int kT, int kH,
                                    int kW, int independent, apexmi_stream_t stream_) {
    return conv3d_cl_impl(in, w, bias, residual, out, zeros, T, H, W, Cin, Cout, Kpad, kT, kH, kW, 0, stream_, 1, 1, -1, -1,
                          0, 0, independent, 1);
}

// can apexmi_conv3d_cl_norm run this shape?  (stride-1 zero-padded conv on the v2 tiles with every output channel of a
// position inside one workgroup tile: Cout <= 96 or 128 < Cout <= 192)
extern "C" int apexmi_conv3d_cl_norm_fusable(int T, int H, int W, int Cin, int Cout, int up) {
    const int64_t M = (int64_t)T * H * W * (up ? 4 : 1);
    return g_conv_v2 && M >= 65536 && (int64_t)T * H * W * Cin * 2 < ((int64_t)1 << 31) && Cout % 8 == 0 &&
           (Cout <= 96 || (Cout > 128 && Cout <= 192));
}

extern "C" int apexmi_conv3d_cl_norm(const void* in, const void* w, const void* bias, const void* residual, void* out,
                                     void* out_norm, const void* gamma, int silu, const void* zeros, int T, int H, int W,
                                     int Cin, int Cout, int Kpad, int kT, int kH, int kW, int independent, int up,
                                     apexmi_stream_t stream_) {
    APEXMI_REQUIRE(out_norm && gamma, "conv3d_cl_norm: out_norm and gamma are required");
    APEXMI_REQUIRE(Cout % 8 == 0, "conv3d_cl_norm: Cout=%d must be a multiple of 8", Cout);
    return conv3d_cl_impl(in, w, bias, residual, out, zeros, T, H, W, Cin, Cout, Kpad, kT, kH, kW, 0, stream_, 1, 1, -1, -1,
                          0, 0, independent, up, gamma, out_norm, silu);
}

extern "C" int apexmi_conv3d_cl_act(const void* in, const void* w, const void* bias, const void* residual, void* out,
                                    const void* zeros, int T, int H, int W, int Cin, int Cout, int Kpad, int kT, int kH,
                                    int kW, int independent, int up, int act, float slope, apexmi_stream_t stream_) {
    return conv3d_cl_impl(in, w, bias, residual, out, zeros, T, H, W, Cin, Cout, Kpad, kT, kH, kW, 0, stream_, 1, 1, -1, -1,
                          0, 0, independent, up, nullptr, nullptr, 0, act, slope);
}

extern "C" int apexmi_conv3d_cl_frames(const void* in, const void* w, const void* bias, const void* residual,
                                       void* out, const void* zeros, int N, int H, int W, int Cin, int Cout, int Kpad,
                                       int kT, int kH, int kW, apexmi_stream_t stream_) {
    return conv3d_cl_impl(in, w, bias, residual, out, zeros, N, H, W, Cin, Cout, Kpad, kT, kH, kW, 0, stream_, 1, 1, -1, -1,
                          0, 0, 1);
}

extern "C" int apexmi_conv3d_cl_strided(const void* in, const void* w, const void* bias, const void* residual,
                                        void* out, const void* zeros, int T, int H, int W, int Cin, int Cout, int Kpad,
                                        int kT, int kH, int kW, int stride_h, int stride_w, int pad_top, int pad_left,
                                        int Ho, int Wo, apexmi_stream_t stream_) {
    APEXMI_REQUIRE(Ho > 0 && Wo > 0, "conv3d_cl_strided: empty output %dx%d", Ho, Wo);
    return conv3d_cl_impl(in, w, bias, residual, out, zeros, T, H, W, Cin, Cout, Kpad, kT, kH, kW, 0, stream_, stride_h,
                          stride_w, pad_top, pad_left, Ho, Wo);
}

extern "C" int apexmi_conv3d_cl_tstrided(const void* in, const void* w, const void* bias, const void* residual, void* out,
                                         const void* zeros, int T, int H, int W, int Cin, int Cout, int Kpad, int kT, int kH,
                                         int kW, int stride_t, int t_first, int To, apexmi_stream_t stream_) {
    APEXMI_REQUIRE(To > 0, "conv3d_cl_tstrided: empty output");
    return conv3d_cl_impl(in, w, bias, residual, out, zeros, T, H, W, Cin, Cout, Kpad, kT, kH, kW, 0, stream_, 1, 1, -1, -1, 0, 0,
                          0, 0, nullptr, nullptr, 0, 0, 0.0f, stride_t, t_first, To);
}

// f32-storage verification mode, every variant of the convolution through one entry point: `in` = the three-way bf16
// split of the float activations ([T, H, W, Cin3], Cin3 = 3 x the layer's input channels), `w` packed with each tap's
// channel run repeated three times, bias bf16, residual / out FLOAT [.., Cout].  flags: 1 replicate padding, 2 independent
// frames, 4 read through a nearest 2x upsample.  stride / pad / output extents as apexmi_conv3d_cl_strided (pad < 0: "same"),
// temporal stride as apexmi_conv3d_cl_tstrided (To <= 0: every frame), act / slope as apexmi_conv3d_cl_act.
extern "C" int apexmi_conv3d_cl_f32(const void* in, const void* w, const void* bias, const void* residual, void* out,
                                    const void* zeros, int T, int H, int W, int Cin3, int Cout, int Kpad, int kT, int kH,
                                    int kW, int flags, int stride_h, int stride_w, int pad_top, int pad_left, int Ho, int Wo,
                                    int stride_t, int t_first, int To, int act, float slope, apexmi_stream_t stream_) {
    return conv3d_cl_impl(in, w, bias, residual, out, zeros, T, H, W, Cin3, Cout, Kpad, kT, kH, kW, flags & 1, stream_,
                          stride_h, stride_w, pad_top, pad_left, Ho, Wo, (flags >> 1) & 1, (flags >> 2) & 1, nullptr, nullptr, 0,
                          act, slope, stride_t, t_first, To, 1);
}

extern "C" int apexmi_conv3d_cl_replicate(const void* in, const void* w, const void* bias, const void* residual,
                                          void* out, const void* zeros, int T, int H, int W, int Cin, int Cout,
                                          int Kpad, int kT, int kH, int kW, apexmi_stream_t stream_) {
    return conv3d_cl_impl(in, w, bias, residual, out, zeros, T, H, W, Cin, Cout, Kpad, kT, kH, kW, 1, stream_);
}

template <typename T>
static int rmsnorm_cl_impl(const void* x, void* y, const void* gamma, int64_t P, int C, int silu,
                           apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(x && y && gamma && P > 0, "rmsnorm_cl: bad arguments");
    APEXMI_REQUIRE(C % 8 == 0 && C <= 1024, "rmsnorm_cl: C=%d must be a multiple of 8 and <= 1024", C);
    ApexmiProfScope prof(3, stream, 0.0, 4.0 * (double)P * C);
    const int lanes = C / 8;
    if (lanes > 64)
        hipLaunchKernelGGL(rmsnorm_cl_wide_kernel<T>, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, stream,
                           (const T*)x, (T*)y, (const bf16_t*)gamma, P, C, silu);
    else if (lanes <= 16)
        hipLaunchKernelGGL((rmsnorm_cl_kernel<T, 16>), dim3((unsigned)((P + 15) / 16)), dim3(256), 0, stream,
                           (const T*)x, (T*)y, (const bf16_t*)gamma, P, C, silu);
    else if (lanes <= 32)
        hipLaunchKernelGGL((rmsnorm_cl_kernel<T, 32>), dim3((unsigned)((P + 7) / 8)), dim3(256), 0, stream,
                           (const T*)x, (T*)y, (const bf16_t*)gamma, P, C, silu);
    else
        hipLaunchKernelGGL((rmsnorm_cl_kernel<T, 64>), dim3((unsigned)((P + 3) / 4)), dim3(256), 0, stream,
                           (const T*)x, (T*)y, (const bf16_t*)gamma, P, C, silu);
    return apexmi_check_launch("rmsnorm_cl");
}

extern "C" int apexmi_rmsnorm_cl(const void* x, void* y, const void* gamma, int64_t P, int C, int silu,
                                 apexmi_stream_t stream_) {
    return rmsnorm_cl_impl<bf16_t>(x, y, gamma, P, C, silu, stream_);
}
extern "C" int apexmi_rmsnorm_cl_f32(const void* x, void* y, const void* gamma, int64_t P, int C, int silu,
                                     apexmi_stream_t stream_) {
    return rmsnorm_cl_impl<float>(x, y, gamma, P, C, silu, stream_);
}

extern "C" int apexmi_upsample2x_cl(const void* x, void* y, int T, int H, int W, int C,
                                    apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(x && y && T > 0 && H > 0 && W > 0 && C % 8 == 0, "upsample2x_cl: bad arguments");
    const int64_t n = (int64_t)T * 2 * H * 2 * W * (C / 8);
    ApexmiProfScope prof(5, stream, 0.0, 2.5 * (double)n * 16);
    hipLaunchKernelGGL(upsample2x_cl_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       (const bf16_t*)x, (bf16_t*)y, T, H, W, C);
    return apexmi_check_launch("upsample2x_cl");
}

template <typename TS>
static int time_interleave_cl_impl(const void* x, void* y, int T, int64_t HW, int C,
                                   apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(x && y && T > 0 && HW > 0 && C % 8 == 0, "time_interleave_cl: bad arguments");
    const int64_t n = (int64_t)2 * T * HW * (C / 8);
    ApexmiProfScope prof(5, stream, 0.0, 2.0 * (double)n * 16);
    hipLaunchKernelGGL(time_interleave_cl_kernel<TS>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       (const TS*)x, (TS*)y, T, HW, C);
    return apexmi_check_launch("time_interleave_cl");
}

extern "C" int apexmi_time_interleave_cl(const void* x, void* y, int T, int64_t HW, int C,
                                         apexmi_stream_t stream_) {
    return time_interleave_cl_impl<bf16_t>(x, y, T, HW, C, stream_);
}
extern "C" int apexmi_time_interleave_cl_f32(const void* x, void* y, int T, int64_t HW, int C,
                                             apexmi_stream_t stream_) {
    return time_interleave_cl_impl<float>(x, y, T, HW, C, stream_);
}

extern "C" int apexmi_tanh_clamp(const void* x, void* y, int64_t n, float inv_scale, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(x && y && n > 0 && n % 8 == 0, "tanh_clamp: n=%lld must be a positive multiple of 8", (long long)n);
    ApexmiProfScope prof(5, stream, 0.0, 4.0 * (double)n);
    hipLaunchKernelGGL(tanh_clamp_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x,
                       (bf16_t*)y, n / 8, inv_scale);
    return apexmi_check_launch("tanh_clamp");
}

template <typename TS>
static int pixel_shuffle_clamp_impl(const void* x, void* y, int T, int H, int W, int Cs, int C, int r, int t0, float lo, float hi,
                                    apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(x && y && T > 0 && H > 0 && W > 0 && C > 0 && t0 >= 0 && t0 < T, "pixel_shuffle_clamp: bad arguments");
    APEXMI_REQUIRE((r == 1 || r == 2) && Cs >= C * r * r, "pixel_shuffle_clamp: patch %d / channel stride %d unsupported", r, Cs);
    const int64_t n = (int64_t)C * (T - t0) * H * W;
    ApexmiProfScope prof(5, stream, 0.0, 2.0 * sizeof(TS) * (double)n * r * r);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (r == 2)
        hipLaunchKernelGGL((pixel_shuffle_clamp_kernel<2, TS>), grid, dim3(256), 0, stream, (const TS*)x, (TS*)y, T, H, W, Cs,
                           C, t0, lo, hi);
    else
        hipLaunchKernelGGL((pixel_shuffle_clamp_kernel<1, TS>), grid, dim3(256), 0, stream, (const TS*)x, (TS*)y, T, H, W, Cs,
                           C, t0, lo, hi);
    return apexmi_check_launch("pixel_shuffle_clamp");
}

extern "C" int apexmi_pixel_shuffle_clamp(const void* x, void* y, int T, int H, int W, int Cs, int C, int r, int t0, float lo,
                                          float hi, apexmi_stream_t stream) {
    return pixel_shuffle_clamp_impl<bf16_t>(x, y, T, H, W, Cs, C, r, t0, lo, hi, stream);
}

// f32-storage verification mode
extern "C" int apexmi_pixel_shuffle_clamp_f32(const void* x, void* y, int T, int H, int W, int Cs, int C, int r, int t0, float lo,
                                              float hi, apexmi_stream_t stream) {
    return pixel_shuffle_clamp_impl<float>(x, y, T, H, W, Cs, C, r, t0, lo, hi, stream);
}

extern "C" int apexmi_tanh_clamp_f32(const void* x, void* y, int64_t n, float inv_scale, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(x && y && n > 0, "tanh_clamp_f32: n=%lld must be positive", (long long)n);
    ApexmiProfScope prof(5, stream, 0.0, 8.0 * (double)n);
    hipLaunchKernelGGL(tanh_clamp_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)x, (float*)y, n,
                       inv_scale);
    return apexmi_check_launch("tanh_clamp_f32");
}

template <typename TS>
static int crossfade_impl(const void* a, void* b, int64_t outer, int E, int64_t inner, int64_t a_so,
                          int64_t a_se, int64_t b_so, int64_t b_se, apexmi_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APEXMI_REQUIRE(a && b && outer > 0 && E > 0 && inner > 0, "crossfade: bad arguments");
    const int64_t n = outer * E * inner;
    ApexmiProfScope prof(5, stream, 0.0, 6.0 * (double)n);
    hipLaunchKernelGGL(crossfade_kernel<TS>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       (const TS*)a, (TS*)b, outer, E, inner, a_so, a_se, b_so, b_se);
    return apexmi_check_launch("crossfade");
}

extern "C" int apexmi_crossfade(const void* a, void* b, int64_t outer, int E, int64_t inner, int64_t a_so,
                                int64_t a_se, int64_t b_so, int64_t b_se, apexmi_stream_t stream_) {
    return crossfade_impl<bf16_t>(a, b, outer, E, inner, a_so, a_se, b_so, b_se, stream_);
}
extern "C" int apexmi_crossfade_f32(const void* a, void* b, int64_t outer, int E, int64_t inner, int64_t a_so,
                                    int64_t a_se, int64_t b_so, int64_t b_se, apexmi_stream_t stream_) {
    return crossfade_impl<float>(a, b, outer, E, inner, a_so, a_se, b_so, b_se, stream_);
}
