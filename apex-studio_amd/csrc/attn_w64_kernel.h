// The C++ shell of attn_fwd_d128_w64_kernel (attention.hip): address set-up, ONE asm statement (the generated loop, W64_BODY), the
// normalising epilogue.  Included once per loop body: the shipped one, and the timing-only ablations of tools/attn_w64_ablate.sh
// (-DAPEXMI_ATTN_W64_ABLATE, side library).  W64_NAME = kernel name, W64_BODY = the generated include.
// W64_FALLBACK (optional) = a second generated loop with the per-tile running maximum: W64_BODY is then the loop WITHOUT one (the
// generator's max=first: every row keeps the integer maximum tile 0 gave it, so a tile carries no row-max chain, no raise test and
// no rescale test).  An integer shift of the maximum scales every probability, row sum and numerator by the same power of two, so
// as long as nothing leaves the f32 range the result has the rounding points of the running-maximum loop (and is that loop's bit
// for bit wherever it would not have raised its maximum after tile 0; elsewhere the exponent's f32 argument s c - m is rounded at
// another magnitude: a bf16 ulp on a few elements, neither closer to nor further from the exact softmax).  The shell checks the
// range — every row sum <= 2^60, which bounds every exponent of the row by +60 while tile 0's own maximum keeps the sum >= 1/2 —
// and a workgroup in which any row fails (scores that tower > 41 nats above the first 64 keys' best, inf, NaN) runs the
// W64_FALLBACK loop from scratch.  The workgroups' vote goes through
// one LDS word behind the rings (the launch allocates 4 * ATT_STAGE + 16 bytes); d_attn_w64_fallbacks counts the re-runs.
__global__ __launch_bounds__(256, 1) void W64_NAME(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ Vt,
    bf16_t* __restrict__ O, int H, int Sq, int Sk, int Skp, int nqb, int total, int64_t o_sb,
    int64_t o_ss, int64_t o_sh, float scale_log2e) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int s = xcd_remap(blockIdx.x, total);
    const int hb = s / nqb, qb = s % nqb;
    const bf16_t* Qp = Q + (int64_t)hb * Sq * HD;
    const bf16_t* Kp = K + (int64_t)hb * Sk * HD;
    const bf16_t* Vp = Vt + (int64_t)hb * HD * Skp;
    const int nt = __builtin_amdgcn_readfirstlane((Sk + KV - 1) / KV);
    const int rem = __builtin_amdgcn_readfirstlane(Sk & (KV - 1));
    const int row_a = qb * 256 + wave * 64 + l31;
    const bf16_t* qa = Qp + (int64_t)min(row_a, Sq - 1) * HD + hi * 8;
    const bf16_t* qbp = Qp + (int64_t)min(row_a + 32, Sq - 1) * HD + hi * 8;
    // LDS-DMA: two raw buffer descriptors (base, stride 0, bytes, flags), one per-lane offset each; K rows past Sk read as zero (their
    // scores are masked), V^T columns past Sk are the zero padding of the buffer.  K piece i of a wave = image rows 16 i + 4 wave +
    // (lane >> 4) <- keys 16 i + perm32(4 wave + (lane >> 4)), chunk (lane & 15) ^ row; V^T piece i = rows (d) 32 i + 8 wave +
    // (lane >> 3), chunk (lane & 7) ^ ((row >> 1) & 7)
    const uint64_t kb = (uint64_t)Kp, vb = (uint64_t)Vp;
    u32x4 rk = {(uint32_t)kb, (uint32_t)(kb >> 32) & 0xffffu, (uint32_t)(Sk * (HD * 2)), 0x00020000u};
    u32x4 rv = {(uint32_t)vb, (uint32_t)(vb >> 32) & 0xffffu, (uint32_t)((int64_t)HD * Skp * 2), 0x00020000u};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        rk[j] = __builtin_amdgcn_readfirstlane(rk[j]);
        rv[j] = __builtin_amdgcn_readfirstlane(rv[j]);
    }
#ifdef W64_DMA_WAVE
    // the generator's dma=wave: a wave's four pieces of a tile are contiguous in the image (K rows 16 wave + 4 i + (lane >> 4), V^T
    // rows 32 wave + 8 i + (lane >> 3)): one M0 write per four loads, the instruction offset i * 1024 moves the LDS address.  That
    // offset enters the global address too, so the K descriptor's base stands 1024 bytes low (piece 2's lane offset is its key
    // offset MINUS 1024) and every lane offset carries + 1024; the range grows by the same 1024.
    rk[0] = __builtin_amdgcn_readfirstlane((uint32_t)(kb - 1024));
    rk[1] = __builtin_amdgcn_readfirstlane((uint32_t)((kb - 1024) >> 32) & 0xffffu);
    rk[2] = __builtin_amdgcn_readfirstlane((uint32_t)(Sk * (HD * 2) + 1024));
    const int krow0 = 16 * wave + (lane >> 4);
    const int voff_k = krow0 * (HD * 2) + (((lane & 15) ^ (lane >> 4)) << 4) + 1024;
    const int vrow0 = 32 * wave + (lane >> 3);
    const int voff_v = vrow0 * Skp * 2 + (((lane & 7) ^ ((vrow0 >> 1) & 3)) << 4);
    const int v_piece = __builtin_amdgcn_readfirstlane(8 * Skp * 2 - 1024);
    const int wbase = wave * 4096;
#else
    const int krow0 = 4 * wave + (lane >> 4);
    const int voff_k = w64_perm32(krow0) * (HD * 2) + (((lane & 15) ^ krow0) << 4);
    const int vrow0 = 8 * wave + (lane >> 3);
    const int voff_v = vrow0 * Skp * 2 + (((lane & 7) ^ ((vrow0 >> 1) & 7)) << 4);
    const int v_piece = __builtin_amdgcn_readfirstlane(32 * Skp * 2);
    const int wbase = wave * 1024;
#endif
    // fragment reads: K image row l31 (+ 32 kt), chunk (2 ks + hi) ^ (row & 15) = ((hi ^ row) & 15) ^ 2 ks; V^T image row l31 (+ 32 dt),
    // chunk (2 kk + hi) ^ ((row >> 1) & 7)
    const int ka = l31 * 256 + (((hi ^ l31) & 15) << 4);
    const int va = l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4);

#if APEXMI_ATTN_TRACE
    unsigned long long* tp = d_attn_trace ? d_attn_trace + ((size_t)blockIdx.x * 4 + wave) * 4 : nullptr;
    tp = (unsigned long long*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)tp >> 32)) << 32) |
                               (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)tp));
#else
    unsigned long long* tp = nullptr;
#endif
    f32x16 o0, o1, o2, o3, o4, o5, o6, o7;
    float la, lb;
#ifdef W64_FALLBACK
    const int flag_at = 4 * ATT_STAGE;                     // one LDS word behind the K / V^T rings
    if (tid == 0) asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(flag_at), "v"(0) : "memory");
#endif
    asm volatile(
#include W64_BODY
        : "={a[0:15]}"(o0), "={a[16:31]}"(o1), "={a[32:47]}"(o2), "={a[48:63]}"(o3), "={a[64:79]}"(o4), "={a[80:95]}"(o5),
          "={a[96:111]}"(o6), "={a[112:127]}"(o7), [la] "=&v"(la), [lb] "=&v"(lb)
        : [qa] "v"(qa), [qb] "v"(qbp), [ka] "v"(ka), [va] "v"(va), [vk] "v"(voff_k), [vv] "v"(voff_v), [rk] "s"(rk), [rv] "s"(rv),
          [sc] "s"(scale_log2e), [nt] "s"(nt), [rem] "s"(rem), [vp] "s"(v_piece), [w] "s"(wbase), [tp] "s"(tp)
        : "memory", "vcc", "scc",
#include "attn_w64_clobbers.inc"
    );
#ifdef W64_FALLBACK
    {
        // !(x <= 2^60) is true for x > 2^60, inf and NaN.  The partial sums of the two half-rows (lanes l, l ^ 32) are positive, so
        // testing each half is testing the row.
        const bool bad = !(la <= 0x1p60f) || !(lb <= 0x1p60f);
        if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0)
            asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(flag_at), "v"(1) : "memory");
        __syncthreads();                                   // every wave has left the loop: the rings may be staged again
        int again;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(again) : "v"(flag_at) : "memory");
        if (__builtin_amdgcn_readfirstlane(again)) {       // workgroup-uniform
            if (tid == 0) atomicAdd(&d_attn_w64_fallbacks, 1u);
            asm volatile(
#include W64_FALLBACK
                : "={a[0:15]}"(o0), "={a[16:31]}"(o1), "={a[32:47]}"(o2), "={a[48:63]}"(o3), "={a[64:79]}"(o4), "={a[80:95]}"(o5),
                  "={a[96:111]}"(o6), "={a[112:127]}"(o7), [la] "=&v"(la), [lb] "=&v"(lb)
                : [qa] "v"(qa), [qb] "v"(qbp), [ka] "v"(ka), [va] "v"(va), [vk] "v"(voff_k), [vv] "v"(voff_v), [rk] "s"(rk),
                  [rv] "s"(rv), [sc] "s"(scale_log2e), [nt] "s"(nt), [rem] "s"(rem), [vp] "s"(v_piece), [w] "s"(wbase), [tp] "s"(tp)
                : "memory", "vcc", "scc",
#include "attn_w64_clobbers.inc"
            );
        }
    }
#endif

    // ---- epilogue: O[q][d] = O^T / l ; lane holds d = 32 dt + 8 g + 4 hi + (0..3).  The two half-rows (lanes l, l ^ 32) exchange
    // group pairs through v_permlane32_swap so that every lane stores 16 bytes (d = 32 dt + 8 g' .. + 7 with g' = 2 p + hi): 16
    // dwordx4 stores per lane instead of 32 dwordx2 — the store tail of a workgroup is issue-bound, and a Flux-shape workgroup
    // (72 tiles) pays it once per 72 tiles ----
    const int b = hb / H, h = hb % H;
    const f32x16* oo[2][4] = {{&o0, &o1, &o2, &o3}, {&o4, &o5, &o6, &o7}};
#ifdef W64_STORE_X2   // A/B arm (side library): the 8-byte stores of round 5
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float inv = 1.0f / sum_xor32(e ? lb : la);
        const int qrow = row_a + 32 * e;
        if (qrow < Sq) {
            bf16_t* op = O + (int64_t)b * o_sb + (int64_t)qrow * o_ss + (int64_t)h * o_sh;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16& x = *oo[e][dt];
                    u32x2 o;
                    o[0] = pack_bf16(x[4 * g + 0] * inv, x[4 * g + 1] * inv);
                    o[1] = pack_bf16(x[4 * g + 2] * inv, x[4 * g + 3] * inv);
                    *(u32x2*)(op + dt * 32 + g * 8 + hi * 4) = o;
                }
        }
    }
#else
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float inv = 1.0f / sum_xor32(e ? lb : la);
        const int qrow = row_a + 32 * e;
        bf16_t* op = O + (int64_t)b * o_sb + (int64_t)min(qrow, Sq - 1) * o_ss + (int64_t)h * o_sh;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const f32x16& x = *oo[e][dt];
                const int g0 = 2 * p, g1 = 2 * p + 1;
                const uint32_t a0 = pack_bf16(x[4 * g0 + 0] * inv, x[4 * g0 + 1] * inv);
                const uint32_t a1 = pack_bf16(x[4 * g0 + 2] * inv, x[4 * g0 + 3] * inv);
                const uint32_t b0 = pack_bf16(x[4 * g1 + 0] * inv, x[4 * g1 + 1] * inv);
                const uint32_t b1 = pack_bf16(x[4 * g1 + 2] * inv, x[4 * g1 + 3] * inv);
                // swap(a, b): {a[0:31] | b[0:31]}, {a[32:63] | b[32:63]} -> low lanes: own g0 + partner's g0; high: partner's g1 + own g1
                const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                if (qrow < Sq) *(u32x4*)(op + dt * 32 + (2 * p + hi) * 8) = o;
            }
    }
#endif
}

