/*
 * apexmi.h — C-ABI of libapex_mi355.so, the MI355X (gfx950) denoise hot path for
 * Apex Studio's render pipeline.
 *
 * Plain pointers, sizes and strides only: no torch types, no allocation inside the
 * library (callers pass outputs and workspace), every entry point enqueues on the
 * hipStream_t it is handed and returns without a host sync.  Return value: 0 = ok,
 * non-zero = error (apexmi_last_error() gives the text; the Python shim raises
 * RuntimeError with it, mirroring how exceptions propagate through
 * engine.run -> _run_engine_from_manifest_impl in the reference,
 * apps/api/src/api/ray_tasks.py:2677).
 *
 * Each entry point cites the reference interface (apps/api/src/...) it replaces.
 * All device pointers are bf16 (uint16 storage) unless stated; "f32" = float.
 */
#ifndef APEXMI_H
#define APEXMI_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* apexmi_stream_t; /* a hipStream_t; NULL = the null stream */

/* dtype codes (apexmi_attn_fwd generic path) */
#define APEXMI_BF16 0
#define APEXMI_F16 1
#define APEXMI_F32 2

/* GEMM epilogues */
#define APEXMI_EPI_BIAS 0          /* C = A W^T + b                                   */
#define APEXMI_EPI_BIAS_GELU 1     /* C = gelu_tanh(A W^T + b)                        */
#define APEXMI_EPI_BIAS_GATE_RES 2 /* C = R + gate[n] * (A W^T + b)   (R may alias C) */
#define APEXMI_EPI_BIAS_GELU_ERF 4 /* C = gelu_erf(A W^T + b): nn.GELU() of the HunyuanVideo-1.5 text/image projections */
#define APEXMI_EPI_BIAS_QUICK_GELU 6 /* C = x sigmoid(1.702 x), x = A W^T + b: CLIPMLP (transformers "quick_gelu")         */
#define APEXMI_EPI_BIAS_SILU 5     /* C = silu(A W^T + b): the "linear-silu" FeedForward of its token refiner        */
#define APEXMI_EPI_BIAS_F32 3      /* C = A W^T + b stored as float (C is float*, ldc in floats): attention scores */

/* OR-ed into any epilogue above: C and R are float (ldc / ldr in floats).  The f32-STORAGE VERIFICATION MODE (DESIGN.md
 * §1.2): the same kernels and epilogue formulas with no bf16 rounding at the store; the activation operand A is then the
 * exact three-way bf16 split written by apexmi_split_bf16x3 (K-concatenated, W repeated three times along K), so the
 * MFMA products are exact and the whole layer is f32-accurate.  Not a production path: 3x the MFMA work. */
#define APEXMI_EPI_F32_IO 0x100

/* GEMV flags */
#define APEXMI_GEMV_PRE_SILU 1   /* x <- silu(x) before the dot product           */
#define APEXMI_GEMV_POST_SILU 2  /* y <- silu(y)                                  */
#define APEXMI_GEMV_POST_GELU 4  /* y <- gelu_tanh(y)                             */
#define APEXMI_GEMV_ACCUM 8      /* y <- y_in + result                            */

/* qk_norm_rope modes */
#define APEXMI_ROPE_INTERLEAVED 0 /* pairs (2i,2i+1), cos/sin f32 [S, D] repeat-interleaved:
                                     diffusers apply_rotary_emb(use_real=True, unbind_dim=-1),
                                     called at transformer/flux/base/attention.py:86-87 */
#define APEXMI_ROPE_COMPLEX 1     /* pairs (2i,2i+1) times complex table f32 [S, D/2, 2]:
                                     apply_rotary_emb_qwen(use_real=False),
                                     transformer/qwenimage/base/model.py:100-151;
                                     wan rope, transformer/efficiency/ops.py:112-160 */
#define APEXMI_ROPE_NONE 2

int apexmi_version(void);
const char* apexmi_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Attention operator.  Replaces the callable registered in attention_register
 * (apps/api/src/attention/functions.py:84; default "sdpa" :338-377):
 *     fn(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, softmax_scale=None)
 * with q:[B,H,Sq,D], k,v:[B,H,Sk,D] possibly permuted views, result [B,H,Sq,D].
 * Strides are in ELEMENTS for (b, h, s); the D axis must be contiguous.
 * out is written as [B,Sq,H,D] with the given strides (so the caller can hand the
 * flux/wan processors the `.permute(0,2,1,3)` view they expect without a copy).
 * No mask, no dropout, non-causal (all hot-path call sites: SURVEY.md §2.4).
 * workspace: at least apexmi_attn_workspace_bytes(...) bytes of device memory
 * (holds V^T for the MFMA path); may be NULL when that returns 0.
 * ------------------------------------------------------------------------------------------- */
size_t apexmi_attn_workspace_bytes(int B, int H, int Sq, int Sk, int D, int dtype);
int apexmi_attn_fwd(const void* q, const void* k, const void* v, void* out,
                    int B, int H, int Sq, int Sk, int D,
                    const int64_t q_strides[3], const int64_t k_strides[3],
                    const int64_t v_strides[3], const int64_t o_strides[3],
                    float softmax_scale, int dtype,
                    void* workspace, size_t workspace_bytes, apexmi_stream_t stream);

/* MFMA flash-attention forward on prepared operands (bf16, D = 128):
 *   q  [B,H,Sq,128]  k [B,H,Sk,128]  vt [B,H,128,Skp]  (Skp = Sk rounded up to 64, zero padded)
 *   out[B,Sq,H,128] with element strides o_strides (b, s, h).
 * Same arithmetic as apexmi_attn_fwd; this is what the fused model path calls. */
int apexmi_attn_fwd_prepared(const void* q, const void* k, const void* vt, void* out,
                             int B, int H, int Sq, int Sk, int Skp,
                             const int64_t o_strides[3], float softmax_scale,
                             apexmi_stream_t stream);

/* Same, with scratch for the TAIL SPLIT: when the 8-wave launch would end in a round that keeps at most a quarter of the
 * CUs busy (QwenImage-Edit: 792 workgroups = 3 rounds + 24), those last workgroups run as a second launch cut into 4 key
 * ranges each and a merge of the partial results, instead of a nearly empty round of full-length workgroups.  The
 * partials are un-normalised f32 numerators with their (integer, base-2) row maxima and partial sums, merged with exact
 * power-of-two weights: a split launch rounds to bf16 exactly where the single launch does.  workspace >= apexmi_attn_prepared_workspace_bytes(B, H, Sq, Sk) (0 when no split applies; NULL / too
 * small simply disables the split). */
size_t apexmi_attn_prepared_workspace_bytes(int B, int H, int Sq, int Sk);
int apexmi_attn_fwd_prepared_ws(const void* q, const void* k, const void* vt, void* out, int B, int H, int Sq, int Sk,
                                int Skp, const int64_t o_strides[3], float softmax_scale, void* workspace,
                                size_t workspace_bytes, apexmi_stream_t stream);

/* The main launch of a large prepared attention (>= 140 workgroups of 256 query rows: attn_fwd_d128_w64_kernel) keeps, for every
 * query row, the INTEGER base-2 maximum its first 64 keys gave it and carries no per-tile running maximum (an integer shift of
 * the maximum scales probabilities, row sum and numerator by one power of two: same rounding points as the running-maximum loop).  At the end every row sum
 * is checked against 2^60; a workgroup in which one fails — a later score more than ~41 nats above the best of the first 64
 * keys, inf, NaN — recomputes its 256 rows with the running-maximum loop (softmax as R/src/attention/functions.py:338-377 defines
 * it for any input).  This returns the number of workgroups that took that second pass since the last call and clears the
 * count; it synchronises the device (a test / diagnostics call, not part of a step). */
int apexmi_attn_w64_fallbacks(uint64_t* count);

/* HunyuanVideo15AttnBlock.forward (vae/hunyuanvideo15/model.py:130-214): one head of C channels over frames x (H W)
 * tokens with the frame-causal mask of prepare_causal_attention_mask (:143-165): token i attends the keys of frames
 * <= its own, `block` = tokens per frame.  bf16, D = C a multiple of 128 up to 1024, any S; materialised through the GEMM
 * kernel; workspace >= apexmi_attn_framecausal_workspace_bytes(S, D), reused for every (batch, head). */
size_t apexmi_attn_framecausal_workspace_bytes(int S, int D);
int apexmi_attn_fwd_framecausal(const void* q, const void* k, const void* v, void* out, int B, int H, int S, int D,
                                int block, const int64_t q_strides[3], const int64_t k_strides[3],
                                const int64_t v_strides[3], const int64_t o_strides[3], float softmax_scale,
                                void* workspace, size_t workspace_bytes, apexmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Linear layers.  Replace torch.nn.Linear on the denoise path
 * (to_q/to_k/to_v/to_out, ff.net.0.proj/net.2, proj_mlp/proj_out:
 *  transformer/flux/base/model.py:106-129,180-182,258-263; wan model.py:551-712).
 * C[M,N] = epi(A[M,K] * W[N,K]^T + bias[N]);  A,W,C,R bf16 row-major with leading
 * dimensions lda/ldw/ldc/ldr (elements); bias bf16 [N] or NULL; gate f32 [N].
 * Requires K % 64 == 0, 16-byte aligned rows and lda, ldw <= 2^22 elements (the K-loops address a tile through 32-bit lane offsets).
 * ------------------------------------------------------------------------------------------- */
int apexmi_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias,
                     void* C, int64_t ldc, int M, int N, int K, int epilogue,
                     const float* gate, const void* R, int64_t ldr, apexmi_stream_t stream);

/* 1 when apexmi_gemm_bf16(M, N, K) with a bf16 epilogue would go out on the 288 x 192 EXACT-FILL tiling instead of 256 x 256
 * (round 5): launches whose 256 x 256 tiles leave a round of the 256 CUs part-filled and whose 288 x 192 tiles do not — the
 * Flux single block's proj_out, `nn.Linear(dim + mlp_hidden, dim)` at R/src/transformer/flux/base/model.py:195-227: 4608 x 3072 x
 * 15360 is 216 tiles of 256 x 256 (40 CUs idle) but 16 x 16 = 256 tiles of 288 x 192.  Results are bit-identical on both tilings
 * (same K order per output element); tune key "gemm.x288": 0 never (the shipped default: the tiling measured 3.5 % SLOWER on
 * that launch — a K-tile's time grows with the number of busy CUs, so idle CUs are not lost time; gemm.hip), 1 this rule, 2 always. */
int apexmi_gemm_uses_x288(int M, int N, int K);

/* Up to 4 problems that share K in ONE launch (per-problem M, N and epilogue; gate/residual
 * problems cannot be mixed with bias/gelu ones).  Two uses on the Flux path:
 *  - the image and text streams of an MM-DiT double block have separate weights but identical
 *    shapes (flux model.py:245-263); together they fill the 256 CUs instead of leaving the 512-row
 *    text GEMM on 28 % of the chip;
 *  - the single block's QKV projection and its MLP-up projection read the same normalised input
 *    (model.py:207-214); as one launch they are 1512 tiles = 5.9 rounds of 256 CUs instead of
 *    3 + 4 rounds.
 * Arrays are host arrays of length `count`; bias/gate/R/ldr may be NULL. */
int apexmi_gemm_bf16_grouped(int count, const void* const* A, const int64_t* lda,
                             const void* const* W, const int64_t* ldw, const void* const* bias,
                             void* const* C, const int64_t* ldc, const int* M, const int* N, int K,
                             const int* epilogue, const float* const* gate, const void* const* R,
                             const int64_t* ldr, apexmi_stream_t stream);

/* apexmi_gemm_bf16_grouped with the q/k/v preparation of the attention processors fused into the epilogue of the fused-QKV
 * projection (reference transformer/flux/base/attention.py:62-94: unflatten to heads, norm_q / norm_k (RMSNorm over the head,
 * eps), apply_rotary_emb on interleaved pairs, the [B, H, S, D] layout; plus V^T for this library's attention kernel) — what
 * apexmi_qkv_prepare does as a separate pass over the [S, 3 H 128] projection, with bit-identical results (the projection is
 * rounded to bf16 exactly where the separate path stores it, and the sums run in the same order).
 *   is_qkv[i] != 0: problem i is a fused QKV projection, N[i] = 3 H 128 ([q | k | v] rows of W); its M[i] rows are rows
 *     [row0[i], row0[i] + M[i]) of the joint sequence (any alignment: an unaligned stream stores its V^T element-wise); norm_q[i] / norm_k[i] = the RMSNorm weights (bf16[128]) of
 *     its stream (the text stream of a joint block brings norm_added_q / norm_added_k); C[i] / ldc[i] are ignored.
 *   is_qkv[i] == 0: an ordinary bias-class problem of the same launch (the single block's MLP up-projection with GELU).
 *   q_out, k_out: bf16 [H, S_out, 128]; vt_out: bf16 [H, 128, Skp] (Skp >= S_out, a multiple of 8; columns >= S_out are not
 *   written — allocate it zeroed); rope: f32 [2, S_out, 128] (cos | sin rows as apexmi_rope_table_axes writes them).
 * Exists on the shipped 256 x 256 v_mfma_f32_16x16x32 tiling only (total M >= 1024; gemm.config / gemm.large left at 7):
 * returns an error otherwise, and the caller keeps apexmi_gemm_bf16_grouped + apexmi_qkv_prepare. */
/* 1 when a grouped launch with this total row count, largest N and K would run on the tiling that has the fused epilogue. */
int apexmi_gemm_qkv_fusable(int64_t m_total, int n_max, int K);
int apexmi_gemm_bf16_grouped_qkv(int count, const void* const* A, const int64_t* lda, const void* const* W,
                                 const int64_t* ldw, const void* const* bias, void* const* C, const int64_t* ldc,
                                 const int* M, const int* N, int K, const int* epilogue, const int* is_qkv,
                                 const void* const* norm_q, const void* const* norm_k, const int* row0, int H, float eps,
                                 const float* rope, void* q_out, void* k_out, void* vt_out, int S_out, int Skp,
                                 apexmi_stream_t stream);
/* The same launch with the rotary table ALSO given as its compact copy `rope_pairs` (f32 [2, S_out, 64], may be NULL = the call
 * above): tables written by apexmi_rope_table_axes hold every cos / sin twice (get_1d_rotary_pos_embed(repeat_interleave_real=True),
 * flux model.py:338-359), and the q / k tiles of a launch pull their rows of the table through the L2 -> CU path once per (row, head) —
 * where they queue behind the K-loop staging traffic of the other CUs (measured: +45..57 us on the ~490 us single-block launch of
 * Flux).  With the compact copy the epilogue prefetches the rows four iterations ahead through a per-wave LDS ring (LDS-DMA, no
 * registers): +12 us.  Results are bit-identical (the same values reach the same arithmetic in the same order). */
int apexmi_gemm_bf16_grouped_qkv_pairs(int count, const void* const* A, const int64_t* lda, const void* const* W,
                                       const int64_t* ldw, const void* const* bias, void* const* C, const int64_t* ldc,
                                       const int* M, const int* N, int K, const int* epilogue, const int* is_qkv,
                                       const void* const* norm_q, const void* const* norm_k, const int* row0, int H, float eps,
                                       const float* rope, const float* rope_pairs, void* q_out, void* k_out, void* vt_out,
                                       int S_out, int Skp, apexmi_stream_t stream);
/* pairs[t][s][q] = rope[t][s][2 q] for a table rope f32 [2, S, D] (t = cos | sin); *mismatch (device int, zeroed by the caller) is
 * incremented for every pair whose two entries differ bit-wise — such a table has no compact copy. */
int apexmi_rope_pairs(const float* rope, int S, int D, float* pairs, int* mismatch, apexmi_stream_t stream);

/* out[m][j K + k] (bf16, j = 0..2) = the j-th part of the exact split x = hi + mid + lo of the float x[m][k]:
 * hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid (representable).  ldx in floats, ldo >= 3 K in bf16 elements,
 * K % 8 == 0.  Operand preparation of the f32-storage verification mode (APEXMI_EPI_F32_IO; apexmi_conv3d_cl_f32). */
int apexmi_split_bf16x3(const float* x, int64_t ldx, int64_t M, int K, void* out, int64_t ldo, apexmi_stream_t stream);

/* Batched C[z] = A[z] W[z]^T over blockIdx.y (z < batch <= 65535): per-head GEMMs of one attention layer in one launch.
 * Strides in elements; epilogue APEXMI_EPI_BIAS (bf16 C, no bias) or APEXMI_EPI_BIAS_F32 (float C).  Same kernels and
 * argument rules as apexmi_gemm_bf16. */
int apexmi_gemm_bf16_batched(const void* A, int64_t lda, int64_t stride_a, const void* W, int64_t ldw, int64_t stride_w,
                             void* C, int64_t ldc, int64_t stride_c, int batch, int M, int N, int K, int epilogue,
                             apexmi_stream_t stream);

/* Self-attention of the text encoders the reference loads by class name from `transformers` (pinned 4.57.1,
 * R/requirements/requirements.txt:79; call site R/src/text_encoder/text_encoder.py:335-342):
 *   T5Attention / UMT5Attention.forward: softmax(q k^T + position_bias + mask) v, NO 1/sqrt(d) scaling (scale = 1);
 *   CLIPAttention.forward: softmax(q k^T / sqrt(d) + causal mask) v.
 *   Qwen2_5_VLAttention / Qwen2_5_VLVisionAttention.forward: grouped-query causal attention with a padding mask /
 *   block-diagonal (window or per-image) attention.
 * q, out: bf16 [Sq, H*D]; k, v: bf16 [Sk, Hkv*D] (row strides ldq/ldk/ldv/ldo), head h in columns [h D, (h+1) D) and
 * query head h reading key/value head h / (H / Hkv).  bias: f32 [H, Sq, Sk] or NULL; keep: uint8 [Sk], 0 = padded key
 * (NULL = all kept); seg: int32 [S] segment id per token, a query sees only keys of its own segment (NULL = one
 * segment); causal != 0 adds the causal mask.  Masked keys get probability exactly 0 (what the additive finfo.min mask
 * gives in f32).  D a multiple of 64, Hkv*D of 128.  Materialised: batched scores GEMM (f32; stride 0 on the shared
 * key head for GQA), row softmax, batched P V GEMM; workspace from the _bytes query. */
size_t apexmi_attn_bias_workspace_bytes(int H, int Sq, int Sk, int D);
int apexmi_attn_fwd_bias(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* out,
                         int64_t ldo, int H, int Hkv, int Sq, int Sk, int D, float softmax_scale, const float* bias,
                         const uint8_t* keep, const int* seg, int causal, void* workspace, size_t workspace_bytes,
                         apexmi_stream_t stream);

/* Tuning knobs for A/B measurements (bench.py --tune, tests); defaults are the shipped choices:
 *   "gemm.config"  0 auto | 1 128x128 | 2 256x256 | 3 256x256 ping-pong 32x32x16 | 6 one wave per SIMD | 7 ping-pong 16x16x32
 *   "gemm.large"   tiling the auto rule picks for large problems (7)      "gemm.group_m"  rows of a tile-order group (8)
 *   "conv.v2"      1 (default): stride-1 zero-padded convolutions over >= 65536 positions use the conv-shaped tiles
 *                  (conv3d_v2_kernel: 512x96 / 256x192 / 256x256 / 512x32|64 by Cout); 0: always the 128x128 kernel.
 *                  Bit-identical results either way.
 *   "conv.slab"    2 (default): stride-1 3x3 (x kT <= 3) convolutions over >= 65536 positions with Cin a multiple of 48 and
 *                  Cout <= 192 or a multiple of 192 (zero padding), or Cin a multiple of 64 and Cout a multiple of 128 (zero or
 *                  replicate padding), run as a DIRECT convolution (conv3d_slab_kernel:
 *                  haloed input slab per temporal tap and 48-channel slice staged once, spatial taps as shifted LDS reads);
 *                  same products as the implicit GEMM, f32 sums in another order (<= 1 bf16 ulp on a few 1e-4 of the outputs).
 *                  1: Cin = 96 / Cout <= 96 layers on the order-preserving 8 x 32 form (bit-identical to the implicit GEMM).
 *                  0: implicit GEMM only.
 *   "gemm.tail"    1: a small last problem of a grouped launch after whole rounds of tiles goes out on the 128x128 tiling
 *   "attn.waves"   0 auto | 4..8 waves per attention workgroup            "attn.mfma"     32 | 16
 *   "attn.c4"      0: plain loop | 1 + bits: 4-cluster ping-pong kernel, bit 0 s_setprio around the matrix clusters, bit 1
 *                  packed-f32 softmax (v_pk_fma_f32 / v_pk_add_f32), bit 2 static priority for waves 4..7.  Default 3
 *                  (= packed softmax: +0.5..0.9 % on the Flux / Qwen / Wan shapes; the other two bits measured neutral / -0.5 %)
 *   "attn.split"   1: a nearly empty last round runs as 4 key ranges + merge (needs the _ws entry point's scratch)
 *   "attn.w64"     1 (default): main launch of >= 140 workgroups on the one-wave-per-SIMD kernel, first-tile maximum + checked
 *                  fallback (apexmi_attn_w64_fallbacks) | 8: the same kernel with the per-tile running maximum | 0: 4-cluster kernel
 *   "qk.group"     1: q/k norm + RoPE four heads per lane group with the V transpose in the same launch | 2: without | 0: one head
 *   "ln.wave"      1: wave-per-row LayerNorm kernel for C in {3072, 3584, 5120}
 * Returns non-zero for an unknown key. */
int apexmi_tune_set(const char* key, int value);

/* y[m, n] = post( dot(W[n, :], pre(x[m, :])) + bias[n] ) for tiny M (conditioning vectors:
 * time_text_embed, norm*.linear AdaLN projections — diffusers layers restated in
 * SURVEY.md App. A; call sites flux model.py:179,242-243,430-437,464).
 * W bf16 [N,K] (ldw), bias bf16 [N] or NULL, x f32 [M,K], y f32 [M,N] (ldy). M <= 8. */
int apexmi_gemv(const void* W, int64_t ldw, const void* bias, const float* x, int64_t ldx,
                float* y, int64_t ldy, int M, int N, int K, int flags, apexmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * AdaLN family: out = LayerNorm(x; eps, no affine) * (1 + scale[c]) + shift[c]
 * (AdaLayerNormZero/Single/Continuous and `norm2 * (1+scale) + shift`,
 *  flux model.py:273-279,299-302,647; wan model.py:56-116 `_chunked_modulated_norm`).
 * x,out bf16 [M,C] (ldx/ldo); scale,shift f32 [C] or NULL (=> plain LayerNorm);
 * gamma,beta bf16 [C] or NULL (affine FP32LayerNorm, wan norm2). stats in f32.
 * rms != 0 selects RMSNorm (x * rsqrt(mean(x^2)+eps) * gamma) instead
 * (InplaceRMSNorm, transformer/efficiency/mod.py:24-35 intended semantics; qwen txt_norm).
 * ------------------------------------------------------------------------------------------- */
int apexmi_ln_modulate(const void* x, int64_t ldx, void* out, int64_t ldo, int M, int C,
                       const float* scale, const float* shift, const void* gamma,
                       const void* beta, float eps, int rms, apexmi_stream_t stream);

/* Same, over a joint [text rows | image rows] buffer: rows < split use (scale2, shift2) — the two
 * streams of an MM-DiT double block in one launch (norm1/norm1_context, norm2/norm2_context). */
int apexmi_ln_modulate2(const void* x, int64_t ldx, void* out, int64_t ldo, int M, int C,
                        const float* scale, const float* shift, const void* gamma, const void* beta,
                        float eps, int rms, int split, const float* scale2, const float* shift2,
                        apexmi_stream_t stream);

/* Wan's q / k preparation in ONE pass (reference transformer/wan/base/attention.py:305-413; InplaceRMSNorm,
 * transformer/efficiency/mod.py:24-35; apply_wan_rope_inplace, transformer/efficiency/ops.py:112-160): RMSNorm over ALL H * 128
 * channels of every q and k row (affine weights wq / wk of H * 128 elements, bf16), the result rounded to the storage type where
 * the reference's in-place norm writes it, rotary embedding (rope_mode as apexmi_qkv_prepare), layout [H, S_out, 128]; v (optional)
 * leaves transposed [H, 128, Skp].  Equals apexmi_ln_modulate2(rms) on q, on k, then apexmi_qkv_prepare without norm weights, bit
 * for bit, in one read of the projection.  k / ko and v / vt may be NULL together (the query side of cross-attention).
 * H * 128 in {3072, 5120} (the widths whose stand-alone norm uses the same one-wave-per-row reduction).  _f32: float q / k / v / outputs (f32-storage verification mode). */
int apexmi_qk_rms_rope_rows(const void* q, const void* k, const void* v, int64_t ld_in, int S, int H, const void* wq, const void* wk,
                            float eps, const float* rope, int rope_mode, void* qo, void* ko, void* vt, int S_out, int Skp,
                            int row0, apexmi_stream_t stream);
int apexmi_qk_rms_rope_rows_f32(const void* q, const void* k, const void* v, int64_t ld_in, int S, int H, const void* wq,
                                const void* wk, float eps, const float* rope, int rope_mode, void* qo, void* ko, void* vt,
                                int S_out, int Skp, int row0, apexmi_stream_t stream);

/* Per-head RMSNorm on q,k + rotary embedding, written in attention layout, and V transposed.
 * Replaces the unflatten / norm_q / norm_k / cat / apply_rotary_emb / permute chain of
 * FluxAttnProcessor.__call__ (transformer/flux/base/attention.py:62-94).
 *   q,k,v : bf16 [S, H*D] row pointers with row stride ld_in (elements) each (a fused
 *           [S, 3*H*D] projection output passes three offsets into one buffer); k and v may be
 *           NULL (cross-attention prepares the query side and the text key/value side separately)
 *   wq,wk : bf16 [D] RMSNorm weights used for rows >= split;  wq2,wk2 for rows < split
 *           (the text stream's norm_added_q/k; pass split = 0 to use only wq,wk). NULL = no norm.
 *   rope  : f32 table, layout per rope_mode, indexed by row s
 *   qo,ko : bf16 [H, S_out, D] at row offset row0 (so two streams can fill one joint buffer)
 *   vt    : bf16 [H, D, Skp]   (column s + row0)
 * D must be 128. */
int apexmi_qkv_prepare(const void* q, const void* k, const void* v, int64_t ld_in,
                       int S, int H, int D, int split,
                       const void* wq, const void* wk, const void* wq2, const void* wk2,
                       float eps, const float* rope, int rope_mode,
                       void* qo, void* ko, void* vt, int S_out, int Skp, int row0,
                       apexmi_stream_t stream);

/* V^T only: v bf16 [S, H*D] (row stride ld, head stride D) -> vt [H, D, Skp] at column row0.
 * Generic strides (elements): v_strides = (h, s). */
int apexmi_v_transpose(const void* v, int64_t v_stride_h, int64_t v_stride_s, int S, int H, int D,
                       void* vt, int Skp, int row0, apexmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * 3-D causal VAE decode (Wan 2.x / QwenImage), channels-last [T, H, W, C] bf16.
 * apexmi_conv3d_cl: WanCausalConv3d.forward (vae/wan/model.py:178-185; time padding 2*(kT-1)/2... all on
 *   the left), nn.Conv2d 3x3 of WanResample (:264-273) with kT = 1, and the 1x1 convolutions.
 *   w is pre-packed [Cout, Kpad] bf16 with k = tap * Cin + ci, tap = (kt * kH + ky) * kW + kx, Kpad =
 *   taps*Cin rounded up to 64 (zero filled); bias [Cout] or NULL; residual [T,H,W,Cout] or NULL (added
 *   after the bias: the `x + h` of WanResidualBlock.forward :441); zeros = any 16 zero bytes on the device.
 *   Cin % 8 == 0, Cout % 4 == 0.
 * apexmi_rmsnorm_cl: WanRMS_norm.forward (:216-222) per position over C channels, optional SiLU.
 * apexmi_upsample2x_cl: WanUpsample nearest-exact 2x (:225-237).
 * apexmi_time_interleave_cl: [T,H,W,2C] -> [2T,H,W,C] (WanResample.forward :332-336).
 * apexmi_crossfade: b[o,e,i] = a[o,e,i] (1 - e/E) + b[o,e,i] e/E (blend_v / blend_h :1404-1422),
 *   element strides (outer, e) for a and b, inner contiguous.
 * ------------------------------------------------------------------------------------------- */
int apexmi_conv3d_cl(const void* in, const void* w, const void* bias, const void* residual, void* out,
                     const void* zeros, int T, int H, int W, int Cin, int Cout, int Kpad, int kT, int kH,
                     int kW, apexmi_stream_t stream);
/* apexmi_conv3d_cl (replicate = 0) / apexmi_conv3d_cl_replicate (1) over T / clip_frames independent CLIPS stacked along T: the
 * spatial tiles of a tiled VAE decode in ONE launch where a single tile's convolution would launch fewer workgroups than the
 * chip has slots (the 1024-channel 8 x 8-latent stages of the HunyuanVideo-1.5 decoder, R/src/vae/hunyuanvideo15/model.py:1060-1119:
 * 45 tiles of 1984 positions).  The causal temporal taps stop at every clip's first frame (zero / replicated there, as for a
 * single clip); each frame is its own image spatially.  Bit-identical to one call per clip.  clip_frames must divide T. */
int apexmi_conv3d_cl_clips(const void* in, const void* w, const void* bias, const void* residual, void* out, const void* zeros,
                           int T, int H, int W, int Cin, int Cout, int Kpad, int kT, int kH, int kW, int replicate,
                           int clip_frames, apexmi_stream_t stream);

/* The same convolution read THROUGH a nearest 2x spatial upsample: in is [T, H, W, Cin], out [T, 2H, 2W, Cout], tap
 * (y, x) of the upsampled image reads stored pixel (y >> 1, x >> 1).  Replaces WanUpsample (nearest-exact 2x,
 * vae/wan/model.py:225-237) + the Conv2d of WanResample "upsample2d/3d" (:264-273) and diffusers' Upsample2D of the Flux
 * VAE without materialising the 4x larger image.  independent != 0: frames are independent
 * images (see apexmi_conv3d_cl_frames). */
int apexmi_conv3d_cl_up2(const void* in, const void* w, const void* bias, const void* residual, void* out,
                         const void* zeros, int T, int H, int W, int Cin, int Cout, int Kpad, int kT, int kH, int kW,
                         int independent, apexmi_stream_t stream);

/* Convolution with the RMS norm of its OUTPUT fused into the epilogue ("RMS-norm(channel) + SiLU"
 * of WanResidualBlock.forward, vae/wan/model.py:389-441, and of the decoder's norm_out :1011-1017, moved from the
 * consumer's prologue — where LDS-DMA staging bypasses the registers — into the PRODUCER's epilogue): besides (or instead
 * of: out may be NULL) y = conv(in) + bias (+ residual) it writes out_norm = [silu](y / max(||y||_2, 1e-12) * sqrt(Cout)
 * * gamma) per position, computed from the bf16-rounded y exactly as the separate apexmi_rmsnorm_cl pass would read it
 * back.  Only for shapes whose every output channel of a position lies in one workgroup tile of the conv-shaped tilings:
 * apexmi_conv3d_cl_norm_fusable(T, H, W, Cin, Cout, up) != 0 (H, W = stored extents).  up != 0: read through the nearest
 * 2x upsample as apexmi_conv3d_cl_up2. */
int apexmi_conv3d_cl_norm_fusable(int T, int H, int W, int Cin, int Cout, int up);
int apexmi_conv3d_cl_norm(const void* in, const void* w, const void* bias, const void* residual, void* out, void* out_norm,
                          const void* gamma, int silu, const void* zeros, int T, int H, int W, int Cin, int Cout, int Kpad,
                          int kT, int kH, int kW, int independent, int up, apexmi_stream_t stream);

/* Convolution with an activation in the epilogue — the TAEHV "light VAE" blocks (vae/tae/model.py:20-45: `conv, act`,
 * `act(conv(cat[x, past]) + skip(x))`; selected by `use_light_vae`, vae/hunyuanvideo15/model.py:958-962, 1163-1234):
 * out = act(conv(in) + bias (+ residual)) evaluated in f32, ONE bf16 rounding.  act: 0 none | 1 leaky ReLU with `slope`
 * (0.0 = ReLU).  MemBlock's `conv(torch.cat([x, past], 1))` (:44, past = the previous frame, zeros before the first) IS a
 * causal kT = 2 convolution: pack the [Cout, 2 Cin, 3, 3] weight with temporal tap 1 <- input channels [0, Cin) and tap 0 <-
 * [Cin, 2 Cin).  independent / up as in apexmi_conv3d_cl_frames / apexmi_conv3d_cl_up2 (nn.Upsample(2) :239-251 folded
 * into the following convolution's gather). */
int apexmi_conv3d_cl_act(const void* in, const void* w, const void* bias, const void* residual, void* out,
                         const void* zeros, int T, int H, int W, int Cin, int Cout, int Kpad, int kT, int kH, int kW,
                         int independent, int up, int act, float slope, apexmi_stream_t stream);
/* TAEHV's input clamp behind the light VAE's 1/scaling_factor (vae/tae/model.py:24-26; hunyuanvideo15/model.py:1225):
 * y = 3 tanh(x * inv_scale / 3) over n packed bf16 values (n % 8 == 0). */
int apexmi_tanh_clamp(const void* x, void* y, int64_t n, float inv_scale, apexmi_stream_t stream);
/* TAEHV's output tail (vae/tae/model.py:318-333): clamp to [lo, hi], F.pixel_shuffle by r in {1, 2} (channel c r^2 + i r + j
 * -> pixel (h r + i, w r + j) of image channel c) and drop the first t0 frames; x [T, H, W, Cs] channels-last (Cs >= C r^2),
 * y [C, T - t0, H r, W r]. */
int apexmi_pixel_shuffle_clamp(const void* x, void* y, int T, int H, int W, int Cs, int C, int r, int t0, float lo, float hi,
                               apexmi_stream_t stream);

/* N INDEPENDENT single-frame clips in one launch: in / out are [N, H, W, C] and every frame is convolved as if it were
 * a one-frame clip — of a causal kT-tap kernel only the last temporal tap touches data, the others fall in the zero
 * padding, so the launch iterates kH*kW taps from the last temporal slice of the packed weight.  (A single-frame call
 * of apexmi_conv3d_cl does the same by itself.)  This is how the 36 spatial tiles of a 1024x1024 QwenImage VAE
 * decode / encode run as 4 shape groups instead of 36 tile passes; results equal the per-tile calls bit for bit.
 * Kpad must leave room for that slice: Kpad >= (kT-1) kH kW Cin + round_up(kH kW Cin, 64). */
int apexmi_conv3d_cl_frames(const void* in, const void* w, const void* bias, const void* residual, void* out,
                            const void* zeros, int N, int H, int W, int Cin, int Cout, int Kpad, int kT, int kH, int kW,
                            apexmi_stream_t stream);

/* Strided variant: out[t, y, x] reads in[t + dt - (kT-1), y stride_h + dy - pad_top, x stride_w + dx - pad_left], taps
 * outside the input read zeros; out is [T, Ho, Wo, Cout].  `nn.ZeroPad2d((0, 1, 0, 1)) + nn.Conv2d(dim, dim, 3, stride=2)`
 * of WanResample "downsample2d/3d" (R/src/vae/wan/model.py:276-283) is kT=1, kH=kW=3, stride 2, pad_top=pad_left=0,
 * Ho = H / 2, Wo = W / 2: the VAE encoders' spatial downsampling without a padded copy. */
int apexmi_conv3d_cl_strided(const void* in, const void* w, const void* bias, const void* residual, void* out,
                             const void* zeros, int T, int H, int W, int Cin, int Cout, int Kpad, int kT, int kH, int kW,
                             int stride_h, int stride_w, int pad_top, int pad_left, int Ho, int Wo,
                             apexmi_stream_t stream);

/* Temporal stride: output frame j (0 <= j < To) is the causal convolution ENDING at input frame j * stride_t + t_first, i.e.
 * it reads frames j * stride_t + t_first + dt - (kT - 1); out is [To, H, W, Cout].  WanResample "downsample3d" in its
 * full-sequence form (vae/wan/model.py:340-365: `time_conv` = Conv3d((3,1,1), stride (2,1,1)) over the cached last frame +
 * the chunk): frame 0 passes through and output j >= 1 convolves frames (2j-2, 2j-1, 2j) = stride_t 2, t_first 2,
 * To = (T - 1) / 2 — half the work of convolving every frame and keeping every other one. */
int apexmi_conv3d_cl_tstrided(const void* in, const void* w, const void* bias, const void* residual, void* out,
                              const void* zeros, int T, int H, int W, int Cin, int Cout, int Kpad, int kT, int kH, int kW,
                              int stride_t, int t_first, int To, apexmi_stream_t stream);

/* HunyuanVideo15CausalConv3d.forward (vae/hunyuanvideo15/model.py:52-90): the same implicit GEMM with REPLICATE padding
 * (coordinates clamped: two frames in front, one pixel around) instead of zeros. */
int apexmi_conv3d_cl_replicate(const void* in, const void* w, const void* bias, const void* residual, void* out,
                               const void* zeros, int T, int H, int W, int Cin, int Cout, int Kpad, int kT, int kH,
                               int kW, apexmi_stream_t stream);
int apexmi_rmsnorm_cl(const void* x, void* y, const void* gamma, int64_t P, int C, int silu,
                      apexmi_stream_t stream);
int apexmi_upsample2x_cl(const void* x, void* y, int T, int H, int W, int C, apexmi_stream_t stream);
int apexmi_time_interleave_cl(const void* x, void* y, int T, int64_t HW, int C, apexmi_stream_t stream);
/* GroupNorm(G, C, eps) [+ SiLU] over a channels-last image x [P, C] (P = H*W positions) for the Flux 2-D
 * VAE decoder (diffusers Decoder / ResnetBlock2D; reference vae/auto/model.py:35-41).
 * gamma, beta bf16 [C]; workspace: apexmi_groupnorm_workspace_bytes(P, C) device bytes. */
size_t apexmi_groupnorm_workspace_bytes(int64_t P, int C);
int apexmi_groupnorm_cl(const void* x, void* y, const void* gamma, const void* beta, int64_t P, int C, int G,
                        float eps, int silu, void* workspace, size_t workspace_bytes, apexmi_stream_t stream);
int apexmi_crossfade(const void* a, void* b, int64_t outer, int E, int64_t inner, int64_t a_so, int64_t a_se,
                     int64_t b_so, int64_t b_se, apexmi_stream_t stream);

/* Sinusoidal timestep embedding (diffusers Timesteps(num_channels, flip_sin_to_cos=True,
 * downscale_freq_shift=0, scale); in-tree copy transformer/qwenimage/base/model.py:46-97).
 * t f32 [M] (device), out f32 [M, dim].  `freqs` (device f32 [dim/2], may be NULL) is the layer's frequency table
 * exp(-ln(10000) i / (dim/2 - shift)); when given it is used verbatim — the host computes it once with the very f32
 * operation sequence of the reference, so the sin / cos ARGUMENTS t*freq*scale are bit-identical to the reference's
 * (a 1-ulp difference in a frequency is a 5e-5 phase error at t = 1000). */
int apexmi_timestep_embedding(const float* t, float* out, int M, int dim, float scale,
                              int flip_sin_to_cos, float downscale_freq_shift, const float* freqs,
                              apexmi_stream_t stream);

/* Rotary table for multi-axis positions (FluxPosEmbed.forward, flux model.py:338-359, i.e.
 * diffusers get_1d_rotary_pos_embed(use_real=True, repeat_interleave_real=True, freqs_dtype=f64)
 * per axis, concatenated): ids f32 [S, n_axes] (device), axes_dim host ints (sum = D),
 * out f32 [2, S, D] = cos plane then sin plane (the APEXMI_ROPE_INTERLEAVED layout).
 * Angles are computed in f64 like the reference. */
int apexmi_rope_table_axes(const float* ids, int S, int n_axes, const int* axes_dim, float theta,
                           float* out, apexmi_stream_t stream);

/* out[l, i] = a[l, i] + b[i] in f32: `scale_shift_table + temb.float()` of every Wan block in one pass
 * (wan model.py:1117-1128, :1849-1856). */
int apexmi_add_bcast_f32(const float* a, const float* b, float* out, int64_t rows, int64_t n,
                         apexmi_stream_t stream);

/* out[r, :] = x[r, :] + v (bf16, f32 add): token stream + type embedding
 * (hunyuanvideo15 model.py:1013-1056 `encoder_hidden_states + cond_type_embed(...)`). */
int apexmi_add_rowvec_bf16(const void* x, int64_t ldx, const void* v, void* out, int64_t ldo, int64_t rows,
                           int cols, apexmi_stream_t stream);
/* the same with float x / out (f32-storage verification mode; v stays a bf16 weight) */
int apexmi_add_rowvec_f32(const void* x, int64_t ldx, const void* v, void* out, int64_t ldo, int64_t rows,
                           int cols, apexmi_stream_t stream);

/* out = a + b, n bf16 elements (n % 8 == 0): `h + shortcut` after the DCAE rearranges of the HunyuanVideo-1.5 VAE
 * (vae/hunyuanvideo15/model.py:274, :709-711). */
int apexmi_add_bf16(const void* a, const void* b, void* out, int64_t n, apexmi_stream_t stream);
/* the same on float tensors (f32-storage verification mode) */
int apexmi_add_f32(const void* a, const void* b, void* out, int64_t n, apexmi_stream_t stream);

/* out[p, c] = mean_{g < gs} x[p, c * gs + g] (f32 sum, one bf16 rounding), x bf16 [P, C * gs], out bf16 [P, C]: the
 * grouped channel mean of the DCAE shortcuts of the HunyuanVideo-1.5 VAE ENCODER (vae/hunyuanvideo15/model.py:318-331
 * `x.view(B, C, group_size, T, H, W).mean(dim=2)` of HunyuanVideo15Downsample, :622-625 of Encoder3D.forward), in
 * channels-last form.  gs in [1, 64]. */
int apexmi_group_mean_bf16(const void* x, void* out, int64_t P, int C, int gs, apexmi_stream_t stream);

/* BaseEngine._tensor_to_frames (engine/base_engine.py:2945-2949 -> diffusers VideoProcessor.postprocess_video):
 * frames uint8 [T, H, W, C] = round(clamp(video / 2 + 1/2, 0, 1) * 255) from a bf16 video [C, T, H, W] given by element
 * strides (a planar decode output or a channels-last tile alike); the intermediate is rounded to bf16 as the reference's
 * bf16 denormalize does, so the bytes are identical.  C <= 4. */
int apexmi_frames_to_u8(const void* video, int64_t stride_c, int64_t stride_t, int64_t stride_h, int64_t stride_w,
                        int C, int T, int H, int W, void* out, apexmi_stream_t stream);

/* In-place rotary embedding of the "rotate_half" form, x <- x cos + rotate_half(x) sin, on a packed projection
 * x bf16 [rows, heads * head_stride] (row stride ldx): each head rotates its first D columns with the row's
 * cos / sin f32 [rows, D] (transformers apply_rotary_pos_emb_vision / apply_multimodal_rotary_pos_emb of Qwen2.5-VL;
 * the caller builds the tables — 2-D patch positions or the 3-D mrope sections). */
int apexmi_rope_half(void* x, int64_t ldx, int64_t rows, int heads, int head_stride, int D, const float* cos_table,
                     const float* sin_table, apexmi_stream_t stream);

/* out = a * b, contiguous bf16, n a multiple of 8 (T5DenseGatedActDense: hidden_gelu * hidden_linear). */
int apexmi_mul_bf16(const void* a, const void* b, void* out, int64_t n, apexmi_stream_t stream);

/* out[r, :] = table[ids[r], :] (+ pos[r % period, :] when pos != NULL): the nn.Embedding lookups of T5Stack
 * (`shared`) and CLIPTextEmbeddings (token + position).  ids int64 on the device, clamped to [0, vocab). */
int apexmi_gather_rows_bf16(const void* table, int64_t ldt, int64_t vocab, const int64_t* ids, const void* pos,
                            int64_t ldp, int period, void* out, int64_t ldo, int64_t rows, int C,
                            apexmi_stream_t stream);

/* T5Attention.compute_bias: out[h, i, j] = weight[bucket[j - i + Sq - 1], h]; weight bf16 [num_buckets, H]
 * (relative_attention_bias.weight), bucket int32 [Sq + Sk - 1] on the device = _relative_position_bucket of every
 * distance, computed by the caller. */
int apexmi_relpos_bias(const void* weight, int num_buckets, int H, const int* bucket, int Sq, int Sk, float* out,
                       apexmi_stream_t stream);

/* f32 <-> bf16 helpers for the small conditioning vectors. */
int apexmi_cast_f32_to_bf16(const float* x, void* out, int64_t n, apexmi_stream_t stream);
int apexmi_cast_bf16_to_f32(const void* x, float* out, int64_t n, apexmi_stream_t stream);

/* FP-scaled checkpoint weights (`*_fp8_e4m3fn_scaled` files with `scale_weight` keys): the reference
 * dequantises on every forward, `weight.to(dtype) * scale_weight.to(dtype)` (quantize/scaled_layer.py:154-167,
 * :496-549; scale is a scalar or one value per output row).  Here it runs once at load:
 * out[r, c] (bf16, row stride ldo) = bf16( float(fp8 w[r, c]) * float(bf16 scale[r or 0]) ), i.e. exactly the
 * bf16 x bf16 product torch computes.  format: 0 = float8_e4m3fn, 1 = float8_e5m2; scale_count = 1 or rows. */
int apexmi_dequant_fp8_scaled(const void* w, int format, const void* scale, int64_t scale_count, int64_t rows,
                              int64_t cols, void* out, int64_t ldo, apexmi_stream_t stream);

/* Scheduler step on device (the loop stays in Python; this is the per-step axpy of
 * FlowMatchEulerDiscreteScheduler.step: prev = sample + dt * model_output in f32,
 * cast back).  sample/out: bf16 or f32 per sample_dtype; v bf16. */
int apexmi_euler_step(const void* sample, const void* model_out, void* out, int64_t n,
                      float dt, int sample_dtype, apexmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * f32-STORAGE VERIFICATION MODE (first route).  BASELINE.json's north_star asks for
 * decoded frames within 1e-3 of the reference's CPU fp32 path; a chain of kernels that ROUNDS every activation to bf16
 * cannot be held to that (§1.1), so the library also runs the SAME kernels with float activation storage:
 *   - every `_f32` entry point below is the template instantiation T = float of the kernel behind the entry point of the
 *     same name without the suffix (identical arguments; activation pointers are float*, leading dimensions in floats;
 *     weights, norm gains and biases stay bf16, modulation vectors stay f32);
 *   - the MFMA kernels (apexmi_gemm_bf16* with APEXMI_EPI_F32_IO, apexmi_conv3d_cl_f32) take the activation operand as
 *     its exact three-way bf16 split (apexmi_split_bf16x3), so their products are exact and they accumulate in f32;
 *   - attention runs in f32 arithmetic (apexmi_attn_fwd with APEXMI_F32; apexmi_attn_fwd_prepared_f32).
 * A model built with activation storage float32 (`storage_dtype=torch.float32` on the Python classes) calls only these;
 * tests/test_gpu_f32_storage.py holds free-running forwards, sampler chains and decoded frames to <= 1e-3 of the fp32
 * oracle with it.  Verification only: 3x the MFMA work, 2x the bytes.
 * ------------------------------------------------------------------------------------------- */
int apexmi_ln_modulate2_f32(const void* x, int64_t ldx, void* out, int64_t ldo, int M, int C, const float* scale,
                            const float* shift, const void* gamma, const void* beta, float eps, int rms, int split,
                            const float* scale2, const float* shift2, apexmi_stream_t stream);
int apexmi_qkv_prepare_f32(const void* q, const void* k, const void* v, int64_t ld_in, int S, int H, int D, int split,
                           const void* wq, const void* wk, const void* wq2, const void* wk2, float eps, const float* rope,
                           int rope_mode, void* qo, void* ko, void* vt, int S_out, int Skp, int row0,
                           apexmi_stream_t stream);
/* q [B,H,Sq,128], k [B,H,Sk,128], vt [B,H,128,Skp] float (what apexmi_qkv_prepare_f32 writes); out float with element
 * strides o_strides (b, s, h). */
int apexmi_attn_fwd_prepared_f32(const void* q, const void* k, const void* vt, void* out, int B, int H, int Sq, int Sk,
                                 int Skp, const int64_t o_strides[3], float softmax_scale, apexmi_stream_t stream);
/* Every variant of the VAE convolution with float out / residual.  in = apexmi_split_bf16x3 of the float activations
 * ([T, H, W, Cin3], Cin3 = 3 x the layer's input channels); w = the packed weight with each tap's channel run repeated
 * three times ([Cout, Kpad], k = tap * Cin3 + ci); bias bf16.  flags: 1 replicate padding | 2 independent frames |
 * 4 read through a nearest 2x upsample.  stride_h / stride_w / pad_top / pad_left / Ho / Wo as apexmi_conv3d_cl_strided
 * (1, 1, -1, -1, 0, 0 = the "same" convolution); stride_t / t_first / To as apexmi_conv3d_cl_tstrided (1, 0, 0 = every
 * frame); act / slope as apexmi_conv3d_cl_act.  Always the 128x128 implicit-GEMM kernel. */
int apexmi_conv3d_cl_f32(const void* in, const void* w, const void* bias, const void* residual, void* out,
                         const void* zeros, int T, int H, int W, int Cin3, int Cout, int Kpad, int kT, int kH, int kW,
                         int flags, int stride_h, int stride_w, int pad_top, int pad_left, int Ho, int Wo, int stride_t,
                         int t_first, int To, int act, float slope, apexmi_stream_t stream);
int apexmi_rmsnorm_cl_f32(const void* x, void* y, const void* gamma, int64_t P, int C, int silu, apexmi_stream_t stream);
int apexmi_groupnorm_cl_f32(const void* x, void* y, const void* gamma, const void* beta, int64_t P, int C, int G, float eps,
                            int silu, void* workspace, size_t workspace_bytes, apexmi_stream_t stream);
int apexmi_time_interleave_cl_f32(const void* x, void* y, int T, int64_t HW, int C, apexmi_stream_t stream);
int apexmi_crossfade_f32(const void* a, void* b, int64_t outer, int E, int64_t inner, int64_t a_so, int64_t a_se,
                         int64_t b_so, int64_t b_se, apexmi_stream_t stream);
int apexmi_frames_to_u8_f32(const void* video, int64_t stride_c, int64_t stride_t, int64_t stride_h, int64_t stride_w,
                            int C, int T, int H, int W, void* out, apexmi_stream_t stream);
/* Text encoders in the f32-storage mode (round 4): the attention of apexmi_attn_fwd_bias with float q / k / v / out (row strides in
 * floats; bias f32 [H, Sq, Sk], keep uint8 [Sk], seg int32 [S], causal, grouped-query key heads; f32 arithmetic, no bf16
 * probabilities), the gated-MLP product and the embedding lookup (bf16 tables, float out, the position sum unrounded). */
int apexmi_attn_fwd_bias_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, float* out,
                             int64_t ldo, int H, int Hkv, int Sq, int Sk, int D, float scale, const float* bias,
                             const uint8_t* keep, const int* seg, int causal, apexmi_stream_t stream);
int apexmi_mul_f32(const float* a, const float* b, float* out, int64_t n, apexmi_stream_t stream);
int apexmi_gather_rows_f32(const void* table, int64_t ldt, int64_t vocab, const int64_t* ids, const void* pos, int64_t ldp,
                           int period, float* out, int64_t ldo, int64_t rows, int C, apexmi_stream_t stream);
/* apexmi_rope_half / apexmi_group_mean_bf16 / apexmi_tanh_clamp / apexmi_pixel_shuffle_clamp on float activations (HunyuanVideo-1.5 VAE encoder and
 * TAEHV in the verification mode; any n > 0). */
int apexmi_rope_half_f32(void* x, int64_t ldx, int64_t rows, int heads, int head_stride, int D, const float* cos_, const float* sin_,
                         apexmi_stream_t stream);
int apexmi_group_mean_f32(const void* x, void* out, int64_t P, int C, int gs, apexmi_stream_t stream);
int apexmi_tanh_clamp_f32(const void* x, void* y, int64_t n, float inv_scale, apexmi_stream_t stream);
int apexmi_pixel_shuffle_clamp_f32(const void* x, void* y, int T, int H, int W, int Cs, int C, int r, int t0, float lo, float hi,
                                   apexmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Built-in kernel timer (HIP events on the launch stream) used by bench.py's roofline leg.
 * classes: 0 gemm, 1 attention, 2 gemv, 3 ln_modulate, 4 qkv_prepare, 5 other.
 * ------------------------------------------------------------------------------------------- */
#define APEXMI_NCLASS 6
int apexmi_prof_enable(int on);
/* Synchronises, then fills ms[c] = summed kernel time, launches[c], flops[c], bytes[c]
 * (algorithmic) per class since the last reset. */
int apexmi_prof_read(double ms[APEXMI_NCLASS], int64_t launches[APEXMI_NCLASS],
                     double flops[APEXMI_NCLASS], double bytes[APEXMI_NCLASS]);
int apexmi_prof_reset(void);

/* Live clock probe (bench.py's `roofline.clock_ghz`): while enabled, every GEMM workgroup adds the shader cycles
 * (`s_memtime`) and the 100 MHz reference ticks (`s_memrealtime`) its K-loop took to two device counters.
 * cycles / ref_ticks x 100 MHz = the effective shader clock while the dominant kernel runs inside the real step —
 * what the DVFS ("power-bound") reading of the roofline fraction rests on.  enable(1) zeroes the counters;
 * read() synchronises the device.  Nothing in the reference corresponds to it (measurement only). */
int apexmi_clk_enable(int on);
int apexmi_clk_read(uint64_t* cycles, uint64_t* ref_ticks);

#ifdef __cplusplus
}
#endif
#endif /* APEXMI_H */
