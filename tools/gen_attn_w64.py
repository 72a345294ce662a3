#!/usr/bin/env python
"""Generator of the loop of attn_fwd_d128_w64_kernel (apex-studio_amd/csrc/attention.hip): writes
apex-studio_amd/csrc/attn_w64_body.inc, the body of ONE asm statement with a fixed register map.

Why a generator: the kernel runs one wave per SIMD on (nearly) the whole 512-register file — two 32-row query blocks per wave, two
score sets, O^T and Q in AGPRs.  hipcc's allocator does not cope (first builds: 164-219 spilled registers, Q fragments reloaded
from scratch before every MFMA, score sets copied at every join), so the registers are assigned here and every instruction of the
loop is placed here: the MFMAs of a phase with the softmax VALU, the LDS fragment reads and the LDS-DMA pieces of its gaps.

Schedule (per KV tile t of 64 keys, per wave: query blocks A and B of 32 rows):
  phase X(t): S(t+1) = K(t+1) Q^T, 32 MFMAs (each K fragment feeds A and B)   | exp2, row-sum terms, bf16 pairs of tile t
  phase Y(t): O^T += V^T(t) P(t)^T, 32 MFMAs (each V^T fragment feeds A and B) | row max, running-max decision, scaling of tile t+1
  one s_barrier per tile; LDS ring of four stages (tile t+3 is staged during X(t)).
Layouts (LDS images, fragment / score / P lane mapping) are those of attn_fwd_d128_c4_kernel.

Register map
  v[0:63]    score set 0: (block e, key half kt) at 32 e + 16 kt        v[64:127]  score set 1
  v[128:159] P (bf16): block e, 16-key step kk at 128 + 16 e + 4 kk
  v[160:175] K fragment ring (2 x 2 fragments)   v[176:207] V^T fragment ring (2 x 4 fragments)
  v[208:211] row-sum partials (2 per block, carried over the tiles)  v[212:219] K fragment read addresses (one per k-step)
  v220/222 m  v221/223 alpha  v[226:233] temporaries (row-max chains in 228/229/232/233)  v[234:237] V^T fragment read addresses
  v[238:243] LDS-DMA offsets of pieces 1..3 (K, V^T)
LDS: K ring = four 16 KiB stages at 0, V^T ring = four 16 KiB stages at 64 KiB; the loop is unrolled over the four stages, so every
stage offset is an immediate of the ds_read / an immediate of the M0 write
  a[0:127]   O^T: block e, d-tile dt at 64 e + 16 dt               a[128:191] Q fragments: 128 + 32 e + 4 ks
  s[40:63]   scalar temporaries (tile counter, stage offsets, DMA offsets)
"""
import os
import sys

# timing-only ablations (WRONG results; tools/attn_w64_ablate.sh): which parts of the loop are emitted
OPT = {"max": "run", "head": 0, "align": 6, "dma": "piece", "rowsum": "add", "mfma4_pos": "end", "pk_add": False, "pk_fma": False, "adds_in": "Y", "dma_in": "X", "vread_early": 8, "kread_early": 2, "pre_x": 0, "mix_y": 0, "drain": 2, "dummy_x": 0, "dummy_y": 0}   # schedule options (CLI --opt k=v)
TRACE = False   # --trace: per-phase cycle accumulators (s_memtime), written through %[tp] at the end (side library)
ABL = {"fill_x": True, "fill_y": True, "mfma": True, "drain": True, "dma": True, "reads": True, "barrier": True,
       "exp": True, "add": True, "cvt": True, "max": True, "fma": True, "dec": True}

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "apex-studio_amd", "csrc", "attn_w64_body.inc")
NEG_BIG = "0xf149f2ca"      # -1.0e30f
DEFER = "0x40c00000"        # 6.0f


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def ar(base, n=1):
    return f"a{base}" if n == 1 else f"a[{base}:{base + n - 1}]"


def S(st, e, kt):
    return st * 64 + e * 32 + kt * 16


def Sreg(st, v):            # value v = 0..63: block v >> 5, (kt, r) = ((v >> 4) & 1, v & 15)
    return S(st, v >> 5, (v >> 4) & 1) + (v & 15)


def P(e, kk):
    return 128 + e * 16 + kk * 4


def KF(r, kt):
    return 160 + r * 8 + kt * 4


def VF(r, dt):
    return 176 + r * 16 + dt * 4


def PS(e, i):               # two row-sum partials per block, carried over the tiles
    return 208 + e * 2 + i


def KA(ks):                 # K fragment read address of k-step ks (lane part; the stage is an immediate offset)
    return 212 + ks


def VA(kk):                 # V^T fragment read address of 16-key step kk
    return 234 + kk


M_, AL_ = (lambda e: 220 + 2 * e), (lambda e: 221 + 2 * e)     # running max / rescale factor of block e (m on an even register: v_pk_fma_f32)
T = [226 + i for i in range(8)]


def MX(e, kt):             # row-max chains: the raise's temporaries (dead while the chains run)
    return T[4 * e + 2 + kt]


def O(e, dt):
    return e * 64 + dt * 16


def Q(e, ks):
    return 128 + e * 32 + ks * 4


class Emit:
    def __init__(self):
        self.lines = []
        self.lds = []          # destination base registers of outstanding ds_reads, issue order
        self.label_n = 0
        self.in_loop = False
        self.pending_raise = None
        self.pending_rescale = None

    def i(self, s):
        self.lines.append(s)

    def label(self, stem):
        self.label_n += 1
        return f".Lw64_{stem}_{self.label_n}_%="

    def ds_read(self, dst, addr, off):
        if not ABL["reads"]:
            return
        self.i(f"ds_read_b128 {vr(dst, 4)}, {vr(addr)} offset:{off}")
        self.lds.append(dst)

    def need(self, dst):
        """wait until the ds_read into `dst` has returned (LDS returns in order)"""
        if dst in self.lds:
            k = self.lds.index(dst)
            after = len(self.lds) - 1 - k
            self.i(f"s_waitcnt lgkmcnt({after})")
            self.lds = self.lds[k + 1:]

    def lds_flush(self):
        if self.lds:
            self.i("s_waitcnt lgkmcnt(0)")
            self.lds = []


def mfma(em, dst, a, b, c, dst_a=False, b_a=False):
    if not ABL["mfma"]:
        return
    d = ar(dst, 16) if dst_a else vr(dst, 16)
    bb = ar(b, 4) if b_a else vr(b, 4)
    cc = "0" if c is None else d
    em.i(f"v_mfma_f32_32x32x16_bf16 {d}, {vr(a, 4)}, {bb}, {cc}")


def drain(em, n=2):
    if not ABL["drain"]:
        return
    for _ in range(n):
        em.i("s_nop 15")


# ------------------------------------------------------------------------------------------------------------------------------
def max_chain_ops(ns):
    chains = []
    for e in range(2):
        for kt in range(2):
            b = S(ns, e, kt)
            ops = [f"v_max3_f32 {vr(MX(e, kt))}, {vr(b)}, {vr(b + 1)}, {vr(b + 2)}"]
            for j in range(6):
                ops.append(f"v_max3_f32 {vr(MX(e, kt))}, {vr(MX(e, kt))}, {vr(b + 3 + 2 * j)}, {vr(b + 4 + 2 * j)}")
            ops.append(f"v_max_f32 {vr(MX(e, kt))}, {vr(MX(e, kt))}, {vr(b + 15)}")
            chains.append(ops)
    return [c[j] for j in range(8) for c in chains]          # four chains, round robin


def dec_test_ops(e):
    """tile max of block e (scaled) in T[4 e]; lanes whose running max must rise in s[50 + 2 e : 51 + 2 e]"""
    t0, t1 = T[4 * e], T[4 * e + 1]
    return [f"v_max_f32 {vr(MX(e, 0))}, {vr(MX(e, 0))}, {vr(MX(e, 1))}",
            f"v_mov_b32 {vr(t0)}, {vr(MX(e, 0))}",
            "s_nop 1",
            f"v_permlane32_swap_b32 {vr(t0)}, {vr(MX(e, 0))}",
            f"v_max_f32 {vr(t0)}, {vr(t0)}, {vr(MX(e, 0))}",
            f"v_mul_f32 {vr(t0)}, %[sc], {vr(t0)}",                      # m1 = tile max, scaled
            f"v_add_f32 {vr(t1)}, {DEFER}, {vr(M_(e))}",
            f"v_cmp_gt_f32 s[{50 + 2 * e}:{51 + 2 * e}], {vr(t0)}, {vr(t1)}"]   # not vcc: the two blocks' chains interleave


def dec_raise_ops(e):
    """the rows of s[50 + 2 e : ...] take the new integer maximum; the row-sum partials take the factor; s[54 + 2 e : ...] = the
    lanes whose O^T needs it (applied after the phase's MFMAs)"""
    t0, t2, t3 = T[4 * e], T[4 * e + 2], T[4 * e + 3]
    return [f"v_max_f32 {vr(t2)}, {vr(M_(e))}, {vr(t0)}",
            f"v_ceil_f32 {vr(t2)}, {vr(t2)}",
            f"v_cndmask_b32 {vr(t2)}, {vr(M_(e))}, {vr(t2)}, s[{50 + 2 * e}:{51 + 2 * e}]",       # m_new
            f"v_sub_f32 {vr(t3)}, {vr(M_(e))}, {vr(t2)}",
            f"v_mov_b32 {vr(M_(e))}, {vr(t2)}",
            f"v_exp_f32 {vr(AL_(e))}, {vr(t3)}",
            "s_nop 0"] + \
        ([] if OPT["rowsum"] == "mfma4" else      # mfma4: the partial sums live in AGPRs and take the factor with O^T (rescale_block)
         [f"v_mul_f32 {vr(PS(e, 0))}, {vr(PS(e, 0))}, {vr(AL_(e))}",
          f"v_mul_f32 {vr(PS(e, 1))}, {vr(PS(e, 1))}, {vr(AL_(e))}"]) + \
        [f"v_cmp_neq_f32 s[{54 + 2 * e}:{55 + 2 * e}], 1.0, {vr(AL_(e))}"]


def scale_ops(ns):
    q = []
    for e in range(2):
        for w in range(0, 32, 2):
            r = Sreg(ns, e * 32 + w)
            if OPT["pk_fma"]:
                q.append(f"v_pk_fma_f32 {vr(r, 2)}, {vr(r, 2)}, s[78:79], {vr(M_(e), 2)} op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]")
            else:
                q.append(f"v_fma_f32 {vr(r)}, {vr(r)}, %[sc], -{vr(M_(e))}")
                q.append(f"v_fma_f32 {vr(r + 1)}, {vr(r + 1)}, %[sc], -{vr(M_(e))}")
    return q


def sm1_ops(ns, inline_raise):
    """row max of set ns, the running-max test, (the raise,) the scaled differences.  inline_raise: the raise of both blocks in line
    (prologue); otherwise the marker @RAISE stands where the loop branches to its out-of-line raise when any row needs one"""
    q = max_chain_ops(ns)
    d0, d1 = dec_test_ops(0), dec_test_ops(1)
    for j in range(len(d0)):
        q.append(d0[j] + " ;dec")
        q.append(d1[j] + " ;dec")
    if inline_raise:
        r0, r1 = dec_raise_ops(0), dec_raise_ops(1)
        for j in range(len(r0)):
            q.append(r0[j])
            q.append(r1[j])
    else:
        q.append("@RAISE")
    return q + scale_ops(ns)


def add_ops(st):
    """row-sum terms of set st (exponentiated) into the two partial sums of each block.
    rowsum = "add": the unrounded f32 exponentials, one v_add_f32 each (64 per tile).
    rowsum = "dot2c": the PACKED bf16 probabilities of the tile (what multiplies V), two per v_dot2c_f32_bf16 against a {1.0, 1.0}
    bf16 pair in s79 — 32 issue slots per tile instead of 64, and the normaliser is the sum of exactly the weights of P V
    (measured, profiles/r06_attn_w64_rowsum_ab.log: no faster — the dot is not a full-rate VALU op).
    rowsum = "mfma4": no VALU at all — see rowsum_mfma()."""
    q = []
    if OPT["rowsum"] == "mfma4":
        return q
    if OPT["rowsum"] == "dot2c":
        assert not OPT["pk_fma"], "s79 holds the bf16 ones pair"
        if OPT["adds_in"] == "X":      # pair order (block-major): sm2_ops places each behind the v_cvt_pk of its pair
            return [f"v_dot2c_f32_bf16 {vr(PS(e, j & 1))}, s79, {vr(P(e, 0) + j)}" for e in range(2) for j in range(16)]
        for j in range(16):            # four chains round robin
            for e in range(2):
                q.append(f"v_dot2c_f32_bf16 {vr(PS(e, j & 1))}, s79, {vr(P(e, 0) + j)}")
        return q
    for v in range(0, 64, 2):
        e, r = v >> 5, Sreg(st, v)
        if OPT["pk_add"]:
            q.append(f"v_pk_add_f32 {vr(PS(e, 0), 2)}, {vr(PS(e, 0), 2)}, {vr(r, 2)}")
        else:
            q.append(f"v_add_f32 {vr(PS(e, 0))}, {vr(PS(e, 0))}, {vr(r)}")
            q.append(f"v_add_f32 {vr(PS(e, 1))}, {vr(PS(e, 1))}, {vr(r + 1)}")
    return q


def sm2_ops(st, with_adds):
    """exp2 of set st in place and the bf16 pairs (and the row-sum terms); the uses trail the exponentials by a pair"""
    q = []
    adds = add_ops(st)
    per = len(adds) // 32

    def use(v):
        e, w = v >> 5, v & 31
        u = [f"v_cvt_pk_bf16_f32 {vr(P(e, w >> 3) + ((w & 7) >> 1))}, {vr(Sreg(st, v))}, {vr(Sreg(st, v + 1))}"]
        if with_adds:
            mine = adds[(v >> 1) * per:(v >> 1) * per + per]
            u = u + mine if OPT["rowsum"] == "dot2c" else mine + u      # dot2c reads the packed pair, the adds the exponentials
        return u
    for v in range(0, 64, 2):
        q.append(f"v_exp_f32 {vr(Sreg(st, v))}, {vr(Sreg(st, v))}")
        q.append(f"v_exp_f32 {vr(Sreg(st, v + 1))}, {vr(Sreg(st, v + 1))}")
        if v >= 2:
            q += use(v - 2)
    q.append("s_nop 0")
    q += use(62)
    return q


def SACC(e, h):              # rowsum = "mfma4": two 4-register accumulators per block in the free AGPRs
    return 192 + 8 * e + 4 * h


ONES = 224                   # v[224:225]: the bf16 pair {1.0, 1.0} twice — the all-ones A operand of the 4x4x4 MFMA


def rowsum_mfma(em, e, kk, h):
    """rowsum = "mfma4": the row-sum terms on the matrix pipe.  v_mfma_f32_4x4x4_16b_bf16 is sixteen independent 4x4x4 products;
    with an all-ones A every lane's four result registers receive the sum of ITS OWN four bf16 B values: one 2-pass MFMA adds four
    packed probabilities of the lane to its partial row sum — 16 of them per tile (128 matrix-pipe cycles) replace 64 v_add_f32
    (256 issue cycles).  Half h of the 8-key fragment P(e, kk)."""
    if not ABL["add"]:
        return
    d = ar(SACC(e, h), 4)
    em.i(f"v_mfma_f32_4x4x4_16b_bf16 {d}, {vr(ONES, 2)}, {vr(P(e, kk) + 2 * h, 2)}, {d}")


DROP = {"exp": "v_exp_f32 v", "add": ("v_add_f32", "v_pk_add_f32", "v_dot2c_f32_bf16"), "cvt": "v_cvt_pk", "max": "v_max3", "fma": ("v_fma_f32", "v_pk_fma_f32")}


def spread(q, nslots, first_extra=0):
    """split queue q over nslots gaps as evenly as possible; `first_extra` ops go before the first MFMA"""
    for k, pre in DROP.items():
        if not ABL[k]:
            q = [op for op in q if not op.startswith(pre)]   # pre: a prefix or a tuple of prefixes
    if not ABL["dec"]:
        q = [op for op in q if not op.endswith(";dec") and op != "@RAISE"]
    if not q:
        return [], [[] for _ in range(nslots)]
    pre, rest = q[:first_extra], q[first_extra:]
    out = [[] for _ in range(nslots)]
    for k, op in enumerate(rest):
        out[k * nslots // len(rest)].append(op)
    return pre, out


def dma_piece(em, j, stage, part=3):
    """LDS-DMA piece j (0..3 K, 4..7 V^T) of a tile into ring stage `stage` (compile time): s45 / s46 = the tile's K / V^T byte
    offsets, v[238:243] = the per-lane offsets of pieces 1..3.  part 1 = the M0 write, 2 = the load (M0 needs one instruction between
    them: in the loop a gap's fillers stand there), 3 = both with an s_nop"""
    if not ABL["dma"] and em.in_loop:
        return
    if OPT["dma"] == "wave":
        # dma = "wave": a wave's four K (V^T) pieces are CONTIGUOUS in the LDS image (the wave owns image rows 16 wave + 4 i + .. /
        # 32 wave + 8 i + ..), so one M0 write serves four loads through the instruction offset i * 1024 — 2 instead of 8 scalar
        # writes per tile.  The offset also enters the global address: the per-piece lane offsets take it back (prologue), and the
        # shell hands over a K descriptor whose base stands 1024 bytes low (attn_w64_kernel.h, W64_DMA_WAVE).  Same LDS image.
        i = j & 3
        if (part & 1) and i == 0:
            em.i(f"s_add_i32 m0, %[w], {stage * 16384 + (0 if j < 4 else 65536)}")
            if part == 3:
                em.i("s_nop 0")
        if part & 2:
            io = f" offset:{i * 1024}" if i else ""
            if j < 4:
                em.i(f"buffer_load_dwordx4 {'%[vk]' if j == 0 else vr(237 + j)}, %[rk], s45 offen{io} lds")
            else:
                em.i(f"buffer_load_dwordx4 {'%[vv]' if j == 4 else vr(236 + j)}, %[rv], s46 offen{io} lds")
        return
    if part & 1:
        off = stage * 16384 + (j * 4096 if j < 4 else 65536 + (j - 4) * 4096)
        em.i(f"s_add_i32 m0, %[w], {off}")
    if part == 3:
        em.i("s_nop 0")
    if part & 2:
        if j < 4:
            em.i(f"buffer_load_dwordx4 {'%[vk]' if j == 0 else vr(237 + j)}, %[rk], s45 offen lds")
        else:
            em.i(f"buffer_load_dwordx4 {'%[vv]' if j == 4 else vr(236 + j)}, %[rv], s46 offen lds")


def k_first_reads(em, stage):
    """first-step K fragments of the tile in K stage `stage` (the tile phase X computes next): issued a phase early"""
    em.ds_read(KF(0, 0), KA(0), stage * 16384)
    em.ds_read(KF(0, 1), KA(0), stage * 16384 + 8192)


def mask_block(em, st):
    """keys >= Sk of the last tile -> -1e30 (set st); %[rem] = Sk % 64 != 0.  T[0] = 8 hi, T[1] = -1e30"""
    em.i(f"v_mbcnt_lo_u32_b32 {vr(T[0])}, -1, 0")
    em.i(f"v_mbcnt_hi_u32_b32 {vr(T[0])}, -1, {vr(T[0])}")
    em.i(f"v_lshrrev_b32 {vr(T[0])}, 2, {vr(T[0])}")
    em.i(f"v_and_b32 {vr(T[0])}, 8, {vr(T[0])}")
    em.i(f"v_mov_b32 {vr(T[1])}, {NEG_BIG}")
    for e in range(2):
        for kt in range(2):
            for r in range(16):
                a, bb = r >> 2, r & 3
                kc = kt * 32 + 16 * (a >> 1) + 4 * (a & 1) + bb
                em.i(f"s_sub_i32 s47, %[rem], {kc}")
                em.i(f"v_cmp_le_i32 vcc, s47, {vr(T[0])}")            # rem - kc <= 8 hi  <=>  key >= rem
                reg = S(st, e, kt) + r
                em.i(f"v_cndmask_b32 {vr(reg)}, {vr(reg)}, {vr(T[1])}, vcc")


def rescale_test(em):
    """hot path: one scalar test; the rescale of O^T itself (rare) is out of line.  Returns (entry, return) labels"""
    go, back = em.label("rescale"), em.label("rescaled")
    em.i("s_cmp_lg_u64 s[58:59], 0")                 # some row of either block was raised in this tile
    em.i(f"s_cbranch_scc1 {go}")
    em.i(f"{back}:")
    return go, back


def rescale_block(em, go, back):
    """O^T *= alpha for a block whose running max rose: through VGPR temporaries.  The raise left `alpha != 1` lane masks in
    s[54:55] (block A) and s[56:57] (block B)"""
    em.i(f"{go}:")
    for e in range(2):
        skip = em.label("nors")
        em.i(f"s_cmp_eq_u64 s[{54 + 2 * e}:{55 + 2 * e}], 0")
        em.i(f"s_cbranch_scc1 {skip}")
        drain(em, 4)
        for k in range(0, 64, 4):
            for u in range(4):
                em.i(f"v_accvgpr_read_b32 {vr(T[u])}, {ar(e * 64 + k + u)}")
            em.i("s_nop 0")
            for u in range(4):
                em.i(f"v_mul_f32 {vr(T[u])}, {vr(T[u])}, {vr(AL_(e))}")
            em.i("s_nop 0")
            for u in range(4):
                em.i(f"v_accvgpr_write_b32 {ar(e * 64 + k + u)}, {vr(T[u])}")
        if OPT["rowsum"] == "mfma4":      # the block's partial row sums (register 0 of each accumulator is the one read at the end)
            for h in range(2):
                em.i(f"v_accvgpr_read_b32 {vr(T[h])}, {ar(SACC(e, h))}")
            em.i("s_nop 0")
            for h in range(2):
                em.i(f"v_mul_f32 {vr(T[h])}, {vr(T[h])}, {vr(AL_(e))}")
            em.i("s_nop 0")
            for h in range(2):
                em.i(f"v_accvgpr_write_b32 {ar(SACC(e, h))}, {vr(T[h])}")
        em.i("s_nop 7")
        em.i(f"{skip}:")
    em.i(f"s_branch {back}")


def tile(em, sg, more, more2, dma):
    """tile t = s40 with t % 4 = sg (compile time: ring stages and the score set follow from it).  more: tile t+1 exists (phase X
    computes its scores; its first K fragments are already in flight); more2: tile t+2 exists (its first K fragments are read at the
    end of phase Y); dma: tile t+3 exists (staged in phase Y)"""
    st, ns = sg & 1, (sg & 1) ^ 1
    kst, vst, k2st, dst = (sg + 1) & 3, sg, (sg + 2) & 3, (sg + 3) & 3      # K stage of t+1, V^T stage of t, K of t+2, DMA of t+3
    # ---------------- phase X: S(t+1) = K(t+1) Q^T beside exp2 / bf16 pairs (/ row-sum terms) of tile t ----------------
    em.in_loop = True
    em.lds = [KF(0, 0), KF(0, 1)] if (more and ABL["reads"]) else []
    if TRACE:
        em.i("s_memtime s[80:81]")
    q = sm2_ops(st, with_adds=OPT["adds_in"] == "X") if ABL["fill_x"] else []
    pre, gaps = spread(q, 32, first_extra=OPT["pre_x"])
    for k in range(OPT["dummy_x"]):   # experiment: independent VALU ops in the gaps
        gaps[k * 32 // OPT["dummy_x"]].append(f"v_mov_b32 v{226 + (k & 3)}, 1.0")
    for op in pre:
        em.i(op)
    vread_slot = 32 - OPT["vread_early"]
    for ks in range(8):
        if more and ks + 1 < 8:
            em.ds_read(KF((ks + 1) & 1, 0), KA(ks + 1), kst * 16384)
            em.ds_read(KF((ks + 1) & 1, 1), KA(ks + 1), kst * 16384 + 8192)
        for qq in range(4):
            e, kt, slot = qq >> 1, qq & 1, ks * 4 + qq
            if more:
                if qq == 0:
                    em.need(KF(ks & 1, 1))               # one wait per step: the later of its two fragments
                mfma(em, S(ns, e, kt), KF(ks & 1, kt), Q(e, ks), None if ks == 0 else 1, b_a=True)
            for op in gaps[slot]:
                em.i(op)
            if slot == vread_slot - 1:
                # V^T fragments of the first step of phase Y: their latency rides under the tail of phase X
                for dt in range(4):
                    em.ds_read(VF(0, dt), VA(0), vst * 16384 + dt * 4096)
    if TRACE:
        em.i("s_memtime s[82:83]")
    if more and not more2:
        # tile t + 1 is the last one (this body only): mask its keys past Sk if it is ragged.  No drain anywhere else: the first VALU
        # reads of S(t+1) (row max) follow the row-sum terms of tile t, ten MFMA gaps into phase Y
        lm = em.label("nomask")
        em.i("s_cmp_eq_u32 %[rem], 0")
        em.i(f"s_cbranch_scc1 {lm}")
        drain(em, 3)
        mask_block(em, ns)
        em.i(f"{lm}:")
    # ---------------- phase Y: O^T += V^T(t) P(t)^T beside (row-sum terms of tile t,) max / decision / scaling of tile t+1 ------
    q = []
    if ABL["fill_y"]:
        adds = add_ops(st) if OPT["adds_in"] == "Y" else []
        if OPT["max"] == "first":    # the running maximum stays what tile 0 made it (the kernel's shell checks the row sums at the end)
            sm1 = scale_ops(ns) if more else []
        else:
            sm1 = sm1_ops(ns, inline_raise=False) if more else []
        q += adds + sm1
    _, gaps = spread(q, 32)
    raise_lbl, raise_ret = em.label("raise"), em.label("raised")
    for k in range(OPT["dummy_y"]):
        gaps[k * 32 // OPT["dummy_y"]].append(f"v_mov_b32 v{226 + (k & 3)}, 1.0")
    for kk in range(4):
        if kk == 2:
            # ---- the tile's one barrier, in the MIDDLE of the phase: tile t + 2 (staged a tile ago) becomes visible, and every
            # wave has left phase Y(t - 1), whose V^T stage the LDS-DMA of tile t + 3 (below) overwrites.  The MFMAs of steps 0 / 1
            # are still in the pipe while the waves meet. ----
            if TRACE:
                em.i("s_memtime s[84:85]")
            em.i("s_waitcnt vmcnt(0)")
            if ABL["barrier"]:
                em.i("s_barrier")
            if TRACE:
                em.i("s_memtime s[86:87]")
        if kk + 1 < 4:
            for dt in range(4):
                em.ds_read(VF((kk + 1) & 1, dt), VA(kk + 1), vst * 16384 + dt * 4096)
        for qq in range(8):
            dt, e, slot = qq >> 1, qq & 1, kk * 8 + qq
            if qq == 0:
                em.need(VF(kk & 1, 3))                   # one wait per step
            mfma(em, O(e, dt), VF(kk & 1, dt), P(e, kk), 1, dst_a=True)
            if OPT["rowsum"] == "mfma4" and OPT["mfma4_pos"] == "start" and (qq & 1) == 1:
                rowsum_mfma(em, qq >> 2, kk, (qq >> 1) & 1)
            piece = dma and slot >= 16 and (slot & 1) == 1
            if piece:
                dma_piece(em, (slot - 16) >> 1, dst, part=1)
                if not gaps[slot]:
                    em.i("s_nop 0")
            for op in gaps[slot]:
                if op == "@RAISE":       # any row of either block above its running max + DEFER?  (rare after the first tiles)
                    em.i("s_or_b64 s[58:59], s[50:51], s[52:53]")
                    em.i("s_mov_b64 s[54:55], 0")
                    em.i("s_mov_b64 s[56:57], 0")
                    em.i("s_cmp_lg_u64 s[58:59], 0")
                    em.i(f"s_cbranch_scc1 {raise_lbl}")
                    em.i(f"{raise_ret}:")
                else:
                    em.i(op)
            if piece:
                dma_piece(em, (slot - 16) >> 1, dst, part=2)
            if OPT["rowsum"] == "mfma4" and OPT["mfma4_pos"] == "end" and (qq & 1) == 1:      # behind the gap's VALU, right in front of the next P V MFMA
                rowsum_mfma(em, qq >> 2, kk, (qq >> 1) & 1)
            if more2 and slot == 31 - OPT["kread_early"]:
                k_first_reads(em, k2st)                  # first K fragments of tile t + 2, for phase X of the next tile
    em.pending_rescale = rescale_test(em) if (more and OPT["max"] != "first") else None
    em.i("s_add_i32 s40, s40, 1")
    em.i("s_add_i32 s45, s45, 16384")                    # K / V^T byte offsets of the tile the NEXT body stages
    em.i("s_add_i32 s46, s46, 128")
    em.pending_raise = (raise_lbl, raise_ret) if (more and ABL["fill_y"] and ABL["dec"] and OPT["max"] != "first") else None
    if TRACE:   # s[64:65] += phase X, s[66:67] += phase Y (both halves), s[68:69] += DMA wait + barrier
        em.i("s_memtime s[88:89]")
        em.i("s_waitcnt lgkmcnt(0)")                     # (drains the early K reads too: the next body's counted waits stay valid)
        for acc, (hi_, lo_) in ((64, (82, 80)), (66, (84, 82)), (68, (86, 84)), (66, (88, 86))):
            em.i(f"s_sub_u32 s76, s{hi_}, s{lo_}")
            em.i(f"s_subb_u32 s77, s{hi_ + 1}, s{lo_ + 1}")
            em.i(f"s_add_u32 s{acc}, s{acc}, s76")
            em.i(f"s_addc_u32 s{acc + 1}, s{acc + 1}, s77")
        em.i("s_sub_u32 s76, s80, s74")                  # tile edge: from the previous tile's last stamp to this tile's first
        em.i("s_subb_u32 s77, s81, s75")
        em.i("s_add_u32 s70, s70, s76")
        em.i("s_addc_u32 s71, s71, s77")
        em.i("s_mov_b64 s[74:75], s[88:89]")
    # the scoreboard at the tile edge: the next body assumes exactly the early K reads (or nothing)
    want = [KF(0, 0), KF(0, 1)] if (more2 and ABL["reads"]) else []
    assert em.lds == want, (em.lds, want)


def dispatch(em, sg, labels, done):
    """choose the body of tile s40 (s40 % 4 = sg): f = tiles t+1..t+3 exist, m2 = t+1, t+2, m1 = t+1 only, l = last"""
    em.i(f"{labels[('top', sg)]}:")
    em.i("s_cmp_ge_u32 s40, %[nt]")
    em.i(f"s_cbranch_scc1 {done}")
    for add, k in ((3, "f"), (2, "m2"), (1, "m1")):
        em.i(f"s_add_i32 s47, s40, {add}")
        em.i("s_cmp_lt_u32 s47, %[nt]")
        em.i(f"s_cbranch_scc1 {labels[(k, sg)]}")
    em.i(f"s_branch {labels[('l', sg)]}")


def main():
    out = OUT
    for a in sys.argv[1:]:
        if a.startswith("--out="):
            out = a[6:]
        elif a.startswith("--no-"):
            ABL[a[5:]] = False
        elif a.startswith("--opt="):
            k, v = a[6:].split("=")
            OPT[k] = type(OPT[k])(int(v)) if isinstance(OPT[k], (bool, int)) else v
        elif a == "--trace":
            global TRACE
            TRACE = True
    em = Emit()
    # ---------------- prologue ----------------
    for e, a in ((0, "%[qa]"), (1, "%[qb]")):
        for ks in range(8):
            em.i(f"global_load_dwordx4 {ar(Q(e, ks), 4)}, {a}, off offset:{ks * 32}")
    for e in range(2):
        em.i(f"v_mov_b32 {vr(M_(e))}, {NEG_BIG}")
        em.i(f"v_mov_b32 {vr(AL_(e))}, 1.0")
        for i in range(2):
            em.i(f"v_mov_b32 {vr(PS(e, i))}, 0")
    for ks in range(8):                               # fragment read addresses: lane part x k-step, the ring stage is an immediate
        em.i(f"v_xor_b32 {vr(KA(ks))}, {ks << 5}, %[ka]")
    for kk in range(4):
        em.i(f"v_xor_b32 {vr(VA(kk))}, {kk << 5}, %[va]")
        em.i(f"v_add_u32 {vr(VA(kk))}, 65536, {vr(VA(kk))}")     # the V^T ring starts at 64 KiB
    if OPT["dma"] == "wave":
        # K piece i: image rows 16 wave + 4 i + (lane >> 4) <- keys 16 wave + 4 swap2(i) + (lane >> 4) (perm32 swaps bits 2, 3 of the
        # row), chunk ^= 4 i; minus the instruction offset i * 1024.  %[vk] = piece 0's offset + 1024 (the descriptor's base is 1024 low)
        em.i(f"v_xor_b32 {vr(238)}, 64, %[vk]")
        em.i(f"v_add_u32 {vr(238)}, 1024, {vr(238)}")             # swap2(1) - 1 = +1
        em.i(f"v_xor_b32 {vr(239)}, 128, %[vk]")
        em.i(f"v_subrev_u32 {vr(239)}, 1024, {vr(239)}")          # swap2(2) - 2 = -1
        em.i(f"v_xor_b32 {vr(240)}, 192, %[vk]")                  # swap2(3) - 3 = 0
        # V^T piece i: image rows (d) 32 wave + 8 i + (lane >> 3), chunk ^= 4 (i & 1); %[vp] = 8 rows of V^T - 1024 bytes
        em.i(f"v_xor_b32 {vr(241)}, 64, %[vv]")
        em.i(f"v_add_u32 {vr(241)}, %[vp], {vr(241)}")
        em.i(f"v_add_u32 {vr(242)}, %[vp], %[vv]")
        em.i(f"v_add_u32 {vr(242)}, %[vp], {vr(242)}")
        em.i(f"v_add_u32 {vr(243)}, %[vp], {vr(241)}")
        em.i(f"v_add_u32 {vr(243)}, %[vp], {vr(243)}")
    else:
        for j in range(1, 4):                         # per-piece LDS-DMA offsets: no scalar offset arithmetic in the loop
            em.i(f"v_add_u32 {vr(237 + j)}, {j * 4096}, %[vk]")
        em.i(f"v_add_u32 {vr(241)}, %[vp], %[vv]")
        em.i(f"v_add_u32 {vr(242)}, %[vp], {vr(241)}")
        em.i(f"v_add_u32 {vr(243)}, %[vp], {vr(242)}")
    em.i("s_mov_b32 s78, %[sc]")                      # {scale, scale} for v_pk_fma_f32
    em.i("s_mov_b32 s79, 0x3f803f80" if OPT["rowsum"] == "dot2c" else "s_mov_b32 s79, %[sc]")     # dot2c: the bf16 pair {1.0, 1.0}
    for e in range(2):
        em.i(f"s_mov_b64 s[{54 + 2 * e}:{55 + 2 * e}], 0")
    for k in range(128):
        em.i(f"v_accvgpr_write_b32 {ar(k)}, 0")
    if OPT["rowsum"] == "mfma4":
        for k in range(192, 208):
            em.i(f"v_accvgpr_write_b32 {ar(k)}, 0")
        em.i(f"v_mov_b32 {vr(ONES)}, 0x3f803f80")
        em.i(f"v_mov_b32 {vr(ONES + 1)}, 0x3f803f80")
    # tiles 0..2 staged
    for tt in range(3):
        skip = em.label("nost")
        em.i(f"s_cmp_le_u32 %[nt], {tt}")
        em.i(f"s_cbranch_scc1 {skip}")
        em.i(f"s_mov_b32 s45, {tt << 14}")
        em.i(f"s_mov_b32 s46, {tt << 7}")
        for j in range(8):
            dma_piece(em, j, tt)
        em.i(f"{skip}:")
    l2, l1, lw = em.label("nt2"), em.label("nt1"), em.label("waited")
    em.i("s_cmp_lt_u32 %[nt], 3")
    em.i(f"s_cbranch_scc1 {l2}")
    em.i("s_waitcnt vmcnt(16)")
    em.i(f"s_branch {lw}")
    em.i(f"{l2}:")
    em.i("s_cmp_lt_u32 %[nt], 2")
    em.i(f"s_cbranch_scc1 {l1}")
    em.i("s_waitcnt vmcnt(8)")
    em.i(f"s_branch {lw}")
    em.i(f"{l1}:")
    em.i("s_waitcnt vmcnt(0)")
    em.i(f"{lw}:")
    em.i("s_barrier")
    # S(0) into set 0, nothing beside it
    for ks in range(8):
        em.ds_read(KF(0, 0), KA(ks), 0)
        em.ds_read(KF(0, 1), KA(ks), 8192)
        for e in range(2):
            for kt in range(2):
                em.need(KF(0, kt))
                mfma(em, S(0, e, kt), KF(0, kt), Q(e, ks), None if ks == 0 else 1, b_a=True)
        em.i("s_nop 7")                               # the fragments are overwritten by the next step's reads
    drain(em)
    lm = em.label("nomask0")
    em.i("s_cmp_lg_u32 %[nt], 1")
    em.i(f"s_cbranch_scc1 {lm}")
    em.i("s_cmp_eq_u32 %[rem], 0")
    em.i(f"s_cbranch_scc1 {lm}")
    mask_block(em, 0)
    em.i(f"{lm}:")
    if OPT["max"] == "first":
        # max=first: the row keeps ceil(tile 0's scaled maximum) + head for the whole loop.  The head room costs no precision (it
        # shifts every probability, the row sum and the numerator by the same power of two) and moves the window the f32 range
        # leaves: a later score may stand 60 + head binades above tile 0's best before the row sum passes the shell's 2^60 test,
        # and everything down to -(126 - 24 - head) binades below it keeps all its bits
        pre = [op for op in sm1_ops(0, inline_raise=True)]
        nscale = len(scale_ops(0))
        for op in pre[:-nscale]:
            em.i(op.replace(' ;dec', ''))
        if OPT["head"]:
            import struct
            lit = "0x%08x" % struct.unpack("<I", struct.pack("<f", float(OPT["head"])))[0]
            for e in range(2):
                em.i(f"v_add_f32 {vr(M_(e))}, {lit}, {vr(M_(e))}")
        for op in pre[-nscale:]:
            em.i(op)
    else:
        for op in sm1_ops(0, inline_raise=True):
            em.i(op.replace(' ;dec', ''))
    for e in range(2):
        em.i(f"v_mov_b32 {vr(AL_(e))}, 1.0")          # O is still zero: nothing to rescale
    l8, lw = em.label("w0"), em.label("waited2")
    em.i("s_cmp_lt_u32 %[nt], 3")
    em.i(f"s_cbranch_scc1 {l8}")
    em.i("s_waitcnt vmcnt(8)")
    em.i(f"s_branch {lw}")
    em.i(f"{l8}:")
    em.i("s_waitcnt vmcnt(0)")
    em.i(f"{lw}:")
    em.i("s_barrier")
    em.i("s_mov_b32 s40, 0")
    em.i(f"s_mov_b32 s45, {3 << 14}")                 # byte offsets of tile 3, the first one the loop stages
    em.i(f"s_mov_b32 s46, {3 << 7}")
    lk = em.label("nok1")
    em.i("s_cmp_lt_u32 %[nt], 2")
    em.i(f"s_cbranch_scc1 {lk}")
    k_first_reads(em, 1)                              # first K fragments of tile 1: phase X of tile 0 expects them in flight
    em.i(f"{lk}:")
    em.lds = []
    if TRACE:
        for k in range(64, 72, 2):
            em.i(f"s_mov_b64 s[{k}:{k + 1}], 0")
        em.i("s_memtime s[74:75]")
        em.i("s_waitcnt lgkmcnt(0)")
    # ---------------- tiles ----------------
    labels = {}
    for par in range(4):
        for k in ("top", "f", "m2", "m1", "l"):
            labels[(k, par)] = em.label(f"{k}{par}")
    done = em.label("done")
    rare = []                            # (raise labels, rescale labels) of every body: emitted behind the bodies

    def body(k, sg, more, more2, dma):
        if OPT["align"] and k == "f" and sg == 0:
            em.i(f".p2align {OPT['align']}")        # the steady-state loop starts on a 2^align-byte boundary
        em.i(f"{labels[(k, sg)]}:")
        tile(em, sg, more, more2, dma)
        rare.append((em.pending_raise, em.pending_rescale))

    em.i(f"s_branch {labels[('top', 0)]}")
    # steady state: the four full bodies one behind the other, falling through while three more tiles exist
    for sg in range(4):
        body("f", sg, True, True, True)
        em.i("s_add_i32 s47, s40, 3")
        em.i("s_cmp_lt_u32 s47, %[nt]")
        if sg < 3:
            em.i(f"s_cbranch_scc0 {labels[('top', sg + 1)]}")
        else:
            em.i(f"s_cbranch_scc1 {labels[('f', 0)]}")
            em.i(f"s_branch {labels[('top', 0)]}")
    # the dispatchers and the tail bodies
    for sg in range(4):
        dispatch(em, sg, labels, done)
        for k, more, more2, dma in (("m2", True, True, False), ("m1", True, False, False), ("l", False, False, False)):
            body(k, sg, more, more2, dma)
            em.i(f"s_branch {labels[('top', (sg + 1) & 3)]}")
    # out of line: the raise of the running maxima (both blocks; lanes that need none keep theirs) and the rescale of O^T
    for pr, ps in rare:
        if pr:
            em.i(f"{pr[0]}:")
            r0, r1 = dec_raise_ops(0), dec_raise_ops(1)
            for j in range(len(r0)):
                em.i(r0[j])
                em.i(r1[j])
            em.i(f"s_branch {pr[1]}")
        if ps:
            rescale_block(em, ps[0], ps[1])
    em.i(f"s_branch {done}")
    em.i(f"{done}:")
    drain(em)
    if TRACE:   # lane 0 of every wave: {phase X, phase Y, DMA wait, barrier} cycles summed over the tiles
        skip = em.label("notrace")
        em.i("s_cmp_eq_u64 %[tp], 0")
        em.i(f"s_cbranch_scc1 {skip}")
        em.i(f"v_mbcnt_lo_u32_b32 {vr(T[0])}, -1, 0")
        em.i(f"v_mbcnt_hi_u32_b32 {vr(T[0])}, -1, {vr(T[0])}")
        em.i(f"v_cmp_eq_u32 vcc, 0, {vr(T[0])}")
        em.i("s_and_saveexec_b64 s[76:77], vcc")
        em.i(f"v_mov_b32 {vr(T[2])}, 0")
        for k in range(4):
            em.i(f"v_mov_b32 {vr(T[0])}, s{64 + 2 * k}")
            em.i(f"v_mov_b32 {vr(T[1])}, s{65 + 2 * k}")
            em.i(f"global_store_dwordx2 {vr(T[2])}, {vr(T[0], 2)}, %[tp] offset:{8 * k}")
            em.i("s_nop 1")
        em.i("s_waitcnt vmcnt(0)")
        em.i("s_mov_b64 exec, s[76:77]")
        em.i(f"{skip}:")
    if OPT["rowsum"] == "mfma4":
        for e in range(2):
            for h in range(2):
                em.i(f"v_accvgpr_read_b32 {vr(PS(e, h))}, {ar(SACC(e, h))}")
        em.i("s_nop 1")
    for e, o in ((0, "%[la]"), (1, "%[lb]")):
        em.i(f"v_add_f32 {o}, {vr(PS(e, 0))}, {vr(PS(e, 1))}")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_attn_w64.py - do not edit (the generator holds the register map and the schedule)\n")
        for ln in em.lines:
            f.write(f'"{ln}\\n\\t"\n')
    if out == OUT or "--clobbers" in sys.argv:      # the asm statement's clobber list: the registers this generator's map owns
        regs = [f"v{i}" for i in range(244)] + [f"a{i}" for i in range(128, 208)] + [f"s{i}" for i in range(40, 90)]
        with open(os.path.join(os.path.dirname(out), "attn_w64_clobbers.inc"), "w") as f:
            f.write("// GENERATED by tools/gen_attn_w64.py with attn_w64_body.inc: the registers the asm body owns (v[0:243], a[128:207] — Q and\n"
                    "// the rowsum accumulators; a[0:127] = O^T are the statement's outputs —, s[40:89])\n")
            for k in range(0, len(regs), 16):
                f.write(", ".join(f'"{r}"' for r in regs[k:k + 16]) + (",\n" if k + 16 < len(regs) else "\n"))
    n_mfma = sum("v_mfma" in ln for ln in em.lines)
    print(f"{out}: {len(em.lines)} instructions, {n_mfma} MFMAs")


if __name__ == "__main__":
    main()
