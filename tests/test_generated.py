"""Generated sources stay in step with their generators (CPU)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_attn_w64_loop_is_what_its_generator_writes(tmp_path):
    """apex-studio_amd/csrc/attn_w64_body.inc (the asm loop of attn_fwd_d128_w64_kernel) is the output of tools/gen_attn_w64.py at
    its default options: an edit of either without the other fails here, before it reaches a GPU."""
    out = tmp_path / "body.inc"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_attn_w64.py"), f"--out={out}"], check=True,
                   capture_output=True)
    committed = open(os.path.join(ROOT, "apex-studio_amd", "csrc", "attn_w64_body.inc")).read()
    assert out.read_text() == committed
    # the shipped first pass: the same generator with max=first (no per-tile running maximum; attn_w64_kernel.h)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_attn_w64.py"), f"--out={out}", "--opt=max=first"], check=True,
                   capture_output=True)
    first = open(os.path.join(ROOT, "apex-studio_amd", "csrc", "attn_w64_first.inc")).read()
    assert out.read_text() == first
    assert "v_max3_f32" in committed and first.count("v_max3_f32") == 28      # only the prologue's tile 0 looks for a maximum
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_attn_w64.py"), f"--out={out}", "--clobbers"], check=True,
                   capture_output=True)
    assert (tmp_path / "attn_w64_clobbers.inc").read_text() == \
        open(os.path.join(ROOT, "apex-studio_amd", "csrc", "attn_w64_clobbers.inc")).read()


def test_attn_w64_clobber_list_covers_the_registers_the_loop_names():
    """Every vN / aN / sN the generated loop names literally is on the asm statement's clobber list or is one of its fixed outputs
    (a[0:127] = O^T): hipcc keeps its own values out of exactly those registers."""
    import re
    csrc = os.path.join(ROOT, "apex-studio_amd", "csrc")
    body = open(os.path.join(csrc, "attn_w64_body.inc")).read() + open(os.path.join(csrc, "attn_w64_first.inc")).read()
    clob = set(re.findall(r'"([vas]\d+)"', open(os.path.join(csrc, "attn_w64_clobbers.inc")).read()))
    used = set()
    for m in re.finditer(r"\b([vas])\[(\d+):(\d+)\]|\b([vas])(\d+)\b", body):
        if m.group(1):
            used |= {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            used.add(f"{m.group(4)}{m.group(5)}")
    outputs = {f"a{i}" for i in range(128)}
    missing = sorted(r for r in used - clob - outputs)
    assert not missing, missing


def test_build_refuses_a_w64_kernel_that_spills():
    """apex-studio_amd/build.py parses hipcc's kernel-resource-usage remarks of attention.hip and refuses a binary whose
    hand-register-mapped kernel uses scratch or spills (a compiler upgrade that allocates differently must fail loudly)."""
    sys.path.insert(0, ROOT)
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import build as b
    import pytest
    ok = """./attn_w64_kernel.h:7:1: remark: Function Name: _ZN12_GLOBAL__N_124attn_fwd_d128_w64_kernelEPKtS1_S1_Ptiiiiiilllf [-Rpass-analysis=kernel-resource-usage]
./attn_w64_kernel.h:7:1: remark:     TotalSGPRs: 96 [-Rpass-analysis=kernel-resource-usage]
./attn_w64_kernel.h:7:1: remark:     VGPRs: 256 [-Rpass-analysis=kernel-resource-usage]
./attn_w64_kernel.h:7:1: remark:     AGPRs: 196 [-Rpass-analysis=kernel-resource-usage]
./attn_w64_kernel.h:7:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]
./attn_w64_kernel.h:7:1: remark:     SGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]
./attn_w64_kernel.h:7:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]
./attn_w64_kernel.h:7:1: remark: Function Name: _ZN12_GLOBAL__N_125attn_fwd_d128_w64r_kernelEPKtS1_S1_Ptiiiiiilllf [-Rpass-analysis=kernel-resource-usage]
./attn_w64_kernel.h:7:1: remark:     VGPRs: 256 [-Rpass-analysis=kernel-resource-usage]
./attn_w64_kernel.h:7:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]
./attn_w64_kernel.h:7:1: remark:     SGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]
./attn_w64_kernel.h:7:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]
attention.hip:1103:1: remark: Function Name: _ZN12_GLOBAL__N_119softmax_rows_kernelEPKflifPtlii [-Rpass-analysis=kernel-resource-usage]
attention.hip:1103:1: remark:     ScratchSize [bytes/lane]: 64 [-Rpass-analysis=kernel-resource-usage]
"""
    res = b.parse_resource_remarks(ok)
    k = next(x for x in res if "w64_kernel" in x)
    assert res[k]["VGPRs"] == 256 and res[k]["AGPRs"] == 196 and res[k]["ScratchSize"] == 0
    b.check_no_spill("attention.hip", ok)                     # another kernel's scratch is not its business
    for field in ("ScratchSize [bytes/lane]: 0", "VGPRs Spill: 0", "SGPRs Spill: 0"):
        with pytest.raises(RuntimeError, match="must not spill"):
            b.check_no_spill("attention.hip", ok.replace(field, field[:-1] + "12", 1))
    with pytest.raises(RuntimeError, match="no resource remark"):
        b.check_no_spill("attention.hip", "")
