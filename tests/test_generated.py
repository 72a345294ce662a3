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


def test_attn_w64_clobber_list_covers_the_registers_the_loop_names():
    """Every vN / aN / sN the generated loop names literally is on the asm statement's clobber list or is one of its fixed outputs
    (a[0:127] = O^T): hipcc keeps its own values out of exactly those registers."""
    import re
    csrc = os.path.join(ROOT, "apex-studio_amd", "csrc")
    body = open(os.path.join(csrc, "attn_w64_body.inc")).read()
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
