"""The committed kernel bodies under simpletuner_amd/csrc/gen/ are what tools/kgen/ emits: every generator, run with its default options, reproduces its `.inc` files byte for
byte (the library build needs no generator run, but nobody should have to wonder whether a committed body was edited by hand), and each run passes the generators' own
checks on the way — the `s_waitcnt lgkmcnt` resolution with its loop back-edge assertion and the gfx950 hazard table (tools/kgen/emit.py)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
GEN = ROOT / "simpletuner_amd" / "csrc" / "gen"

CASES = [("fwd64", "FWD64", 128, "attn_fwd64_body.inc"), ("fwd64", "FWD64", 96, "attn_fwd64_hd96_body.inc"),
         ("dq64", "DQ64", 128, "attn_dq64_body.inc"), ("dq64", "DQ64", 96, "attn_dq64_hd96_body.inc"), ("dq64", "DQ64", 64, "attn_dq64_hd64_body.inc"),
         ("dkv", "DKV", 128, "attn_dkv4_body.inc"), ("dkv", "DKV", 96, "attn_dkv4_hd96_body.inc"), ("dkv", "DKV", 64, "attn_dkv4_hd64_body.inc")]


@pytest.mark.parametrize("mod,prefix,hd,name", CASES)
def test_generator_reproduces_the_committed_body(tmp_path, mod, prefix, hd, name):
    out = tmp_path / name
    env = {k: v for k, v in os.environ.items() if not k.startswith(("FWD64_", "DQ64_", "DKV_"))}
    env.update({f"{prefix}_HD": str(hd), f"{prefix}_OUT": str(out)})
    r = subprocess.run([sys.executable, "-m", f"tools.kgen.{mod}"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.read_bytes() == (GEN / name).read_bytes(), f"{name}: the committed body differs from what tools/kgen/{mod}.py emits"
    clob = {"fwd64": "attn_fwd64_clobbers.inc", "dq64": "attn_dq64_clobbers.inc", "dkv": "attn_dkv4_clobbers.inc"}[mod]
    if (tmp_path / clob).exists():          # the generators write the register list next to the body
        assert (tmp_path / clob).read_bytes() == (GEN / clob).read_bytes()
