"""The ctypes mirrors of the C-ABI argument structs (simpletuner_amd/lib.py) against the header itself: include/st355.h is compiled with the host C compiler and
sizeof / offsetof of every field are compared with the ctypes layout — a field added on one side only, or in another order, fails here instead of corrupting
pointers on the GPU box."""
import ctypes as C
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

from simpletuner_amd import lib

ROOT = Path(__file__).resolve().parent.parent
PAIRS = [("st355_gemm_args", lib.GemmArgs), ("st355_qk_rope", lib.QkRope), ("st355_vae_encoder", lib.VaeEncoder),
         ("st355_flux_single_fwd_args", lib.FluxSingleFwdArgs), ("st355_flux_single_bwd_args", lib.FluxSingleBwdArgs),
         ("st355_flux_double_fwd_args", lib.FluxDoubleFwdArgs), ("st355_flux_double_bwd_args", lib.FluxDoubleBwdArgs),
         ("st355_pixart_block_fwd_args", lib.PixartBlockFwdArgs), ("st355_pixart_block_bwd_args", lib.PixartBlockBwdArgs),
         ("st355_sd3_joint_fwd_args", lib.Sd3JointFwdArgs), ("st355_sd3_joint_bwd_args", lib.Sd3JointBwdArgs)]


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no host C compiler")
def test_ctypes_structs_match_the_header(tmp_path):
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "st355.h"', "int main(void) {"]
    for cname, py in PAIRS:
        lines.append(f'  printf("{cname} sizeof %zu\\n", sizeof({cname}));')
        for fname, _ in py._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c11", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)     # a missing / renamed field fails to compile
    got = {}
    for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        c, f, v = ln.split()
        got[(c, f)] = int(v)
    for cname, py in PAIRS:
        assert got[(cname, "sizeof")] == C.sizeof(py), (cname, got[(cname, "sizeof")], C.sizeof(py))
        for fname, _ in py._fields_:
            assert got[(cname, fname)] == getattr(py, fname).offset, (cname, fname, got[(cname, fname)], getattr(py, fname).offset)
    # and the other direction: the header declares no field the mirror lacks (same size + every mirrored field at its offset + equal field counts)
    hdr = (ROOT / "include" / "st355.h").read_text()
    for cname, py in PAIRS:
        body = hdr[hdr.index(f"typedef struct {cname} {{"):hdr.index(f"}} {cname};")]
        assert all(fname in body for fname, _ in py._fields_), cname


def test_block_entry_point_call_sites_name_every_struct_field():
    """the Flux engine fills the block-level argument structs by keyword (ops._fill): every keyword must be a field of the struct (a typo would otherwise go
    unnoticed on a CPU box) and every field but the workspaces the wrapper owns must be given — checked on the source, no GPU needed"""
    import re
    flux = (ROOT / "simpletuner_amd" / "flux" / "transformer.py").read_text()
    pixart = (ROOT / "simpletuner_amd" / "pixart" / "transformer.py").read_text()
    owned = {"gemm_ws", "gemm_ws_bytes", "attn_ws", "skinny_ws", "gA", "gB", "gA_qkv", "gB_qkv", "gA_out", "gB_out"}
    for src, fn, st in ((flux, "block_flux_single_fwd", lib.FluxSingleFwdArgs), (flux, "block_flux_double_fwd", lib.FluxDoubleFwdArgs),
                        (flux, "block_flux_single_bwd", lib.FluxSingleBwdArgs), (flux, "block_flux_double_bwd", lib.FluxDoubleBwdArgs),
                        (pixart, "block_pixart_fwd", lib.PixartBlockFwdArgs), (pixart, "block_pixart_bwd", lib.PixartBlockBwdArgs)):
        i = src.index("ops." + fn + "(")
        depth, j = 0, i
        while True:
            if src[j] == "(":
                depth += 1
            elif src[j] == ")":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        names = set(re.findall(r"[\(,\s]([A-Za-z_][A-Za-z0-9_]*)=", src[i:j])) - {"dtype", "device"}
        fields = {f[0] for f in st._fields_}
        assert names <= fields, (fn, sorted(names - fields))
        assert fields - owned <= names, (fn, sorted(fields - owned - names))
    with pytest.raises(Exception):
        from simpletuner_amd import ops
        ops._fill(lib.FluxSingleFwdArgs(), not_a_field=1)
