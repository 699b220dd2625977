#!/usr/bin/env python3
"""Instruction mix of a kernel's loops from its gfx950 assembly (tools/kasm.sh writes /tmp/kasm.s): for every loop (a label that a later `s_branch` /
`s_cbranch_*` jumps back to) the counts per issue class, and the matrix-pipe cycles those MFMAs need against the single-issue cycles of everything else the SAME
wave issues — a static bound on how busy one wave can keep its SIMD's matrix pipe, to read next to the PMC busy fractions (profiles/*_pmc_*.md).

  usage: tools/kasm.sh attention.hip k_attn_fwd4ILi128ELb0ELb0 && python tools/isa_mix.py [/tmp/kasm.s] [label-regex]
  (label-regex keeps only the loops whose head label matches: the generated bodies jump back from their tails into shared code, which reads as extra "loops";
   `_loop_` keeps their steady-state loop)

The table below holds the back-to-back issue cycles per MFMA on one SIMD (MI355X_MICROARCH.md) for the shapes these kernels use; every other VALU instruction
occupies the SIMD's vector ALU for 4 cycles (64 lanes over 16), transcendental and packed-fp32 ones 8."""
import re
import sys
from collections import Counter

MFMA_CYCLES = {"32x32x16": 32, "16x16x32": 17, "32x32x8": 32, "16x16x16": 17, "32x32x64": 32, "16x16x128": 17}      # back-to-back issue cycles on ONE SIMD (MI355X_MICROARCH.md: 32x32x16 bf16 = 32, 16x16x32 = ~17)


def klass(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("ds_read", "ds_load")):
        return "lds_read"
    if op.startswith(("ds_write", "ds_store")):
        return "lds_write"
    if op.startswith("ds_"):
        return "lds_other"
    if op.startswith(("buffer_load", "global_load", "flat_load", "scratch_load")):
        return "vmem_load" + ("_lds" if "lds" in op else "")
    if op.startswith(("buffer_store", "global_store", "flat_store", "scratch_store", "buffer_atomic", "global_atomic")):
        return "vmem_store"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "valu_trans"
    if op.startswith(("v_accvgpr", "v_mov")):
        return "valu_mov"
    if op.startswith("v_pk_") and op.endswith("_f32"):
        return "valu_packed_f32"          # two passes: measured no cheaper than the two scalar instructions it replaces (profiles/r04_isa_mix_generated_kernels.md)
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main(path, only=None):
    lines = open(path).read().splitlines()
    labels = {}
    ops = []                     # (line index, opcode, text)
    for i, ln in enumerate(lines):
        m = re.match(r"^\s*(\.L[A-Za-z0-9_$]+):", ln)          # compiler blocks (.LBB3_2) and the generated bodies' own labels (.Ldkv_loop_0)
        if m:
            labels[m.group(1)] = i
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        ops.append((i, t.split()[0], t))
    loops = []
    for i, op, t in ops:
        if op.startswith(("s_branch", "s_cbranch")):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] < i:
                loops.append((labels[tgt], i, tgt))
    print(f"{path}: {len(ops)} instructions, {sum(1 for _, o, _ in ops if o.startswith('v_mfma'))} MFMA, {len(loops)} loops")
    for lo, hi, tgt in loops:
        if only and not re.search(only, tgt):
            continue
        body = [(o, t) for i, o, t in ops if lo <= i <= hi]
        c = Counter(klass(o) for o, _ in body)
        if not c.get("mfma"):
            continue
        mf = 0
        for o, t in body:
            if o.startswith("v_mfma"):
                shape = next((k for k in MFMA_CYCLES if k in o), None)
                mf += MFMA_CYCLES.get(shape, 32)
        other = sum(v for k, v in c.items() if k not in ("mfma", "waitcnt", "barrier"))
        # every non-MFMA instruction takes >= 4 issue cycles of the wave's SIMD slot (one 64-lane instruction over a 16-lane SIMD); transcendental ops 8+
        issue = 4 * other + 4 * c.get("valu_trans", 0) + 4 * c.get("valu_packed_f32", 0)
        print(f"\nloop {tgt} (lines {lo + 1}-{hi + 1}): {len(body)} instructions")
        print("  " + ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
        print(f"  matrix-pipe cycles of its MFMAs: {mf}; issue cycles of its other instructions (>= 4 each, 8 transcendental): {issue}")
        valu = 4 * (c.get("valu", 0) + c.get("valu_packed", 0) + c.get("valu_mov", 0)) + 8 * (c.get("valu_trans", 0) + c.get("valu_packed_f32", 0))
        print(f"  vector-ALU cycles of its VALU instructions: {valu}  (ratio to the matrix-pipe cycles: {valu / mf:.2f} — the two pipes of a SIMD run concurrently, so the larger one bounds the loop)")
        print(f"  -> one wave alone keeps the matrix pipe busy at most {mf / (mf + issue):.0%} if nothing overlaps; with k waves per SIMD the bound is min(1, {mf} / max({mf}, {valu})) = {min(1.0, mf / max(mf, valu)):.0%} once the other waves' issue hides this wave's")
        nm = c.get("mfma", 0)
        print(f"  -> one wave per SIMD (the generated 64-row kernels): an MFMA holds the issue slot 4 cycles and the matrix pipe for the figure above, the wave issues its other instructions underneath: "
              f"bound = {mf} / max({mf}, 4 x {nm} + {issue}) = {min(1.0, mf / max(mf, 4 * nm + issue)):.0%}, issue-slot slack {1 - (4 * nm + issue) / mf:+.0%} of the matrix time")
        top = Counter(o for o, _ in body if klass(o).startswith("valu"))
        print("  VALU opcodes: " + ", ".join(f"{k} {v}" for k, v in top.most_common(14)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/kasm.s", sys.argv[2] if len(sys.argv) > 2 else None)
