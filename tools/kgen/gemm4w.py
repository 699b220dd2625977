"""kgen.gemm4w — LAB generator: the main loop of a 256 x 256 x 64 bf16 NT GEMM tile on FOUR waves (one per SIMD, 128 x 128 outputs per wave, the whole
accumulator tile in 256 AGPRs), emitted with the same instruction-stream model as the attention bodies (tools/kgen/emit.py).

Why it exists: the shipped GEMM (csrc/gemm.hip k_gemm_pq / k_gemm_pz: 8 waves as two ping-pong groups, 128 x 64 outputs per wave, hipcc-scheduled) keeps the
matrix pipe 68-74 % busy (profiles/archive/r03_gemm_pmc_mfma_lds.md).  The attention kernels went from 0.47 to 0.61-0.73 busy on a one-wave-per-SIMD structure whose
fragments feed more MFMAs each; the same structure for the GEMM reads 8 LDS fragments per 16 MFMAs (the shipped kernel: 6 per 8) and has no second wave group to
synchronise with.  This generator + tools/gemm_kg_lab.hip MEASURE that structure's main loop before anybody rebuilds the product kernel's seven epilogues around it
(DESIGN.md §7).  Not part of libst355.

  C[m, n] = sum_k A[m, k] B[n, k]      A [M, K] (token rows), B [N, K] (nn.Linear weight rows), bf16, K-contiguous; M, N % 256 == 0, K % 64 == 0, K >= 128
  workgroup = one 256 x 256 tile, wave w = quadrant (wm, wn) = (w >> 1, w & 1)
  LDS: two stages of [A tile 256 rows x 128 B | B tile 256 rows x 128 B] (64 KiB each); row r keeps its 16-byte chunk c at slot c ^ ((r >> 1) & 7): ds_read_b128
  fragment reads (32 rows x 2 chunks per instruction) touch every bank once per 16 lanes
  per k-tile and wave: 16 LDS-DMA pieces (1 KiB each: 8 rows), 32 ds_read_b128, 64 MFMAs, ONE barrier
  pipeline: k-step ks of tile t runs its 16 MFMAs from fragment buffer ks & 1 while the 8 fragments of k-step ks + 1 arrive in the other one; at k-step 3 the
  wave waits for its DMA pieces of tile t + 1, meets the barrier (every wave is past its last read of tile t's stage, every piece of tile t + 1 has landed),
  reads k-step 0 of tile t + 1 and stages tile t + 2 into the stage tile t just left
"""
from __future__ import annotations

import os
import sys

from .emit import Stream, ar, check_hazards, resolve_lgkm, vr, weave_budget

CAP = float(os.environ.get("GEMM4W_CAP", "4"))
DBG = set(filter(None, os.environ.get("GEMM4W_DBG", "").split(",")))      # "nodma": no staging inside the loop; "noread": no fragment reads inside the loop (timing probes: wrong results)

S_M0, S_KA, S_KB, S_T0, S_CNT, S_REM, S_INC, S_DEC = 40, 42, 44, 46, 47, 48, 49, 50
VOA, VOB = list(range(40, 48)), list(range(48, 56))
XA, WA = list(range(56, 60)), list(range(60, 64))
XF = lambda buf, mb: vr(64 + buf * 16 + mb * 4, 4)
WF = lambda buf, nb: vr(96 + buf * 16 + nb * 4, 4)
ACC = lambda nb, mb: ar((nb * 4 + mb) * 16, 16)
T = list(range(128, 144))          # epilogue temporaries
CP = [144, 146, 148, 150]          # C row pointers of the four 32-token blocks


def dma_pieces() -> list[list[str]]:
    pcs = []
    for p in range(8):
        pcs.append([f"s_add_u32 m0, s{S_T0}, {p * 1024}", "s_nop 0", f"global_load_lds_dwordx4 {vr(VOA[p])}, s[{S_KA}:{S_KA + 1}]"])
    for p in range(8):
        pcs.append([f"s_add_u32 m0, s{S_T0}, {32768 + p * 1024}", "s_nop 0", f"global_load_lds_dwordx4 {vr(VOB[p])}, s[{S_KB}:{S_KB + 1}]"])
    return pcs


def frag_reads(ks: int, buf: int) -> list[str]:
    out = []
    for b in range(4):
        out.append(f"ds_read_b128 {XF(buf, b)}, {vr(XA[ks])} offset:{b * 4096} ;@ld:f{ks}")
        out.append(f"ds_read_b128 {WF(buf, b)}, {vr(WA[ks])} offset:{b * 4096} ;@ld:f{ks}")
    return out


def build() -> str:
    st = Stream()
    o = st.op
    st.comment("---- prologue")
    o(f"s_mov_b32 s{S_M0}, m0")
    for p in range(8):
        src_a, src_b = ("%[voae]", "%[vobe]") if p % 2 == 0 else ("%[voao]", "%[vobo]")
        if p < 2:
            o(f"v_mov_b32_e32 {vr(VOA[p])}, {src_a}")
            o(f"v_mov_b32_e32 {vr(VOB[p])}, {src_b}")
        else:
            o(f"v_add_u32_e32 {vr(VOA[p])}, %[stepa], {vr(VOA[p - 2])}")
            o(f"v_add_u32_e32 {vr(VOB[p])}, %[stepb], {vr(VOB[p - 2])}")
    for ks in range(4):
        o(f"v_xor_b32_e32 {vr(XA[ks])}, {ks * 32}, %[fa]")
        o(f"v_xor_b32_e32 {vr(WA[ks])}, {ks * 32}, %[fb]")
    o(f"s_mov_b64 s[{S_KA}:{S_KA + 1}], %[abase]")
    o(f"s_mov_b64 s[{S_KB}:{S_KB + 1}], %[bbase]")
    for i in range(256):
        o(f"v_accvgpr_write_b32 a{i}, 0")
    o(f"s_add_u32 s{S_T0}, %[lds], %[wv8k]")
    for pc in dma_pieces():
        st.extend(pc)
    for base in (S_KA, S_KB):
        o(f"s_add_u32 s{base}, s{base}, 128")
        o(f"s_addc_u32 s{base + 1}, s{base + 1}, 0")
    o(f"s_xor_b32 s{S_T0}, s{S_T0}, 0x10000")
    for pc in dma_pieces():
        st.extend(pc)
    o(f"s_xor_b32 s{S_T0}, s{S_T0}, 0x10000")
    o("s_waitcnt vmcnt(0)")
    o("s_barrier")
    st.extend(frag_reads(0, 0))
    o(f"s_sub_u32 s{S_REM}, %[nkt], 2")
    o(f"s_mov_b32 s{S_CNT}, %[nkt]")
    st.comment("---- main loop: one k-tile per iteration")
    o("1:")

    groups: list[list[str]] = []
    for ks in range(4):
        for nb in range(4):
            for mb in range(4):
                head = []
                if nb == 0 and mb == 0:
                    head.append(f"@wait:f{ks}")
                    if ks == 3:
                        head += ["s_waitcnt vmcnt(0)", "s_barrier"]
                groups.append(head + [f"v_mfma_f32_32x32x16_bf16 {ACC(nb, mb)}, {WF(ks & 1, nb)}, {XF(ks & 1, mb)}, {ACC(nb, mb)}"])
    segs: list[tuple] = []
    for ks in range(3):
        if "noread" not in DBG:
            segs.append((frag_reads(ks + 1, (ks + 1) & 1), 16 * ks, 16 * ks + 13))
    # k-step 3: the stage toggles (fragment addresses: next tile's stage; DMA target: the stage this tile leaves), k-step 0 of the next tile, the staging of tile t + 2
    toggles = [f"v_xor_b32_e32 {vr(r)}, 0x10000, {vr(r)}" for r in (XA[0], WA[0])]
    segs.append((toggles, 48, 50, "chain"))
    if "noread" not in DBG:
        segs.append((frag_reads(0, 0), 48, 62, "chain"))
    segs.append(([f"v_xor_b32_e32 {vr(r)}, 0x10000, {vr(r)}" for r in XA[1:] + WA[1:]], 50, 63))
    # (the selects first: the adds overwrite SCC)
    adv = [f"s_cmp_gt_u32 s{S_REM}, 0", f"s_cselect_b32 s{S_INC}, 128, 0", f"s_cselect_b32 s{S_DEC}, 1, 0", f"s_add_u32 s{S_KA}, s{S_KA}, s{S_INC}",
           f"s_addc_u32 s{S_KA + 1}, s{S_KA + 1}, 0", f"s_add_u32 s{S_KB}, s{S_KB}, s{S_INC}", f"s_addc_u32 s{S_KB + 1}, s{S_KB + 1}, 0",
           f"s_sub_u32 s{S_REM}, s{S_REM}, s{S_DEC}"]
    segs.append(([adv], 48, 63, "chain"))
    if "nodma" not in DBG:
        segs.append((dma_pieces(), 48, 63, "chain"))
    segs.append(([f"s_xor_b32 s{S_T0}, s{S_T0}, 0x10000"], 63, 63))
    body = weave_budget(groups, segs, cap=CAP)
    st.extend(body)
    o(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    o(f"s_cmp_lg_u32 s{S_CNT}, 0")
    o("s_cbranch_scc1 1b")
    st.comment("---- epilogue: C[m, n] = bf16(acc): lane = token m (lane & 31), registers 4a + b = feature 8a + 4 (lane >> 5) + b of the 32 x 32 block")
    o("s_waitcnt vmcnt(0) lgkmcnt(0)")
    o("s_nop 7")
    o("s_nop 7")
    o(f"v_mov_b32_e32 {vr(CP[0])}, %[cplo]")
    o(f"v_mov_b32_e32 {vr(CP[0] + 1)}, %[cphi]")
    for mb in range(1, 4):
        o(f"v_add_co_u32_e32 {vr(CP[mb])}, vcc, %[cstep], {vr(CP[mb - 1])}")
        o(f"v_addc_co_u32_e32 {vr(CP[mb] + 1)}, vcc, 0, {vr(CP[mb - 1] + 1)}, vcc")
    k = 0
    for nb in range(4):
        for mb in range(4):
            a0 = (nb * 4 + mb) * 16
            for a in range(4):
                t = T[(k % 4) * 4:(k % 4) * 4 + 4]
                k += 1
                for b in range(4):
                    o(f"v_accvgpr_read_b32 {vr(t[b])}, a{a0 + 4 * a + b}")
                o(f"v_cvt_pk_bf16_f32 {vr(t[0])}, {vr(t[0])}, {vr(t[1])}")
                o(f"v_cvt_pk_bf16_f32 {vr(t[1])}, {vr(t[2])}, {vr(t[3])}")
                o(f"global_store_dwordx2 {vr(CP[mb], 2)}, {vr(t[0], 2)}, off offset:{nb * 64 + a * 16}")
    o("s_waitcnt vmcnt(0)")
    o(f"s_mov_b32 m0, s{S_M0}")
    lines = resolve_lgkm(st.lines, loop_label="1:", loop_branch="s_cbranch_scc1 1b")
    bad = check_hazards(lines)
    if bad:
        raise SystemExit("hazard check failed:\n" + "\n".join(bad[:20]))
    return "\n".join('"' + ln.replace("\\", "\\\\") + '\\n"' for ln in lines) + "\n"


def main() -> None:
    out = os.environ.get("GEMM4W_OUT") or os.path.join(os.path.dirname(__file__), "..", "gemm4w_body.inc")
    body = build()
    with open(out, "w") as f:
        f.write("// GENERATED by tools/kgen/gemm4w.py (lab only) — regenerate with  python -m tools.kgen.gemm4w\n")
        f.write(body)
    regs = [f'"v{i}"' for i in range(40, 152)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(40, 52)]
    with open(os.path.join(os.path.dirname(out), "gemm4w_clobbers.inc"), "w") as f:
        f.write("// GENERATED by tools/kgen/gemm4w.py — the registers the body owns\n")
        f.write(",\n".join(", ".join(regs[i:i + 16]) for i in range(0, len(regs), 16)) + "\n")
    print(f"wrote {os.path.normpath(out)}: {body.count(chr(10))} lines", file=sys.stderr)


if __name__ == "__main__":
    main()
