"""kgen.dq64 — instruction stream of k_attn_bwd_dq64 (attention backward, dQ; head_dim 128; 4 waves x 64 queries, one wave per SIMD).

  python -m tools.kgen.dq64        -> simpletuner_amd/csrc/gen/attn_dq64_body.inc   (the text of one asm statement; see attention_bwd.hip)

Math per 32-key block j of the key stream (two blocks per 64-key tile), both 32-query blocks qb of the wave:
    A(j): S^T  = K Q^T, dP^T = V dO^T               32 MFMAs   K / V row fragments from LDS feed both query blocks
    B(j): dS   = exp2(S * scale2 - lse) * (dP - delta)   VALU, in place in the S registers, packed to bf16 MFMA operands
    C(j): dQ^T += K^T dS^T                          16 MFMAs   K^T fragments by transposing LDS reads feed both query blocks
Pipeline: step j issues  A(j+1) with B(j) woven into its MFMA gaps, then C(j).  Two generations of S / dP registers alternate.

Registers (the kernel lists v[40:255], a[0:255], s[40:63] as clobbers):
    a[0:127]    dQ^T accumulators  ACC(qb, dt)            a[128:191] Q fragments QF(qb, ks)      a[192:255] dO fragments DOF(qb, ks)
    v[64:127]   generation 0: S(qb) 64.., 80..; dP(qb) 96.., 112..        v[128:191] generation 1
    v[192:207]  dS fragments DSF(qb, m)        v[208:223] K / V row fragments (2 deep)     v[224:231] K^T fragments (2 deep)
    v[232:239]  LDS-DMA lane offsets (4 K pieces, 4 V pieces)      v[240:247] scratch addresses
LDS: ring of three 32-KiB slots [K image 16 KiB | V image 16 KiB], tile t in slot t % 3; after the loop the ring is reused to park dQ (bf16) for the
token-major epilogue written in HIP.
"""
from __future__ import annotations

import os
import sys

from .emit import Stream, ar, check_hazards, vr, weave

MFMA = "v_mfma_f32_32x32x16_bf16"
TILE = 16384           # one tile image
SLOT = 2 * TILE


def ACC(qb, dt): return ar(16 * (4 * qb + dt), 16)
def QF(qb, ks): return ar(128 + 4 * (8 * qb + ks), 4)
def DOF(qb, ks): return ar(192 + 4 * (8 * qb + ks), 4)
def S(g, qb): return 64 + 64 * g + 16 * qb
def P(g, qb): return 96 + 64 * g + 16 * qb
def DSF(qb, m): return 192 + 4 * (2 * qb + m)
def KF(i): return vr(208 + 8 * i, 4)
def VF(i): return vr(212 + 8 * i, 4)
def TF(i): return 224 + 4 * i
KOF = [232, 233, 234, 235]
VOF = [236, 237, 238, 239]
V_ROWN, V_RA, V_TRC, V_TA0, V_TA1 = 240, 241, 242, 243, 244       # slot-relative row base of the A block, temp row address, tr base of the C block, temps
# SGPRs
S_KP, S_VP = 40, 42          # running global pointers of the NEXT tile to stage (64-bit)
S_CNT = 44                   # tiles left for the main loop
S_CUR, S_NXT, S_STG = 45, 46, 47     # LDS byte addresses of the slots: tile kt, tile kt+1, stage target
S_M0, S_T0, S_T1, S_LEFT = 48, 49, 50, 51


def a_groups(g_new: int, sb: int, first_frag_loaded: bool) -> list[list[str]]:
    """A of one 32-key block into generation g_new: 8 k-steps x 4 MFMAs; fragments of k-step ks+1 are requested before the MFMAs of ks.
    V_ROWN holds the (slot + lane) row base of the tile the block lives in; sb*8192 picks the block."""
    groups: list[list[str]] = []
    for ks in range(8):
        cur = ks & 1
        head = []
        if ks == 0 and not first_frag_loaded:
            head += [f"ds_read_b128 {KF(0)}, {vr(V_ROWN)} offset:{sb * 8192}", f"ds_read_b128 {VF(0)}, {vr(V_ROWN)} offset:{sb * 8192 + TILE}"]
        head.append("s_waitcnt lgkmcnt(0)")
        if ks < 7:
            head += [f"v_xor_b32_e32 {vr(V_RA)}, {hex((ks + 1) << 5)}, {vr(V_ROWN)}",
                     f"ds_read_b128 {KF(cur ^ 1)}, {vr(V_RA)} offset:{sb * 8192}",
                     f"ds_read_b128 {VF(cur ^ 1)}, {vr(V_RA)} offset:{sb * 8192 + TILE}"]
        c = (lambda r: "0") if ks == 0 else (lambda r: r)
        groups.append(head + [f"{MFMA} {vr(S(g_new, 0), 16)}, {KF(cur)}, {QF(0, ks)}, {c(vr(S(g_new, 0), 16))}"])
        groups.append([f"{MFMA} {vr(P(g_new, 0), 16)}, {VF(cur)}, {DOF(0, ks)}, {c(vr(P(g_new, 0), 16))}"])
        groups.append([f"{MFMA} {vr(S(g_new, 1), 16)}, {KF(cur)}, {QF(1, ks)}, {c(vr(S(g_new, 1), 16))}"])
        groups.append([f"{MFMA} {vr(P(g_new, 1), 16)}, {VF(cur)}, {DOF(1, ks)}, {c(vr(P(g_new, 1), 16))}"])
    return groups


def b_ops(g: int) -> list[str]:
    """dS of generation g, in place: S <- exp2(S*scale2 - lse) ; P <- dP - delta ; S <- S * P ; DSF <- bf16 pairs.  Emitted in groups of four scores
    so that no instruction reads the result of its predecessor (a transcendental needs one state before its consumer)."""
    ops: list[str] = []
    for qb in range(2):
        nl, dl = f"%[nlse{qb}]", f"%[del{qb}]"
        for r0 in range(0, 16, 4):
            rs = [S(g, qb) + r0 + i for i in range(4)]
            ps = [P(g, qb) + r0 + i for i in range(4)]
            ops += [f"v_fma_f32 {vr(r)}, {vr(r)}, %[scale2], {nl}" for r in rs]
            ops += [f"v_exp_f32_e32 {vr(r)}, {vr(r)}" for r in rs]
            ops += [f"v_sub_f32_e32 {vr(p)}, {vr(p)}, {dl}" for p in ps]
            ops += [f"v_mul_f32_e32 {vr(r)}, {vr(r)}, {vr(p)}" for r, p in zip(rs, ps)]
            m, i0 = r0 >> 3, (r0 & 7) >> 1
            ops += [f"v_cvt_pk_bf16_f32 {vr(DSF(qb, m) + i0 + i)}, {vr(rs[2 * i])}, {vr(rs[2 * i + 1])}" for i in range(2)]
    return ops


def tr_reads(i: int, sb: int) -> list[str]:
    """the K^T fragment of C iteration i (dt = i >> 1, m = i & 1): two transposing reads; V_TRC = slot + lane tr base of the C block's tile"""
    dt, m = i >> 1, i & 1
    t = TF(i & 1)
    x0, x1 = (4 * dt) << 4, ((4 * dt) ^ 1) << 4
    out = []
    out.append(f"v_xor_b32_e32 {vr(V_TA0)}, {hex(x0)}, {vr(V_TRC)}" if x0 else f"v_mov_b32_e32 {vr(V_TA0)}, {vr(V_TRC)}")
    out.append(f"v_xor_b32_e32 {vr(V_TA1)}, {hex(x1)}, {vr(V_TRC)}")
    out.append(f"ds_read_b64_tr_b16 {vr(t, 2)}, {vr(V_TA0)} offset:{sb * 8192 + 16 * m * 256}")
    out.append(f"ds_read_b64_tr_b16 {vr(t + 2, 2)}, {vr(V_TA1)} offset:{sb * 8192 + (16 * m + 4) * 256}")
    return out


def c_groups(sb: int, extra_at: dict[int, list[str]] | None = None) -> list[list[str]]:
    """C of one block: 8 iterations x 2 MFMAs; the fragment of iteration i+1 is requested before the MFMAs of i (fragment 0 was requested by the caller).
    extra_at[i]: lines placed right after the fragment request of iteration i (barrier, prefetch of the next A block)."""
    extra_at = extra_at or {}
    groups: list[list[str]] = []
    for i in range(8):
        dt, m = i >> 1, i & 1
        head = ["s_waitcnt lgkmcnt(0)"]
        if i < 7:
            head += tr_reads(i + 1, sb)
        head += extra_at.get(i, [])
        t = vr(TF(i & 1), 4)
        groups.append(head + [f"{MFMA} {ACC(0, dt)}, {t}, {vr(DSF(0, m), 4)}, {ACC(0, dt)}"])
        groups.append([f"{MFMA} {ACC(1, dt)}, {t}, {vr(DSF(1, m), 4)}, {ACC(1, dt)}"])
    return groups


def stage_ops() -> list[str]:
    """LDS-DMA of one 64-key tile (this wave's 4 K pieces + 4 V pieces) into the slot at S_STG, from S_KP / S_VP; then advance the pointers.
    %[wvoff] = wave * 1024 (the wave's first piece inside an image)."""
    ops = [f"s_add_u32 s{S_T0}, s{S_STG}, %[wvoff]"]
    for p in range(4):
        ops += [f"s_add_u32 m0, s{S_T0}, {p * 4096}", "s_nop 0", f"global_load_lds_dwordx4 {vr(KOF[p])}, s[{S_KP}:{S_KP + 1}]"]
    for p in range(4):
        ops += [f"s_add_u32 m0, s{S_T0}, {p * 4096 + TILE}", "s_nop 0", f"global_load_lds_dwordx4 {vr(VOF[p])}, s[{S_VP}:{S_VP + 1}]"]
    ops += [f"s_add_u32 s{S_KP}, s{S_KP}, {64 * 256}", f"s_addc_u32 s{S_KP + 1}, s{S_KP + 1}, 0",
            f"s_add_u32 s{S_VP}, s{S_VP}, %[vstep]", f"s_addc_u32 s{S_VP + 1}, s{S_VP + 1}, 0"]
    return ops


def rotate_slots() -> list[str]:
    """(cur, nxt, stg) <- (nxt, stg, cur)"""
    return [f"s_mov_b32 s{S_T1}, s{S_CUR}", f"s_mov_b32 s{S_CUR}, s{S_NXT}", f"s_mov_b32 s{S_NXT}, s{S_STG}", f"s_mov_b32 s{S_STG}, s{S_T1}"]


def set_block_bases(a_slot: int | None, c_slot: int | None) -> list[str]:
    ops = []
    if a_slot is not None:
        ops.append(f"v_add_u32_e32 {vr(V_ROWN)}, s{a_slot}, %[rowb]")
    if c_slot is not None:
        ops.append(f"v_add_u32_e32 {vr(V_TRC)}, s{c_slot}, %[trb]")
    return ops


def prefetch_a(sb: int) -> list[str]:
    return [f"ds_read_b128 {KF(0)}, {vr(V_ROWN)} offset:{sb * 8192}", f"ds_read_b128 {VF(0)}, {vr(V_ROWN)} offset:{sb * 8192 + TILE}"]


def build(b_in_a: int = 144) -> str:
    """b_in_a: how many of the 144 VALU instructions of B are woven into A's gaps (the rest rides in front of C's first MFMAs)"""
    st = Stream()
    o = st.op
    st.comment("---- prologue: Q / dO fragments -> a[128:255], zero dQ accumulators, slot addresses, first two tiles")
    o(f"s_mov_b32 s{S_M0}, m0")
    for qb in range(2):
        for ks in range(8):
            o(f"global_load_dwordx4 {QF(qb, ks)}, %[qp{qb}], off offset:{32 * ks}")
            o(f"global_load_dwordx4 {DOF(qb, ks)}, %[dp{qb}], off offset:{32 * ks}")
    for i in range(128):
        o(f"v_accvgpr_write_b32 a{i}, 0")
    # lane offsets of the DMA pieces: piece p of this wave starts 16 rows (K: 4096 B, V: 16 * ld_v * 2 B = %[vrow16]) after piece p - 1
    o(f"v_mov_b32_e32 {vr(KOF[0])}, %[koff]")
    o(f"v_mov_b32_e32 {vr(VOF[0])}, %[voff]")
    for p in range(1, 4):
        o(f"v_add_u32_e32 {vr(KOF[p])}, {p * 4096}, {vr(KOF[0])}")
        o(f"v_add_u32_e32 {vr(VOF[p])}, %[vrow16], {vr(VOF[p - 1])}")
    o(f"s_mov_b64 s[{S_KP}:{S_KP + 1}], %[kbase]")
    o(f"s_mov_b64 s[{S_VP}:{S_VP + 1}], %[vbase]")
    o(f"s_mov_b32 s{S_CUR}, %[lds]")
    o(f"s_add_u32 s{S_NXT}, %[lds], {SLOT}")
    o(f"s_add_u32 s{S_STG}, %[lds], {2 * SLOT}")
    # stage tile 0 -> slot 0, tile 1 -> slot 1 (if any): stage_ops() targets S_STG, so point it at the slot in turn
    o(f"s_mov_b32 s{S_T1}, s{S_STG}")
    o(f"s_mov_b32 s{S_STG}, s{S_CUR}")
    st.extend(stage_ops())
    o("s_cmp_lt_u32 %[nkt], 2")
    o("s_cbranch_scc1 .Ldq64_one_tile_%=")
    o(f"s_mov_b32 s{S_STG}, s{S_NXT}")
    st.extend(stage_ops())
    o(".Ldq64_one_tile_%=:")
    o(f"s_mov_b32 s{S_STG}, s{S_T1}")
    o(f"s_sub_u32 s{S_CNT}, %[nkt], 1")                      # main-loop trips: tiles 0 .. nkt-2 (the last tile is peeled)
    o("s_waitcnt vmcnt(0)")
    o("s_barrier")
    st.comment("---- A(tile 0, block 0) -> generation 0")
    st.extend(set_block_bases(S_CUR, None))
    for g in a_groups(0, 0, first_frag_loaded=False):
        st.extend(g)
    st.extend(prefetch_a(1))                                 # first fragments of A(tile 0, block 1): every later step finds its A block's first fragments requested
    o("s_nop 7")                                             # MFMA results -> the VALU of the first B (in steady state C's 16 MFMAs sit in between)

    def step(g_cur: int, sb_c: int, a_sb: int | None, c_extra: dict[int, list[str]] | None, pre_c: list[str] | None = None) -> None:
        """[A(next block) woven with B(g_cur)] ; C(block sb_c of the current tile)."""
        b = b_ops(g_cur)
        if a_sb is not None:
            ag = a_groups(g_cur ^ 1, a_sb, first_frag_loaded=True)
            nb = min(b_in_a, len(b))
            # the fragment of C's first iteration is requested in A's last k-step (no A request there)
            ag[28] = ag[28][:1] + tr_reads(0, sb_c) + ag[28][1:]
            st.extend(weave(ag, b[:nb]))
            rest = b[nb:]
        else:
            st.extend(tr_reads(0, sb_c))
            rest = b
        st.extend(rest)
        if pre_c:
            st.extend(pre_c)
        if rest or a_sb is None:
            o("s_nop 1")                                  # VALU -> MFMA operand
        for g in c_groups(sb_c, c_extra):
            st.extend(g)

    st.comment("---- main loop: tiles 0 .. nkt-2")
    o(f"s_cmp_eq_u32 s{S_CNT}, 0")
    o("s_cbranch_scc1 .Ldq64_last_%=")
    o(".Ldq64_loop_%=:")
    st.comment("step 1: A(kt, 1) -> gen 1 | B(gen 0) ; C(kt, 0); barrier for tile kt+1 inside C, then the first fragments of A(kt+1, 0)")
    st.extend(set_block_bases(None, S_CUR))
    barrier = ["s_waitcnt vmcnt(0)", "s_barrier"]
    nxt_base = [f"v_add_u32_e32 {vr(V_ROWN)}, s{S_NXT}, %[rowb]"]
    step(0, 0, 1, {5: barrier, 6: nxt_base, 7: prefetch_a(0)})
    st.comment("step 2: stage(kt+2) ; A(kt+1, 0) -> gen 0 | B(gen 1) ; C(kt, 1); then the first fragments of A(kt+1, 1)")
    o(f"s_cmp_lt_u32 s{S_CNT}, 2")                          # tiles kt+2 exists iff trips left >= 2
    o("s_cbranch_scc1 .Ldq64_nostage_%=")
    st.extend(stage_ops())
    o(".Ldq64_nostage_%=:")
    step(1, 1, 0, {7: prefetch_a(1)})
    st.extend(rotate_slots())
    st.extend(set_block_bases(S_CUR, None))                  # V_ROWN was already pointing at the new current tile; keep it explicit
    o(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    o(f"s_cmp_lg_u32 s{S_CNT}, 0")
    o("s_cbranch_scc1 .Ldq64_loop_%=")
    o(".Ldq64_last_%=:")
    st.comment("---- last tile: A(last, 1) | B(gen 0) ; C(last, 0) ; B(gen 1) ; C(last, 1)")
    st.extend(set_block_bases(S_CUR, S_CUR))
    step(0, 0, 1, None)
    step(1, 1, None, None)
    st.comment("---- park dQ^T * scale as bf16 in the (idle) ring: token rows of 256 bytes, 16-byte chunk c of token t at c ^ (t & 15)  (rope_bwd_store's image)")
    o("s_nop 15")
    o("s_barrier")                                           # every wave is past its last ring read
    # %[park] = lane part: l31*256 + ((l31 & 15) << 4) + 8*h ; wave slice (2 x 8 KiB) at lds + wave*16384
    o(f"s_lshl_b32 s{S_T0}, %[wvoff], 4")
    o(f"s_add_u32 s{S_T0}, s{S_T0}, %[lds]")
    o(f"v_add_u32_e32 {vr(V_ROWN)}, s{S_T0}, %[park]")
    for qb in range(2):
        for dt in range(4):
            for a in range(4):
                base = 16 * (4 * qb + dt) + 4 * a
                t = 64 + 4 * ((4 * dt + a) & 3)       # rotate through 4 temp quads
                for bb in range(4):
                    o(f"v_accvgpr_read_b32 {vr(t + bb)}, a{base + bb}")
                for bb in range(4):
                    o(f"v_mul_f32_e32 {vr(t + bb)}, %[scale], {vr(t + bb)}")
                o(f"v_cvt_pk_bf16_f32 {vr(t)}, {vr(t)}, {vr(t + 1)}")
                o(f"v_cvt_pk_bf16_f32 {vr(t + 1)}, {vr(t + 2)}, {vr(t + 3)}")
                ch = 4 * dt + a
                o(f"v_xor_b32_e32 {vr(V_RA)}, {hex(ch << 4)}, {vr(V_ROWN)}" if ch else f"v_mov_b32_e32 {vr(V_RA)}, {vr(V_ROWN)}")
                o(f"ds_write_b64 {vr(V_RA)}, {vr(t, 2)} offset:{qb * 8192}")
    o("s_waitcnt lgkmcnt(0)")
    o(f"s_mov_b32 m0, s{S_M0}")
    lines = st.lines
    bad = check_hazards(lines)
    if bad:
        raise SystemExit("hazard check failed:\n" + "\n".join(bad[:20]))
    return "\n".join('"' + ln.replace("\\", "\\\\") + '\\n"' for ln in lines) + "\n"


def main() -> None:
    out = os.path.join(os.path.dirname(__file__), "..", "..", "simpletuner_amd", "csrc", "gen", "attn_dq64_body.inc")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    b_in_a = int(os.environ.get("DQ64_B_IN_A", "144"))
    body = build(b_in_a)
    with open(out, "w") as f:
        f.write("// GENERATED by tools/kgen/dq64.py — do not edit; regenerate with  python -m tools.kgen.dq64\n")
        f.write(body)
    print(f"wrote {os.path.normpath(out)}: {body.count(chr(10))} lines", file=sys.stderr)


if __name__ == "__main__":
    main()
