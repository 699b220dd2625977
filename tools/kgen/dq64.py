"""kgen.dq64 — instruction stream of k_attn_bwd_dq64 (attention backward, dQ; head_dim 128; 4 waves x 64 queries, one wave per SIMD).

  python -m tools.kgen.dq64        -> simpletuner_amd/csrc/gen/attn_dq64_body.inc   (the text of one asm statement; see attention_bwd.hip)

Math per 32-key block j of the key stream (two blocks per 64-key tile), both 32-query blocks qb of the wave:
    A(j): S^T  = K Q^T, dP^T = V dO^T               32 MFMAs   K / V row fragments from LDS feed both query blocks
    B(j): dS   = exp2(S * scale2 - lse) * (dP - delta)   VALU, in place in the S registers, packed to bf16 MFMA operands
    C(j): dQ^T += K^T dS^T                          16 MFMAs   K^T fragments by transposing LDS reads feed both query blocks
Pipeline: step j issues  A(j+1) with B(j) woven into its MFMA gaps, then C(j).  Two generations of S / dP registers alternate.
LDS operand fragments are requested TWO groups ahead of the MFMAs that use them (three register buffers per fragment kind, counted lgkmcnt waits
resolved by kgen.emit.resolve_lgkm on the final instruction order): one request per group of 4 (A) / 2 (C) MFMAs, the A groups without a request of
their own (k-steps 6, 7) carry C's first two, C's last two carry the next A block's first two.
K / V tiles arrive by LDS-DMA into a ring of three 32-KiB slots [K image | V image], tile t in slot t % 3, staged one tile ahead: the 8 pieces a wave
issues per tile are spread over the MFMA gaps of C in the second step of a tile; one workgroup barrier per tile (inside C of the first step).

Registers (the kernel lists v[32:255], a[0:255], s[40:63] as clobbers):
    a[0:127]    dQ^T accumulators  ACC(qb, dt)            a[128:191] Q fragments QF(qb, ks)      a[192:255] dO fragments DOF(qb, ks)
    v[64:127]   generation 0: S(qb) 64.., 80..; dP(qb) 96.., 112..        v[128:191] generation 1
    v[192:207]  dS fragments DSF(qb, m)        v[208:231] K / V row fragments (3 deep)     v[232:243] K^T fragments (3 deep)
    v[244:251]  LDS-DMA lane offsets (4 K pieces, 4 V pieces)      v[56:63] scratch      v[32:39] row-fragment addresses ROWA[ks] of the A block's tile
    v[40:47]    transposed-fragment addresses TRA[j] of the C block's tile (both sets rebuilt once per tile)
After the loop the ring is reused to park dQ (bf16) for the token-major epilogue written in HIP.
"""
from __future__ import annotations

import os
import sys

from .emit import Stream, ar, check_hazards, resolve_lgkm, vr, weave, weave_budget

MFMA = "v_mfma_f32_32x32x16_bf16"
TILE = 16384           # one tile image (256-byte row pitch at every head_dim: narrower heads leave chunk slots unused, as in k_attn_bwd_dq<., TR>)
SLOT = 2 * TILE
HD = int(os.environ.get("DQ64_HD", "128"))       # head_dim: 128 (Flux), 96 (PixArt's 72, zero padded), 64 (SD3)
assert HD in (64, 96, 128)
# head_dim 96 = a zero-padded narrower head (<= 80 valid channels): S = Q K^T and dP = dO V^T contract over 5 k-steps (80 channels); dQ^T keeps its 3 d tiles.
# r5 lab, B1 H16 S16384: 1.680 -> 1.618 ms, dQ bit-identical to the 6-k-step 32-row kernel (the sixth k-step only adds +0)
NKS, NDT = int(os.environ.get("DQ64_KS", {128: 8, 96: 5, 64: 4}[HD])), HD // 32                     # MFMA k-steps over the head dim, 32-row d tiles of dQ^T
NACC = 2 * NDT * 16                               # dQ^T accumulator registers; Q fragments follow, then dO fragments


def ACC(qb, dt): return ar(16 * (NDT * qb + dt), 16)
def QF(qb, ks): return ar(NACC + 4 * (NKS * qb + ks), 4)
def DOF(qb, ks): return ar(NACC + 8 * NKS + 4 * (NKS * qb + ks), 4)
def S(g, qb): return 64 + 64 * g + 16 * qb
def P(g, qb): return 96 + 64 * g + 16 * qb
def DSF(qb, m): return 192 + 4 * (2 * qb + m)
def KF(i): return vr(208 + 8 * (i % 3), 4)
def VF(i): return vr(212 + 8 * (i % 3), 4)
def TF(i): return 232 + 4 * (i % 3)
KOF = [244, 245, 246, 247]
VOF = [248, 249, 250, 251]
V_ROWN, V_RA = 56, 57                                             # scratch (park epilogue)
ROWA = [32 + k for k in range(8)]                                 # (slot + lane row base) ^ (ks << 5): K / V row fragments of k-step ks
TRX = [0x00, 0x10, 0x40, 0x50, 0x80, 0x90, 0xc0, 0xd0]            # chunk XORs of the transposed reads: (4 dt) << 4 and ((4 dt) ^ 1) << 4
TRA = [40 + k for k in range(8)]                                  # (slot + lane tr base) ^ TRX[j]
CAP = float(os.environ.get("DQ64_CAP", {128: "5", 96: "7", 64: "9"}[int(os.environ.get("DQ64_HD", "128"))]))    # issues per MFMA gap besides the MFMA (narrower heads: the same VALU per score under fewer MFMAs)
# SGPRs
S_KP, S_VP = 40, 42          # global pointers of the tile to stage next (64-bit), always a valid tile
S_CNT = 44                   # main-loop trips left
S_CUR, S_NXT, S_STG = 45, 46, 47     # LDS byte addresses of the slots: tile kt, tile kt+1, stage target
S_M0, S_T0, S_T1, S_INCK, S_INCV = 48, 49, 50, 51, 52

TRACE = bool(os.environ.get("DQ64_TRACE"))       # lab builds: s_memtime stamps at the phase boundaries of the main loop, dumped by block 0 through %[trace]
DBG = set(filter(None, os.environ.get("DQ64_DBG", "").split(",")))      # timing experiments only (results are wrong): nostage, nobarrier, nob, nosub


def a_request(ks: int, sb: int) -> list[str]:
    """request the K / V row fragments of k-step ks of the A block (its tile's addresses in ROWA, block sb) into buffer ks % 3"""
    return [f"ds_read_b128 {KF(ks)}, {vr(ROWA[ks])} offset:{sb * 8192} ;@ld:A{ks}",
            f"ds_read_b128 {VF(ks)}, {vr(ROWA[ks])} offset:{sb * 8192 + TILE} ;@ld:A{ks}"]


def rowa_update(slot_sgpr: int) -> list[str]:
    return [f"v_add_u32_e32 {vr(ROWA[0])}, s{slot_sgpr}, %[rowb]"] + [f"v_xor_b32_e32 {vr(ROWA[k])}, {hex(k << 5)}, {vr(ROWA[0])}" for k in range(1, NKS)]


def tra_update(slot_sgpr: int) -> list[str]:
    return [f"v_add_u32_e32 {vr(TRA[0])}, s{slot_sgpr}, %[trb]"] + [f"v_xor_b32_e32 {vr(TRA[j])}, {hex(TRX[j])}, {vr(TRA[0])}" for j in range(1, 2 * NDT)]


def c_request(i: int, sb: int) -> list[str]:
    """request the K^T fragment of C iteration i (m = i >> 2, dt = i & 3: every accumulator still sees m = 0 before m = 1): two transposing reads
    from the C block's tile (addresses in TRA)"""
    dt, m = i % NDT, i // NDT
    t = TF(i)
    return [f"ds_read_b64_tr_b16 {vr(t, 2)}, {vr(TRA[2 * dt])} offset:{sb * 8192 + 16 * m * 256} ;@ld:C{i}",
            f"ds_read_b64_tr_b16 {vr(t + 2, 2)}, {vr(TRA[2 * dt + 1])} offset:{sb * 8192 + (16 * m + 4) * 256} ;@ld:C{i}"]


def a_groups(g_new: int, sb: int, tail_requests: list[list[str]]) -> list[list[str]]:
    """A of one 32-key block into generation g_new: 8 k-steps x 4 MFMAs.  Group ks waits for fragment ks and requests fragment ks + 2;
    k-steps 6 and 7 carry tail_requests[0 / 1] instead (the first two fragments of the C that follows)."""
    groups: list[list[str]] = []
    for ks in range(NKS):
        head = [f"@wait:A{ks}"]
        head += a_request(ks + 2, sb) if ks < NKS - 2 else tail_requests[ks - (NKS - 2)]
        c = (lambda r: "0") if ks == 0 else (lambda r: r)
        groups.append(head + [f"{MFMA} {vr(S(g_new, 0), 16)}, {KF(ks)}, {QF(0, ks)}, {c(vr(S(g_new, 0), 16))}"])
        groups.append([f"{MFMA} {vr(P(g_new, 0), 16)}, {VF(ks)}, {DOF(0, ks)}, {c(vr(P(g_new, 0), 16))}"])
        groups.append([f"{MFMA} {vr(S(g_new, 1), 16)}, {KF(ks)}, {QF(1, ks)}, {c(vr(S(g_new, 1), 16))}"])
        groups.append([f"{MFMA} {vr(P(g_new, 1), 16)}, {VF(ks)}, {DOF(1, ks)}, {c(vr(P(g_new, 1), 16))}"])
    return groups


def b_ops(g: int) -> list[str]:
    """dS of generation g, in place: S <- exp2(S*scale2 - lse) ; P <- dP - delta ; S <- S * P ; DSF <- bf16 pairs.  Emitted in groups of four scores
    so that no instruction reads the result of its predecessor (a transcendental needs one state before its consumer).  Order: the first 8 key rows of
    both query blocks (-> DSF(., 0), what C's first four iterations contract over), then the other 8 (-> DSF(., 1))."""
    ops: list[str] = []
    for m in range(2):
        for qb in range(2):
            nl, dl = f"%[nlse{qb}]", f"%[del{qb}]"
            for r0 in (8 * m, 8 * m + 4):
                rs = [S(g, qb) + r0 + i for i in range(4)]
                ps = [P(g, qb) + r0 + i for i in range(4)]
                ops += [f"v_fma_f32 {vr(r)}, {vr(r)}, %[scale2], {nl}" for r in rs]
                ops += [f"v_exp_f32_e32 {vr(r)}, {vr(r)}" for r in rs]
                if "nosub" not in DBG:
                    ops += [f"v_sub_f32_e32 {vr(p)}, {vr(p)}, {dl}" for p in ps]
                ops += [f"v_mul_f32_e32 {vr(r)}, {vr(r)}, {vr(p)}" for r, p in zip(rs, ps)]
                i0 = (r0 & 7) >> 1
                ops += [f"v_cvt_pk_bf16_f32 {vr(DSF(qb, m) + i0 + i)}, {vr(rs[2 * i])}, {vr(rs[2 * i + 1])}" for i in range(2)]
    return ops


def c_groups(sb: int, tail_requests: list[list[str]], extra_at: dict[int, list[str]] | None = None) -> list[list[str]]:
    """C of one block: 8 iterations x 2 MFMAs.  Iteration i waits for fragment i and requests fragment i + 2; iterations 6 and 7 carry
    tail_requests[0 / 1] (the first two fragments of the next A block).  extra_at[i]: lines placed in front of the request of iteration i."""
    extra_at = extra_at or {}
    groups: list[list[str]] = []
    for i in range(2 * NDT):
        dt, m = i % NDT, i // NDT
        head = [f"@wait:C{i}"] + extra_at.get(i, [])
        head += c_request(i + 2, sb) if i < 2 * NDT - 2 else tail_requests[i - (2 * NDT - 2)]
        t = vr(TF(i), 4)
        groups.append(head + [f"{MFMA} {ACC(0, dt)}, {t}, {vr(DSF(0, m), 4)}, {ACC(0, dt)}"])
        groups.append([f"{MFMA} {ACC(1, dt)}, {t}, {vr(DSF(1, m), 4)}, {ACC(1, dt)}"])
    return groups


def stage_pieces() -> list[list[str]]:
    """LDS-DMA of one 64-key tile (this wave's 4 K pieces + 4 V pieces) into the slot at S_STG, from S_KP / S_VP: one list per piece (each fits an MFMA gap).
    %[wvoff] = wave * 1024 (the wave's first piece inside an image).  S_T0 = S_STG + wvoff must be set first (stage_begin)."""
    pcs = []
    for p in range(4):
        pcs.append([f"s_add_u32 m0, s{S_T0}, {p * 4096}", "s_nop 0", f"global_load_lds_dwordx4 {vr(KOF[p])}, s[{S_KP}:{S_KP + 1}]"])
    for p in range(4):
        pcs.append([f"s_add_u32 m0, s{S_T0}, {p * 4096 + TILE}", "s_nop 0", f"global_load_lds_dwordx4 {vr(VOF[p])}, s[{S_VP}:{S_VP + 1}]"])
    return pcs


def stage_begin() -> list[str]:
    return [f"s_add_u32 s{S_T0}, s{S_STG}, %[wvoff]"]


def stage_advance(inck: str, incv: str) -> list[str]:
    return [f"s_add_u32 s{S_KP}, s{S_KP}, {inck}", f"s_addc_u32 s{S_KP + 1}, s{S_KP + 1}, 0",
            f"s_add_u32 s{S_VP}, s{S_VP}, {incv}", f"s_addc_u32 s{S_VP + 1}, s{S_VP + 1}, 0"]


def rotate_slots() -> list[str]:
    """(cur, nxt, stg) <- (nxt, stg, cur)"""
    return [f"s_mov_b32 s{S_T1}, s{S_CUR}", f"s_mov_b32 s{S_CUR}, s{S_NXT}", f"s_mov_b32 s{S_NXT}, s{S_STG}", f"s_mov_b32 s{S_STG}, s{S_T1}"]


def build(b_in_a: int = 144) -> str:
    """b_in_a: how many of the 144 VALU instructions of B are woven into A's gaps (the rest rides in front of C's first MFMAs)"""
    st = Stream()
    o = st.op
    st.comment("---- prologue: Q / dO fragments -> a[128:255], zero dQ accumulators, slot addresses, first two tiles")
    o(f"s_mov_b32 s{S_M0}, m0")
    if TRACE:
        o("s_memtime s[74:75]")
    for qb in range(2):
        for ks in range(NKS):
            o(f"global_load_dwordx4 {QF(qb, ks)}, %[qp{qb}], off offset:{32 * ks}")
            o(f"global_load_dwordx4 {DOF(qb, ks)}, %[dp{qb}], off offset:{32 * ks}")
    for i in range(NACC):
        o(f"v_accvgpr_write_b32 a{i}, 0")
    # lane offsets of the DMA pieces: piece p of this wave starts 16 rows (K: 16 * head_dim * 2 B, V: 16 * ld_v * 2 B = %[vrow16]) after piece p - 1
    o(f"v_mov_b32_e32 {vr(KOF[0])}, %[koff]")
    o(f"v_mov_b32_e32 {vr(VOF[0])}, %[voff]")
    for p in range(1, 4):
        o(f"v_add_u32_e32 {vr(KOF[p])}, {p * 16 * HD * 2}, {vr(KOF[0])}")
        o(f"v_add_u32_e32 {vr(VOF[p])}, %[vrow16], {vr(VOF[p - 1])}")
    o(f"s_mov_b64 s[{S_KP}:{S_KP + 1}], %[kbase]")
    o(f"s_mov_b64 s[{S_VP}:{S_VP + 1}], %[vbase]")
    o(f"s_mov_b32 s{S_CUR}, %[lds]")
    o(f"s_add_u32 s{S_NXT}, %[lds], {SLOT}")
    o(f"s_add_u32 s{S_STG}, %[lds], {2 * SLOT}")
    # tile 0 -> slot 0, tile 1 -> slot 1 (if any); afterwards S_KP / S_VP point at tile min(2, nkt - 1)
    o(f"s_mov_b32 s{S_T1}, s{S_STG}")
    o(f"s_mov_b32 s{S_STG}, s{S_CUR}")
    st.extend(stage_begin())
    for pc in stage_pieces():
        st.extend(pc)
    o("s_cmp_lt_u32 %[nkt], 2")
    o("s_cbranch_scc1 .Ldq64_one_tile_%=")
    st.extend(stage_advance(str(64 * HD * 2), "%[vstep]"))
    o(f"s_mov_b32 s{S_STG}, s{S_NXT}")
    st.extend(stage_begin())
    for pc in stage_pieces():
        st.extend(pc)
    o("s_cmp_lt_u32 %[nkt], 3")
    o("s_cbranch_scc1 .Ldq64_one_tile_%=")
    st.extend(stage_advance(str(64 * HD * 2), "%[vstep]"))
    o(".Ldq64_one_tile_%=:")
    o(f"s_mov_b32 s{S_STG}, s{S_T1}")
    o(f"s_sub_u32 s{S_CNT}, %[nkt], 1")                      # main-loop trips: tiles 0 .. nkt-2 (the last tile is peeled)
    o("s_waitcnt vmcnt(0)")
    o("s_barrier")
    st.comment("---- A(tile 0, block 0) -> generation 0; then the first two fragments of A(tile 0, block 1)")
    st.extend(rowa_update(S_CUR))
    st.extend(tra_update(S_CUR))
    st.extend(a_request(0, 0))
    st.extend(a_request(1, 0))
    for g in a_groups(0, 0, [[], []]):
        st.extend(g)
    st.extend(a_request(0, 1))                               # (not inside the block above: with three fragment buffers its k-steps 6 and 7 still read buffers 0 and 1)
    st.extend(a_request(1, 1))
    o("s_nop 7")                                             # MFMA results -> the VALU of the first B (in steady state C's 16 MFMAs sit in between)

    def stamp(k: int) -> None:
        if TRACE:
            o(f"s_memtime s[{64 + 2 * k}:{65 + 2 * k}]")
            o("s_waitcnt lgkmcnt(0)")

    def step(g_cur: int, sb_c: int, a_sb: int | None, a_next_sb: int | None, c_extra: dict[int, list[str]] | None = None,
             early: list | None = None, late: list | None = None, late_from: int = 8, mid: list | None = None, mid_window: tuple[int, int] | None = None,
             mid_stamp: int | None = None, spread: tuple[list, int] | None = None) -> None:
        """[A(block a_sb of the tile in ROWA) with B(g_cur) in its MFMA gaps] ; C(block sb_c of the tile in TRA).  All fillers are placed by
        weave_budget (<= CAP issues per MFMA gap, the next group's wait + reads included):
          early : fillers for the first gaps (the per-tile TRA rebuild)         B : dS of generation g_cur; its DSF(., 0) half must be complete before
          C's first MFMA, its DSF(., 1) half before C's fifth iteration         mid : fillers with an explicit gap window (the ROWA rebuild)
          late  : fillers for C's gaps from iteration late_from / 2 on (LDS-DMA pieces).
        a_next_sb: block whose first two A fragments C's last iterations request (None: nothing follows)."""
        b = [] if "nob" in DBG else b_ops(g_cur)
        for flt, keep in (("onlyexp", lambda x: x.startswith("v_exp")), ("noexp", lambda x: not x.startswith("v_exp")), ("nocvt", lambda x: not x.startswith("v_cvt"))):
            if flt in DBG:
                b = [x for x in b if keep(x)]
        half = len(b) // 2
        tails = [a_request(0, a_next_sb), a_request(1, a_next_sb)] if a_next_sb is not None else [[], []]
        cg = c_groups(sb_c, tails, c_extra)
        if a_sb is None:                                  # drain step: nothing to hide B under
            st.extend(c_request(0, sb_c))
            st.extend(c_request(1, sb_c))
            st.extend(b)
            o("s_nop 1")                                  # VALU -> MFMA operand
            for g in cg:
                st.extend(g)
            return
        ag = a_groups(g_cur ^ 1, a_sb, [c_request(0, sb_c), c_request(1, sb_c)])
        groups = ag + cg
        na = len(ag)
        segs: list[tuple[list, int, int]] = []
        if spread:                                        # one chunk per window of `stride` gaps, placed before B claims the gaps
            chunks, stride = spread
            for k, ch in enumerate(chunks):
                segs.append(([ch], k * stride, min(k * stride + stride - 1, len(groups) - 1)))
        if early:
            segs.append((early, 0, max(0, na - 9)))
        segs.append((b[:half], 0, na - 1))                # DSF(., 0): before C's first MFMA (its head's wait + reads keep the two states to the MFMA)
        segs.append((b[half:], 0, na + 2 * NDT - 1))      # DSF(., 1): before C's iteration NDT (group na + 2 NDT)
        if mid:
            segs.append((mid, mid_window[0], mid_window[1]))
        if late:
            segs.append((late, na + late_from, len(groups) - 1))
        lines = weave_budget(groups, segs, CAP)
        if TRACE and mid_stamp is not None:               # stamp between A's last and C's first group
            cut = lines.index(cg[0][0]) if cg[0][0] in lines else None
            k = max(i for i, x in enumerate(lines) if x == ag[-1][-1])      # A's last MFMA
            lines = lines[:k + 1] + [f"s_memtime s[{64 + 2 * mid_stamp}:{65 + 2 * mid_stamp}]", "s_waitcnt lgkmcnt(0)"] + lines[k + 1:]
        st.extend(lines)

    if TRACE:
        o("s_memtime s[76:77]")
        o("s_waitcnt lgkmcnt(0)")
    st.comment("---- main loop: tiles 0 .. nkt-2")
    o(f"s_cmp_eq_u32 s{S_CNT}, 0")
    o("s_cbranch_scc1 .Ldq64_last_%=")
    o(".Ldq64_loop_%=:")
    st.comment("step 1: A(kt, 1) -> gen 1 | B(gen 0) ; C(kt, 0); barrier for tile kt+1 inside C, then the first fragments of A(kt+1, 0)")
    stamp(0)
    barrier = [] if "nobarrier" in DBG else ["s_waitcnt vmcnt(0)", "s_barrier"]
    # ROWA still serves A(kt, 1)'s requests up to A's group 20 (k-step 5 requests k-step 7); C's iteration 6 (group 44) requests from the next tile
    # ROWA serves A(kt, 1)'s requests up to k-step NKS - 3 (which requests the last fragment): group 4 (NKS - 3); C's iteration 2 NDT - 2 requests from the next tile
    step(0, 0, 1, 0, {2 * NDT - 3: barrier}, early=None, mid=rowa_update(S_NXT), mid_window=(4 * (NKS - 3) + 1, 4 * NKS + 2 * (2 * NDT - 2) - 2), mid_stamp=1)
    stamp(2)
    st.comment("step 2: A(kt+1, 0) -> gen 0 | B(gen 1) ; C(kt, 1) with the LDS-DMA of tile min(kt+2, nkt-1) in its last gaps; then the first fragments of A(kt+1, 1)")
    # pointer increments for after this stage: advance only while another tile exists (trips left >= 3  <=>  kt + 3 <= nkt - 1)
    o(f"s_cmp_ge_u32 s{S_CNT}, 3")
    o(f"s_cselect_b32 s{S_INCK}, {64 * HD * 2}, 0")
    o(f"s_cselect_b32 s{S_INCV}, %[vstep], 0")
    pcs = stage_pieces()
    dma = [] if "nostage" in DBG else [stage_begin() + pcs[0]] + pcs[1:]
    if os.environ.get("DQ64_DMA_IN_A"):                     # experiment: the pieces in A's gaps (spread: one piece every DQ64_DMA_IN_A gaps), B overflows into C instead
        stride = int(os.environ["DQ64_DMA_IN_A"])
        step(1, 1, 0, 1, None, early=None, mid=None, mid_stamp=3, spread=(dma, stride))
    else:
        step(1, 1, 0, 1, None, late=dma, late_from=int(os.environ.get("DQ64_DMA_FROM", str(min(4, 2 * NDT - 2)))), mid_stamp=3)
    stamp(4)
    st.extend(stage_advance(f"s{S_INCK}", f"s{S_INCV}"))
    st.extend(rotate_slots())
    st.extend(tra_update(S_CUR))                             # C(kt+1, .) reads its K^T fragments from the new current tile (first request: A's k-step 6 of the next step)
    o(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    o(f"s_cmp_lg_u32 s{S_CNT}, 0")
    o("s_cbranch_scc1 .Ldq64_loop_%=")
    o(".Ldq64_last_%=:")
    st.comment("---- last tile: A(last, 1) | B(gen 0) ; C(last, 0) ; B(gen 1) ; C(last, 1)")
    step(0, 0, 1, None)
    step(1, 1, None, None)
    if TRACE:
        o("s_memtime s[78:79]")
        o("s_waitcnt lgkmcnt(0)")
    st.comment("---- park dQ^T * scale as bf16 in the (idle) ring: token rows of 256 bytes, 16-byte chunk c of token t at c ^ (t & 15)  (rope_bwd_store's image)")
    o("s_nop 15")
    o("s_waitcnt vmcnt(0)")                                  # the (re-)staged last tile may still be landing in the ring
    o("s_barrier")                                           # every wave is past its last ring read
    # %[park] = lane part: l31*256 + ((l31 & 15) << 4) + 8*h ; wave slice (2 x 8 KiB) at lds + wave*16384
    o(f"s_lshl_b32 s{S_T0}, %[wvoff], 4")
    o(f"s_add_u32 s{S_T0}, s{S_T0}, %[lds]")
    o(f"v_add_u32_e32 {vr(V_ROWN)}, s{S_T0}, %[park]")
    for qb in range(2):
        for dt in range(NDT):
            for a in range(4):
                base = 16 * (NDT * qb + dt) + 4 * a
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
    if TRACE:
        o("s_memtime s[80:81]")
        o("s_waitcnt lgkmcnt(0)")
        o("s_cmp_lg_u32 %[blk0], 0")                          # %[blk0] = blockIdx.x | y | z
        o("s_cbranch_scc1 .Ldq64_notrace_%=")
        o(f"s_or_b32 s{S_T0}, %[tracelo], %[tracehi]")
        o(f"s_cmp_eq_u32 s{S_T0}, 0")
        o("s_cbranch_scc1 .Ldq64_notrace_%=")
        o(f"s_lshr_b32 s{S_T0}, %[wvoff], 3")                # wave * 128 bytes
        o(f"s_add_u32 s{S_KP}, %[tracelo], s{S_T0}")
        o(f"s_addc_u32 s{S_KP + 1}, %[tracehi], 0")
        o("s_mov_b64 exec, 1")
        o("v_mov_b32_e32 v56, 0")
        for k in range(9):
            o(f"v_mov_b32_e32 v58, s{64 + 2 * k}")
            o(f"v_mov_b32_e32 v59, s{65 + 2 * k}")
            o(f"global_store_dwordx2 v56, v[58:59], s[{S_KP}:{S_KP + 1}] offset:{8 * k}")
            o("s_nop 1")
        o("s_waitcnt vmcnt(0)")
        o("s_mov_b64 exec, -1")
        o(".Ldq64_notrace_%=:")
    lines = resolve_lgkm(st.lines, loop_label=".Ldq64_loop_%=:", loop_branch=None if TRACE else "s_cbranch_scc1 .Ldq64_loop_%=")     # (trace builds drain the queue at every stamp)
    bad = check_hazards(lines)
    if bad:
        raise SystemExit("hazard check failed:\n" + "\n".join(bad[:20]))
    return "\n".join('"' + ln.replace("\\", "\\\\") + '\\n"' for ln in lines) + "\n"


def main() -> None:
    name = "attn_dq64_body.inc" if HD == 128 else f"attn_dq64_hd{HD}_body.inc"
    out = os.environ.get("DQ64_OUT") or os.path.join(os.path.dirname(__file__), "..", "..", "simpletuner_amd", "csrc", "gen", name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    b_in_a = int(os.environ.get("DQ64_B_IN_A", "144"))
    body = build(b_in_a)
    with open(out, "w") as f:
        f.write("// GENERATED by tools/kgen/dq64.py — do not edit; regenerate with  python -m tools.kgen.dq64\n")
        f.write(body)
    if not os.environ.get("DQ64_OUT") and HD == 128:
        regs = [f'"v{i}"' for i in range(32, 256)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(40, 84)]
        with open(os.path.join(os.path.dirname(out), "attn_dq64_clobbers.inc"), "w") as f:
            f.write("// GENERATED by tools/kgen/dq64.py — the registers the dq64 body owns\n")
            f.write(",\n".join(", ".join(regs[i:i + 16]) for i in range(0, len(regs), 16)) + "\n")
    print(f"wrote {os.path.normpath(out)}: {body.count(chr(10))} lines", file=sys.stderr)


if __name__ == "__main__":
    main()
