"""kgen.fwd64 — instruction stream of k_attn_fwd64 (attention forward; head_dim 128; 4 waves x 64 queries, one wave per SIMD).

  python -m tools.kgen.fwd64        -> simpletuner_amd/csrc/gen/attn_fwd64_body.inc   (the text of one asm statement; see attention.hip)

Per 64-key tile t, both 32-query blocks qb of the wave (scores TRANSPOSED as in k_attn_fwd4: a lane owns one query column):
    A(t): S^T = K Q^T                      32 MFMAs   (2 key blocks sb x 8 k-steps x 2 qb; every K row fragment feeds both query blocks)
    B(t): online softmax, VALU: p = exp2(S), row sums, bf16 packing into PF(sb, qb, m);  and the tile maximum of tile t+1 (v_max3 chains + half-wave swap)
    C(t): O^T += V^T P^T                   32 MFMAs   (iterations v = (sb, m, dt); every V^T fragment feeds both query blocks)
Pipeline: step t = [A(t+1) with B(t) in its MFMA gaps] ; [C(t) with the rest of B(t) and the maximum of tile t+1 in its gaps] ; re-reference decision for t+1.
Two generations of S registers alternate.
The reference subtraction is folded into the MFMAs: every S accumulation chain STARTS from -m_ref / scale2 (srcC = a 16-register block holding it), so the
accumulators come out as  s - m_ref / scale2  and p = exp2(scale2 * acc): a multiply, an exponential, a row-sum add, half a pack and half a max3 per score.
The scores stay the exact fp32 sums of bf16 products every other attention kernel forms (lse2 agrees with k_attn_fwd4's to fp32 rounding).  FWD64_EXACT=0
builds the variant that also folds the scale into the Q fragments (multiplied by scale * log2(e) once per workgroup and re-rounded to bf16): one VALU less
per score, 2688 instead of 2892 cycles per tile — but the scores then move by ~2^-9 |s|, i.e. lse2 moves against the scores the BACKWARD kernels form
(1e-3 typical, 2e-2 on a spiked row: up to 1-4 % on that row's dominant P in dK / dV); not adopted.

The running maximum is a REFERENCE m_ref that may go stale: exponentials are taken against m_ref as long as no score of the tile exceeds it by more than
THR (log2 units; P then reaches 2^THR instead of 1 — same relative precision in bf16, numerator and denominator share the reference, lse2 = m_ref + log2(l)
is unchanged).  The O accumulators live in AGPRs, where a rescale costs three instructions per register: with a stale reference it happens in the first
tile and then almost never, in an out-of-line block behind a wave-uniform branch (cdna_hip_programming.md T13; the order here — decide before any
exponential of the tile is taken, rescale O, l together — is the textbook one).

LDS: ring of four 32-KiB slots [K image 64 keys x 256 B, swz_q chunk swizzle | V^T image 128 channels x 128 B, chunk ^ ((row >> 1) & 7)], tile t in slot
t % 4, staged THREE tiles ahead by LDS-DMA (8 pieces per wave and tile in C's gaps, right after the per-tile barrier).

Registers (the kernel lists v[32:255], a[0:255], s[40:83] as clobbers):
    a[0:127]    O^T accumulators OACC(qb, dt)          a[128:191] Q fragments QF(qb, ks) (pre-scaled)   a[192:203] K row fragments (3 deep)   a[204:215] V^T fragments (3 deep)
    v[64:127]   generation 0: S(sb, qb) at 64 + 32 sb + 16 qb        v[128:191] generation 1
    v[192:223]  P fragments PF(sb, qb, m)              v[224:239], v[240:255] -m_ref blocks NMB(qb) (16 equal registers each: the srcC of every S chain's first MFMA)
    v[32:39] ROWA[ks]    v[40:43] VTA[2 sb + m]   v[44:51] LDS-DMA lane offsets   v[52:61] scratch   v[62:63] l(qb)
"""
from __future__ import annotations

import os
import sys

from .emit import Stream, ar, check_hazards, resolve_lgkm, vr, weave_budget

MFMA = "v_mfma_f32_32x32x16_bf16"
TILE = 16384
SLOT = 2 * TILE
NSLOT = 4
THR_BITS = "0x41000000"     # 8.0f
HD = int(os.environ.get("FWD64_HD", "128"))      # head_dim: 128 (Flux) or 96 (PixArt-Sigma's 72, zero padded)
assert HD in (64, 96, 128)
# head_dim 96 is only ever a zero-padded narrower head (PixArt-Sigma 72, SD 1.5 80): the d-CONTRACTIONS (S^T = K Q^T) take 5 k-steps of 16 = 80 channels, the
# d-OUTPUT side (O^T tiles, the V^T image) keeps 96.  r5 lab, B1 H16 S16384: 1.407 -> 1.335 ms, O unchanged (profiles/r05_attn_lab_hd96_contraction_k_steps.log)
NKS, NDT = int(os.environ.get("FWD64_KS", {128: 8, 96: 5, 64: 4}[HD])), HD // 32                     # MFMA k-steps over the head dim; 32-row d tiles of O^T
NU, NV = 2 * NKS, 4 * NDT                         # K fragments (A groups) and V^T fragments (C iterations) per 64-key tile


def OACC(qb, dt): return ar(16 * (NDT * qb + dt), 16)
def QF(qb, ks): return ar(128 + 4 * (NKS * qb + ks), 4)
def S(g, sb, qb): return 64 + 64 * g + 32 * sb + 16 * qb
def PF(sb, qb, m): return 192 + 4 * (4 * sb + 2 * qb + m)
def KF(i): return ar(192 + 4 * (i % 3), 4)
def VF(i): return ar(204 + 4 * (i % 3), 4)
def NMB(qb): return 224 + 16 * qb          # 16 registers, all -m_ref of query block qb
NM = [NMB(0), NMB(1)]                      # (the first register of each block doubles as the scalar)
PK = os.environ.get("FWD64_PK", "0") != "0"     # packed fp32 VALU: one v_pk_mul_f32 scales two scores, one v_pk_add_f32 feeds both partial row sums (same operations
                                                # in the same order as the scalar forms: bit-identical output, 8 issues less per 8 scores)
L = [60, 62] if PK else [62, 63]             # running row sum per query block (this lane's half of the keys)
L2 = [61, 63] if PK else [60, 61]            # the second partial row sum (breaks the dependent-add chain); PK: the aligned pair v[L : L2]
S_SC2 = 54                                   # PK: s[54:55] = (scale2, scale2), the scalar operand of v_pk_mul_f32
ROWA = [32 + k for k in range(8)]
VTA = [40 + k for k in range(4)]
KOF = [44, 45, 46, 47]
VOF = [48, 49, 50, 51]
T = [52 + k for k in range(8)]         # scratch v52..v59 (v60..v63: the row sums L / L2)
# SGPRs
S_KP, S_VP = 40, 42
S_CNT = 44
S_SLOT = [45, 46, 47, 48]            # LDS byte addresses of the slots of tiles t, t+1, t+2 and the stage target (t+3)
S_M0, S_T0, S_T1, S_INCK, S_INCV = 49, 50, 51, 52, 53

TRACE = bool(os.environ.get("FWD64_TRACE"))
DBG = set(filter(None, os.environ.get("FWD64_DBG", "").split(",")))
CAP = float(os.environ.get("FWD64_CAP", {128: ("6" if os.environ.get("FWD64_EXACT", "1") != "0" else "5"), 96: "8.5", 64: "10"}[int(os.environ.get("FWD64_HD", "128"))]))      # issues per MFMA gap besides the MFMA
EXACT = os.environ.get("FWD64_EXACT", "1") != "0"    # scores as the exact fp32 sums of bf16 products (chains start from -m_ref / scale2, p = exp2(scale2 * acc));
                                                     # "0": Q pre-multiplied by scale2 and re-rounded (one VALU less per score, scores move by ~2^-9 |s|)


def k_request(u: int) -> list[str]:
    """K row fragment u = 8 sb + ks of the A tile (addresses in ROWA) into buffer u % 3"""
    sb, ks = divmod(u, NKS)
    return [f"ds_read_b128 {KF(u)}, {vr(ROWA[ks])} offset:{sb * 8192} ;@ld:K{u}"]


def v_request(v: int) -> list[str]:
    """V^T fragment v = 8 sb + 4 m + dt of the C tile (addresses in VTA) into buffer v % 3"""
    sb, m, dt = v // (2 * NDT), (v // NDT) & 1, v % NDT
    return [f"ds_read_b128 {VF(v)}, {vr(VTA[2 * sb + m])} offset:{TILE + dt * 4096} ;@ld:V{v}"]


def rowa_update(slot_sgpr: int) -> list[str]:
    return [f"v_add_u32_e32 {vr(ROWA[0])}, s{slot_sgpr}, %[rowb]"] + [f"v_xor_b32_e32 {vr(ROWA[k])}, {hex(k << 5)}, {vr(ROWA[0])}" for k in range(1, NKS)]


def vta_update(slot_sgpr: int) -> list[str]:
    # lane base %[vtb] = l31 * 128 + ((h ^ ((l31 >> 1) & 7)) << 4); fragment chunk (4 sb + 2 m + h): XOR (4 sb + 2 m) << 4
    return [f"v_add_u32_e32 {vr(VTA[0])}, s{slot_sgpr}, %[vtb]"] + [f"v_xor_b32_e32 {vr(VTA[j])}, {hex((2 * j) << 4)}, {vr(VTA[0])}" for j in range(1, 4)]


def a_groups(g_new: int, tail: list[list[str]], from_zero: bool = False) -> list[list[str]]:
    """A: 16 fragments u x 2 MFMAs.  Group u waits for fragment u and requests u + 2; u = 14, 15 carry tail[0 / 1] (C's first two fragments).
    Every accumulation chain starts from -m_ref (srcC = NMB(qb)); from_zero: the prologue's tile 0, whose reference is chosen afterwards."""
    groups = []
    for u in range(NU):
        sb, ks = divmod(u, NKS)
        head = [f"@wait:K{u}"] + (k_request(u + 2) if u < NU - 2 else tail[u - (NU - 2)])
        c = (lambda r, qb: ("0" if from_zero else vr(NMB(qb), 16))) if ks == 0 else (lambda r, qb: r)
        groups.append(head + [f"{MFMA} {vr(S(g_new, sb, 0), 16)}, {KF(u)}, {QF(0, ks)}, {c(vr(S(g_new, sb, 0), 16), 0)}"])
        groups.append([f"{MFMA} {vr(S(g_new, sb, 1), 16)}, {KF(u)}, {QF(1, ks)}, {c(vr(S(g_new, sb, 1), 16), 1)}"])
    return groups


def c_groups(tail: list[list[str]], extra_at: dict[int, list[str]] | None = None) -> list[list[str]]:
    """C: 16 fragments v = (sb, m, dt) x 2 MFMAs.  Iteration v waits for fragment v and requests v + 2; v = 14, 15 carry tail[0 / 1]
    (the next A's first two K fragments).  extra_at[v]: lines in front of iteration v's request."""
    extra_at = extra_at or {}
    groups = []
    for v in range(NV):
        sb, m, dt = v // (2 * NDT), (v // NDT) & 1, v % NDT
        head = [f"@wait:V{v}"] + extra_at.get(v, []) + (v_request(v + 2) if v < NV - 2 else tail[v - (NV - 2)])
        groups.append(head + [f"{MFMA} {OACC(0, dt)}, {VF(v)}, {vr(PF(sb, 0, m), 4)}, {OACC(0, dt)}"])
        groups.append([f"{MFMA} {OACC(1, dt)}, {VF(v)}, {vr(PF(sb, 1, m), 4)}, {OACC(1, dt)}"])
    return groups


def b_max(g: int, qb: int) -> list:
    """maximum over the 32 scores per lane of generation g, query block qb (already relative to the reference: s * scale2 - m_ref), both half-waves -> T[4 qb]"""
    t0, t1 = T[4 * qb], T[4 * qb + 1]
    s0, s1 = S(g, 0, qb), S(g, 1, qb)
    ops: list = []
    ops.append(f"v_max3_f32 {vr(t0)}, {vr(s0)}, {vr(s0 + 1)}, {vr(s0 + 2)}")          # two independent max3 chains (key blocks 0 and 1)
    ops.append(f"v_max3_f32 {vr(t1)}, {vr(s1)}, {vr(s1 + 1)}, {vr(s1 + 2)}")
    for i in range(3, 15, 2):
        ops.append(f"v_max3_f32 {vr(t0)}, {vr(t0)}, {vr(s0 + i)}, {vr(s0 + i + 1)}")
        ops.append(f"v_max3_f32 {vr(t1)}, {vr(t1)}, {vr(s1 + i)}, {vr(s1 + i + 1)}")
    ops.append(f"v_max3_f32 {vr(t0)}, {vr(t0)}, {vr(s0 + 15)}, {vr(t1)}")
    ops.append(f"v_max_f32_e32 {vr(t0)}, {vr(t0)}, {vr(s1 + 15)}")
    ops.append([f"v_mov_b32_e32 {vr(t1)}, {vr(t0)}", "s_nop 1", f"v_permlane32_swap_b32_e32 {vr(t0)}, {vr(t1)}"])     # lanes l and l ^ 32 share a query column
    ops.append(f"v_max_f32_e32 {vr(t0)}, {vr(t0)}, {vr(t1)}")
    return ops


def decide(g: int, qb: int, site: str) -> list[str]:
    """any lane whose tile maximum exceeds the reference by more than THR: re-reference in the out-of-line block (which returns to .Lback)"""
    thr = "%[thr]" if EXACT else THR_BITS                  # EXACT: the accumulators are in raw-score units: THR / scale2
    return [f"v_cmp_lt_f32_e32 vcc, {thr}, {vr(T[4 * qb])}", f"s_cbranch_vccnz .Lf64_resc_{site}_{qb}_%=", f".Lf64_back_{site}_{qb}_%=:"]


def rescale_ops(g: int, qb: int, first: bool) -> list[str]:
    """d = max(tile maximum - m_ref, 0) (first tile: the maximum itself): m_ref += d, i.e. NMB -= d; the scores of the tile (generation g, taken against the old
    reference) -= d; and unless this is the first tile: alpha = exp2(-d), l *= alpha, O(qb) *= alpha."""
    t0, t1, t3 = T[4 * qb], T[4 * qb + 1], T[4 * qb + 3]
    o = []
    if not first:
        o.append(f"v_max_f32_e32 {vr(t0)}, 0, {vr(t0)}")                  # lanes at or below their reference keep it
        if EXACT:
            o.append(f"v_mul_f32_e64 {vr(t1)}, -{vr(t0)}, %[scale2]")     # the accumulators are in raw-score units
        else:
            o.append(f"v_sub_f32_e32 {vr(t1)}, 0, {vr(t0)}")
        o.append(f"v_exp_f32_e32 {vr(t1)}, {vr(t1)}")                     # alpha
    o.append(f"v_sub_f32_e32 {vr(NMB(qb))}, {vr(NMB(qb))}, {vr(t0)}")
    for r in range(1, 16):
        o.append(f"v_mov_b32_e32 {vr(NMB(qb) + r)}, {vr(NMB(qb))}")
    for sb in range(2):
        for r in range(16):
            o.append(f"v_sub_f32_e32 {vr(S(g, sb, qb) + r)}, {vr(S(g, sb, qb) + r)}, {vr(t0)}")
    if not first:
        o.append(f"v_mul_f32_e32 {vr(L[qb])}, {vr(L[qb])}, {vr(t1)}")
        o.append(f"v_mul_f32_e32 {vr(L2[qb])}, {vr(L2[qb])}, {vr(t1)}")       # the second partial row sum lives at the old reference too
        o.append("s_nop 15")                                             # the last MFMAs of C own the O accumulators
        o.append("s_nop 15")
        for r in range(16 * NDT):
            a = 16 * NDT * qb + r
            o.append(f"v_accvgpr_read_b32 {vr(t3)}, a{a}")
            o.append(f"v_mul_f32_e32 {vr(t3)}, {vr(t3)}, {vr(t1)}")
            o.append(f"v_accvgpr_write_b32 a{a}, {vr(t3)}")
    o.append("s_nop 1")
    return o


def rescale_block(g: int, qb: int, site: str) -> list[str]:
    return [f".Lf64_resc_{site}_{qb}_%=:"] + rescale_ops(g, qb, False) + [f"s_branch .Lf64_back_{site}_{qb}_%="]


def b_exp(g: int, sb: int, qb: int, m: int) -> list[str]:
    """8 scores (registers 8 m .. 8 m + 7 of S(sb, qb), already s * scale2 - m_ref) -> p = exp2(.), row-sum partials, PF(sb, qb, m)"""
    rs = [S(g, sb, qb) + 8 * m + i for i in range(8)]
    ops = []
    for h4 in (0, 4):
        q = rs[h4:h4 + 4]
        if EXACT and PK:
            ops += [f"v_pk_mul_f32 {vr(q[i], 2)}, {vr(q[i], 2)}, s[{S_SC2}:{S_SC2 + 1}]" for i in (0, 2)]
        elif EXACT:
            ops += [f"v_mul_f32_e32 {vr(r)}, %[scale2], {vr(r)}" for r in q]
        ops += [f"v_exp_f32_e32 {vr(r)}, {vr(r)}" for r in q]
        # two partial sums per query block break the dependent-add chain: L[qb] and L2[qb]
        if PK:
            ops += [f"v_pk_add_f32 {vr(L[qb], 2)}, {vr(L[qb], 2)}, {vr(q[i], 2)}" for i in (0, 2)]
        else:
            ops += [f"v_add_f32_e32 {vr(L[qb])}, {vr(L[qb])}, {vr(q[0])}", f"v_add_f32_e32 {vr(L2[qb])}, {vr(L2[qb])}, {vr(q[1])}",
                    f"v_add_f32_e32 {vr(L[qb])}, {vr(L[qb])}, {vr(q[2])}", f"v_add_f32_e32 {vr(L2[qb])}, {vr(L2[qb])}, {vr(q[3])}"]
        ops += [f"v_cvt_pk_bf16_f32 {vr(PF(sb, qb, m) + (h4 >> 1) + i)}, {vr(q[2 * i])}, {vr(q[2 * i + 1])}" for i in range(2)]
    return ops


def stage_pieces() -> list[list[str]]:
    pcs = []
    for p in range(4):
        pcs.append([f"s_add_u32 m0, s{S_T0}, {p * 4096}", "s_nop 0", f"global_load_lds_dwordx4 {vr(KOF[p])}, s[{S_KP}:{S_KP + 1}]"])
    for p in range(NDT):       # the V^T image holds head_dim rows of 128 bytes: head_dim / 8 pieces, head_dim / 32 per wave
        pcs.append([f"s_add_u32 m0, s{S_T0}, {p * 4096 + TILE}", "s_nop 0", f"global_load_lds_dwordx4 {vr(VOF[p])}, s[{S_VP}:{S_VP + 1}]"])
    return pcs


def stage_begin(slot: int) -> list[str]:
    return [f"s_add_u32 s{S_T0}, s{slot}, %[wvoff]"]


def stage_advance(inck: str, incv: str) -> list[str]:
    return [f"s_add_u32 s{S_KP}, s{S_KP}, {inck}", f"s_addc_u32 s{S_KP + 1}, s{S_KP + 1}, 0",
            f"s_add_u32 s{S_VP}, s{S_VP}, {incv}", f"s_addc_u32 s{S_VP + 1}, s{S_VP + 1}, 0"]


def rotate_slots() -> list[str]:
    """(t, t+1, t+2, stage) <- (t+1, t+2, stage, t)"""
    a, b, c, d = S_SLOT
    return [f"s_mov_b32 s{S_T1}, s{a}", f"s_mov_b32 s{a}, s{b}", f"s_mov_b32 s{b}, s{c}", f"s_mov_b32 s{c}, s{d}", f"s_mov_b32 s{d}, s{S_T1}"]


def build() -> str:
    st = Stream()
    o = st.op
    out_of_line: list[str] = []
    st.comment("---- prologue")
    o(f"s_mov_b32 s{S_M0}, m0")
    if PK:
        o(f"s_mov_b32 s{S_SC2}, %[scale2]")
        o(f"s_mov_b32 s{S_SC2 + 1}, %[scale2]")
    if EXACT:
        for qb in range(2):
            for ks in range(NKS):
                o(f"global_load_dwordx4 {QF(qb, ks)}, %[qp{qb}], off offset:{32 * ks}")
    else:        # Q fragments -> v[128:191] -> * scale2 (fp32 multiply, one rounding to bf16) -> a[128:191]
        for qb in range(2):
            for ks in range(NKS):
                o(f"global_load_dwordx4 {vr(128 + 4 * (NKS * qb + ks), 4)}, %[qp{qb}], off offset:{32 * ks}")
    for i in range(2 * NDT * 16):
        o(f"v_accvgpr_write_b32 a{i}, 0")
    for qb in range(2):
        for r in range(16):
            o(f"v_mov_b32_e32 {vr(NMB(qb) + r)}, 0")        # the reference is chosen after tile 0 (whose chains start from 0)
        o(f"v_mov_b32_e32 {vr(L[qb])}, 0")
        o(f"v_mov_b32_e32 {vr(L2[qb])}, 0")
    if not EXACT:
        o("s_waitcnt vmcnt(0)")
        for i in range(8 * NKS):
            src, lo, hi = 128 + i, T[0], T[1]
            o(f"v_lshlrev_b32_e32 {vr(lo)}, 16, {vr(src)}")
            o(f"v_and_b32_e32 {vr(hi)}, 0xffff0000, {vr(src)}")
            o(f"v_mul_f32_e32 {vr(lo)}, %[scale2], {vr(lo)}")
            o(f"v_mul_f32_e32 {vr(hi)}, %[scale2], {vr(hi)}")
            o(f"v_cvt_pk_bf16_f32 {vr(lo)}, {vr(lo)}, {vr(hi)}")
            o(f"v_accvgpr_write_b32 a{128 + i}, {vr(lo)}")
    o(f"v_mov_b32_e32 {vr(KOF[0])}, %[koff]")
    o(f"v_mov_b32_e32 {vr(VOF[0])}, %[voff]")
    for p in range(1, 4):
        o(f"v_add_u32_e32 {vr(KOF[p])}, {p * 16 * HD * 2}, {vr(KOF[0])}")
        o(f"v_add_u32_e32 {vr(VOF[p])}, %[vrow32], {vr(VOF[p - 1])}")
    o(f"s_mov_b64 s[{S_KP}:{S_KP + 1}], %[kbase]")
    o(f"s_mov_b64 s[{S_VP}:{S_VP + 1}], %[vbase]")
    for k in range(NSLOT):
        o(f"s_add_u32 s{S_SLOT[k]}, %[lds], {k * SLOT}")
    # tiles 0, 1, 2 -> slots 0, 1, 2 (as far as they exist); afterwards S_KP / S_VP point at tile min(3, nkt - 1)
    for k in range(3):
        st.extend(stage_begin(S_SLOT[k]))
        for pc in stage_pieces():
            st.extend(pc)
        o(f"s_cmp_lt_u32 %[nkt], {k + 2}")
        o("s_cbranch_scc1 .Lf64_staged_%=")
        st.extend(stage_advance(str(64 * HD * 2), "128"))
    o(".Lf64_staged_%=:")
    o(f"s_sub_u32 s{S_CNT}, %[nkt], 1")                      # full steps (with an A for the next tile): tiles 0 .. nkt-2
    o("s_waitcnt vmcnt(0)")
    o("s_barrier")
    st.comment("---- A(tile 0) -> generation 0; then the first two K fragments of A(tile 1)")
    st.extend(rowa_update(S_SLOT[0]))
    st.extend(vta_update(S_SLOT[0]))
    st.extend(k_request(0))
    st.extend(k_request(1))
    for g in a_groups(0, [[], []], from_zero=True):
        st.extend(g)
    st.extend(rowa_update(S_SLOT[1]))
    st.extend(k_request(0))
    st.extend(k_request(1))
    o("s_nop 7")
    st.comment("---- the first reference: m_ref = maximum of tile 0 per query column (O and l are still zero: nothing to rescale)")
    for qb in range(2):
        for item in b_max(0, qb):
            st.extend([item] if isinstance(item, str) else item)
        st.extend(rescale_ops(0, qb, True))

    def stamp(k: int) -> None:
        if TRACE:
            o(f"s_memtime s[{64 + 2 * k}:{65 + 2 * k}]")
            o("s_waitcnt lgkmcnt(0)")

    def step(g_cur: int, site: str, full: bool, loop_body: bool) -> None:
        """full: [A(t+1) -> generation g_cur ^ 1 | B(t)] ; C(t) with barrier, LDS-DMA of tile min(t+3, nkt-1), address rebuilds, the maximum of tile t+1 in its
        gaps ; the re-reference decision for tile t+1 (so that A(t+2) already starts from the new reference).   not full: B(t) ; C(t)."""
        nob = "nob" in DBG
        chunks = {(sb, m): ([] if nob else b_exp(g_cur, sb, 0, m) + b_exp(g_cur, sb, 1, m)) for sb in range(2) for m in range(2)}
        barrier = [] if "nobarrier" in DBG else ["s_waitcnt vmcnt(0)", "s_barrier"]
        if not full:
            st.extend(vta_update(S_SLOT[0]))
            st.extend(v_request(0))
            st.extend(v_request(1))
            for key in ((0, 0), (0, 1), (1, 0), (1, 1)):
                st.extend(chunks[key])
            o("s_nop 1")
            for g in c_groups([[], []]):
                st.extend(g)
            return
        ag = a_groups(g_cur ^ 1, [v_request(0), v_request(1)])
        # C: barrier for tile t+2 at iteration 1 (every wave is then past its reads of tile t-1, whose slot the DMA below refills), then the DMA pieces;
        # iterations 14 / 15 request the first K fragments of A(t+2) from the tile the barrier has just covered
        cg = c_groups([k_request(0), k_request(1)], {1: barrier})
        groups = ag + cg
        na = len(ag)
        pcs = stage_pieces()
        dma = [] if "nostage" in DBG else [stage_begin(S_SLOT[3]) + pcs[0]] + pcs[1:]
        bm = [] if nob else b_max(g_cur ^ 1, 0) + b_max(g_cur ^ 1, 1)
        segs: list[tuple] = []
        segs.append((vta_update(S_SLOT[0]), 0, 2 * (NU - 2) - 4))          # C(t)'s V^T addresses (first request: A's fragment NU - 2, group 2 (NU - 2))
        segs.append((chunks[(0, 0)], 0, na - 1, "chain"))                  # PF(0, ., 0): before C's first MFMA
        segs.append((chunks[(0, 1)], 0, na + 2 * NDT - 1, "chain"))        # PF(0, ., 1): before C's iteration NDT (group na + 2 NDT)
        segs.append((chunks[(1, 0)], 0, na + 4 * NDT - 1, "chain"))
        segs.append((chunks[(1, 1)], 0, na + 6 * NDT - 1, "chain"))
        segs.append((rowa_update(S_SLOT[2]), 2 * (NU - 3) + 1, na + 2 * (NV - 2) - 2))    # K addresses of tile t+2: after A's last request, before C's iteration NV - 2
        segs.append((dma, na + 3, len(groups) - 1))
        segs.append((bm, na + 8, len(groups) - 1))                         # tile t+1's scores are complete 12 states after A's last MFMA (group na - 1)
        lines = weave_budget(groups, segs, CAP)
        if TRACE and site == "l0":
            k = max(i for i, x in enumerate(lines) if x == ag[-1][-1])
            lines = lines[:k + 1] + ["s_memtime s[66:67]", "s_waitcnt lgkmcnt(0)"] + lines[k + 1:]
        st.extend(lines)
        if not nob:
            for qb in range(2):
                st.extend(decide(g_cur ^ 1, qb, site))
                out_of_line.extend(rescale_block(g_cur ^ 1, qb, site))
        st.extend(stage_advance(f"s{S_INCK}", f"s{S_INCV}"))
        st.extend(rotate_slots())

    def inc_select() -> None:
        # pointer increments after this step's stage (tile min(t+3, nkt-1)): advance while tile t+4 exists  <=>  full steps left (incl. this one) >= 4
        o(f"s_cmp_ge_u32 s{S_CNT}, 4")
        o(f"s_cselect_b32 s{S_INCK}, {64 * HD * 2}, 0")
        o(f"s_cselect_b32 s{S_INCV}, 128, 0")

    st.comment("---- main loop: two full steps (tiles t, t+1) per trip while at least two full steps are left")
    o(f"s_cmp_lt_u32 s{S_CNT}, 2")
    o("s_cbranch_scc1 .Lf64_tail_%=")
    o(".Lf64_loop_%=:")
    stamp(0)
    inc_select()
    step(0, "l0", True, True)
    stamp(2)
    o(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    inc_select()
    step(1, "l1", True, True)
    o(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    o(f"s_cmp_ge_u32 s{S_CNT}, 2")
    o("s_cbranch_scc1 .Lf64_loop_%=")
    o(".Lf64_tail_%=:")
    st.comment("---- tail: one more full step if one is left, then the last tile without an A")
    o(f"s_cmp_eq_u32 s{S_CNT}, 0")
    o("s_cbranch_scc1 .Lf64_last0_%=")
    inc_select()
    step(0, "t0", True, False)
    step(1, "t1", False, False)
    o("s_branch .Lf64_done_%=")
    o(".Lf64_last0_%=:")
    step(0, "t2", False, False)
    o(".Lf64_done_%=:")
    st.comment("---- l = both partial sums, both half-waves; park O^T / l as bf16 token rows in the (idle) ring; lse2 = m_ref + log2(l)")
    o("s_nop 15")
    o("s_waitcnt vmcnt(0)")
    o("s_barrier")
    o(f"s_lshl_b32 s{S_T0}, %[wvoff], 4")
    o(f"s_add_u32 s{S_T0}, s{S_T0}, %[lds]")
    o(f"v_add_u32_e32 {vr(T[0])}, s{S_T0}, %[park]")
    for qb in range(2):
        o(f"v_add_f32_e32 {vr(L[qb])}, {vr(L[qb])}, {vr(L2[qb])}")
        o(f"v_mov_b32_e32 {vr(T[1])}, {vr(L[qb])}")
        o("s_nop 1")
        o(f"v_permlane32_swap_b32_e32 {vr(L[qb])}, {vr(T[1])}")
        o(f"v_add_f32_e32 {vr(L[qb])}, {vr(L[qb])}, {vr(T[1])}")           # l_tot in every lane of the column
        o(f"v_log_f32_e32 {vr(T[2])}, {vr(L[qb])}")
        o(f"v_rcp_f32_e32 {vr(T[3])}, {vr(L[qb])}")
        o("s_nop 0")
        if EXACT:
            o(f"v_fma_f32 {vr(T[4 + qb])}, -{vr(NM[qb])}, %[scale2], {vr(T[2])}")     # lse2 = log2(l) + m_ref,  m_ref = -NM * scale2
        else:
            o(f"v_sub_f32_e32 {vr(T[4 + qb])}, {vr(T[2])}, {vr(NM[qb])}")      # lse2 = log2(l) + m_ref   -> output operand copy below
        for dt in range(NDT):
            for a in range(4):
                base = 16 * (NDT * qb + dt) + 4 * a
                t = 64 + 4 * ((4 * dt + a) & 3)
                for bb in range(4):
                    o(f"v_accvgpr_read_b32 {vr(t + bb)}, a{base + bb}")
                for bb in range(4):
                    o(f"v_mul_f32_e32 {vr(t + bb)}, {vr(T[3])}, {vr(t + bb)}")
                o(f"v_cvt_pk_bf16_f32 {vr(t)}, {vr(t)}, {vr(t + 1)}")
                o(f"v_cvt_pk_bf16_f32 {vr(t + 1)}, {vr(t + 2)}, {vr(t + 3)}")
                ch = 4 * dt + a
                o(f"v_xor_b32_e32 {vr(T[1])}, {hex(ch << 4)}, {vr(T[0])}" if ch else f"v_mov_b32_e32 {vr(T[1])}, {vr(T[0])}")
                o(f"ds_write_b64 {vr(T[1])}, {vr(t, 2)} offset:{qb * 8192}")
    o(f"v_mov_b32_e32 %[lse0], {vr(T[4])}")
    o(f"v_mov_b32_e32 %[lse1], {vr(T[5])}")
    o("s_waitcnt lgkmcnt(0)")
    o(f"s_mov_b32 m0, s{S_M0}")
    if TRACE:
        o("s_cmp_lg_u32 %[blk0], 0")
        o("s_cbranch_scc1 .Lf64_notrace_%=")
        o(f"s_or_b32 s{S_T0}, %[tracelo], %[tracehi]")
        o(f"s_cmp_eq_u32 s{S_T0}, 0")
        o("s_cbranch_scc1 .Lf64_notrace_%=")
        o(f"s_lshr_b32 s{S_T0}, %[wvoff], 3")
        o(f"s_add_u32 s{S_KP}, %[tracelo], s{S_T0}")
        o(f"s_addc_u32 s{S_KP + 1}, %[tracehi], 0")
        o("s_mov_b64 exec, 1")
        o("v_mov_b32_e32 v56, 0")
        for k in range(3):
            o(f"v_mov_b32_e32 v58, s{64 + 2 * k}")
            o(f"v_mov_b32_e32 v59, s{65 + 2 * k}")
            o(f"global_store_dwordx2 v56, v[58:59], s[{S_KP}:{S_KP + 1}] offset:{8 * k}")
            o("s_nop 1")
        o("s_waitcnt vmcnt(0)")
        o("s_mov_b64 exec, -1")
        o(".Lf64_notrace_%=:")
    o("s_branch .Lf64_end_%=")
    st.extend(out_of_line)
    o(".Lf64_end_%=:")
    lines = resolve_lgkm(st.lines, loop_label=".Lf64_loop_%=:", loop_branch=None if TRACE else "s_cbranch_scc1 .Lf64_loop_%=")
    bad = check_hazards(lines)
    if bad:
        raise SystemExit("hazard check failed:\n" + "\n".join(bad[:20]))
    return "\n".join('"' + ln.replace("\\", "\\\\") + '\\n"' for ln in lines) + "\n"


def main() -> None:
    name = "attn_fwd64_body.inc" if HD == 128 else f"attn_fwd64_hd{HD}_body.inc"
    out = os.environ.get("FWD64_OUT") or os.path.join(os.path.dirname(__file__), "..", "..", "simpletuner_amd", "csrc", "gen", name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    body = build()
    with open(out, "w") as f:
        f.write("// GENERATED by tools/kgen/fwd64.py — do not edit; regenerate with  python -m tools.kgen.fwd64\n")
        f.write(body)
    if not os.environ.get("FWD64_OUT") and HD == 128:
        regs = [f'"v{i}"' for i in range(32, 256)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(40, 84)]
        with open(os.path.join(os.path.dirname(out), "attn_fwd64_clobbers.inc"), "w") as f:
            f.write("// GENERATED by tools/kgen/fwd64.py — the registers the fwd64 body owns\n")
            f.write(",\n".join(", ".join(regs[i:i + 16]) for i in range(0, len(regs), 16)) + "\n")
    print(f"wrote {os.path.normpath(out)}: {body.count(chr(10))} lines", file=sys.stderr)


if __name__ == "__main__":
    main()
