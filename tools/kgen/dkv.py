"""kgen.dkv — instruction stream of k_attn_bwd_dkv4 (attention backward, dK / dV; head_dim 128; 8 waves x 32 keys, TWO waves per SIMD, 256 registers per wave).

  python -m tools.kgen.dkv        -> simpletuner_amd/csrc/gen/attn_dkv4_body.inc   (the text of one asm statement; see attention_bwd.hip)

Why a hand-scheduled body at the SAME geometry as k_attn_bwd_dkv3 (64 keys per wave would need 256 accumulator registers + 128 operand registers: no
budget holds it): dkv3 sits at the 256-register limit, so hipcc consumes its LDS fragments in pairs right behind their requests (sched_barrier pins) and
spends five VALU instructions per score; the matrix pipe idles ~45 % of the time although the LDS pipe is only half used (256 B/clk/CU for b64 / b128
reads, MI355X_MICROARCH.md LDS table).  Here
  * both VALU subtractions ride in the MFMAs: every S chain STARTS from lse / scale2 (the prep kernel writes the padded statistics row pre-divided; the
    accumulator block is loaded from the stats image with four ds_read_b128 instead of being zeroed) and accumulates  q . (-k):  acc = lse / scale2 - s,
    p = exp2(-scale2 * acc);  every dP chain starts from +delta and accumulates  dO . (-v):  acc = delta - dP,  dS = p * (-acc).  K and V are NEGATED
    (exact: the scores are the same fp32 sums of bf16 products the other kernels form; a first version pre-multiplied K by -scale2 and re-rounded it, which
    moved extreme scores by ~2^-9 |s| against the forward's lse — up to 4 % on the dominant P of a spiked row).  Four VALU instructions per score (mul,
    exp, mul, two half packs) instead of five, and no registers for the statistics;
  * P and dS are packed IN PLACE into the first halves of the S / dP accumulator blocks;
  * operand fragments are requested two MFMAs ahead into a ring of four 4-register buffers (counted lgkmcnt), C's first two during B, the next block's
    statistics behind C's last MFMA.
Per 32-query block qb of a 64-query tile:  A: S, dP (16 MFMAs, alternating chains)   B: p, dS (VALU)   C: dV^T += dO^T P, dK^T += Q^T dS (16 MFMAs).
The partner wave on the SIMD fills the matrix pipe while this one is in B.

LDS (as dkv3): two slots [Q image 64 x 256 B | dO image | lse 64 f32 | delta 64 f32], tile t in slot t & 1 (the stream is unrolled over the slot pair so that
every LDS address is lane base + immediate), LDS-DMA one tile ahead, one drained barrier per tile.  After the loop dK * scale and dV are parked as bf16
token rows (2 x 8 KiB per wave) for the HIP epilogue (fused RoPE + RMSNorm backward for dK, row stores for dV).

Registers: a[0:63] dK^T(dt)  a[64:127] dV^T(dt)  |  v[32:63] K' fragments  v[64:95] V' fragments  v[96:111] S block  v[112:127] dP block
           v[16:31] fragment ring (4 x 4)  v[10:15] scratch.   (the kernel lists v[10:127], a[0:127], s[40:75] as clobbers)
"""
from __future__ import annotations

import os
import sys

from .emit import Stream, ar, check_hazards, resolve_lgkm, vr, weave_budget

MFMA = "v_mfma_f32_32x32x16_bf16"
QT = 16384                 # one tile image (64 queries x 256 B)
STAT = 0                   # slot layout [lse 64 f32 | delta 64 f32 | Q image | dO image]: every ds_read immediate (slot + image + block + row) stays below 64 KiB
QOFF = 512
GOFF = 512 + QT
BUF = 2 * QT + 512         # one slot
HD = int(os.environ.get("DKV_HD", "128"))        # head_dim: 128 (Flux) or 96 (PixArt-Sigma's 72, zero padded); tile images keep the 256-byte row pitch
assert HD in (64, 96, 128)
# head_dim 96 = a zero-padded narrower head (<= 80 valid channels): S and dP contract over 5 k-steps (80 channels); dK^T / dV^T keep their 3 d tiles.  r5 lab, B1 H16
# S16384: 2.562 -> 2.438 ms, dK / dV unchanged against the 32-row kernel (3.8e-4 / 4.1e-4, the same figures as the 6-k-step body)
NKS, NDT = int(os.environ.get("DKV_KS", {128: 8, 96: 5, 64: 4}[HD])), HD // 32


def DK(dt): return ar(16 * dt, 16)
def DV(dt): return ar(16 * NDT + 16 * dt, 16)
def KFR(ks): return vr(32 + 4 * ks, 4)
def VFR(ks): return vr(64 + 4 * ks, 4)
SACC, DPACC = 96, 112
def RING(i, ph=0): return 16 + 4 * ((i + ph) % 4)     # four buffers.  ph = the ring position at which the phase (A or C of a block) starts: with an even k-step count
                                                       # every phase advances the ring by a multiple of four (ph stays 0); 5 k-steps (head_dim 96) advance A by 10, so the
                                                       # position is carried from phase to phase (PHASE below) — fragment i + 2 must never land in the buffer fragment i is read from
PHASE = [0]                                            # ring position of the A phase of the block being emitted
T = [10, 11, None, None, 12, 13, 14, 15]     # scratch v10..v15 (T[0], T[1]: DMA / park; T[4], T[5]: transposed addresses; T[6], T[7]: row addresses)
# SGPRs
S_QP, S_GP, S_LP, S_DP = 40, 42, 44, 46      # global bases: Q head, dO head, lse row, delta row (64-bit)
S_CNT, S_QQ0, S_M0, S_T0, S_T1, S_T2, S_SQ1, S_STQ = 48, 49, 50, 51, 52, 53, 54, 55

CAP = float(os.environ.get("DKV_CAP", "6"))
DBG = set(filter(None, os.environ.get("DKV_DBG", "").split(",")))


def row_request(i: int, slot: int, qb: int, ph: int = 0) -> list[str]:
    """A-phase fragment i = 2 ks + which (0: Q row fragment, 1: dO row fragment) of query block qb into ring buffer i"""
    ks, which = i >> 1, i & 1
    x = ks << 5
    t = T[6 + (i & 1)]
    out = [f"v_xor_b32_e32 {vr(t)}, {hex(x)}, %[rowb]"] if x else []
    addr = t if x else None
    a = vr(addr) if addr is not None else "%[rowb]"
    out.append(f"ds_read_b128 {vr(RING(i, ph), 4)}, {a} offset:{slot * BUF + (GOFF if which else QOFF) + qb * 8192} ;@ld:R{i}")
    return out


def tr_request(i: int, slot: int, qb: int, ph: int = 0) -> list[str]:
    """C-phase fragment i = 4 dt + 2 m + which (0: dO^T fragment -> dV, 1: Q^T fragment -> dK) of query block qb: two transposing reads into ring buffer i"""
    dt, m, which = i >> 2, (i >> 1) & 1, i & 1
    x0, x1 = (4 * dt) << 4, ((4 * dt) ^ 1) << 4
    t0, t1 = T[4], T[5]
    base = slot * BUF + (QOFF if which else GOFF) + qb * 8192 + m * 4096
    out = []
    if which == 0:           # the Q^T fragment of the same (dt, m) follows with the same two addresses
        out.append(f"v_xor_b32_e32 {vr(t0)}, {hex(x0)}, %[trb]" if x0 else f"v_mov_b32_e32 {vr(t0)}, %[trb]")
        out.append(f"v_xor_b32_e32 {vr(t1)}, {hex(x1)}, %[trb]")
    r = RING(i, ph)
    out.append(f"ds_read_b64_tr_b16 {vr(r, 2)}, {vr(t0)} offset:{base} ;@ld:C{i}")
    out.append(f"ds_read_b64_tr_b16 {vr(r + 2, 2)}, {vr(t1)} offset:{base + 1024} ;@ld:C{i}")
    return out


def stat_request(slot: int, qb: int) -> list[str]:
    """S block <- +lse, dP block <- +delta of the lane's 16 accumulator rows (queries 32 qb + 16 (r >> 3) + 8 h + (r & 7)): four 16-byte reads each"""
    out = []
    for which, acc in ((0, SACC), (1, DPACC)):
        for m in range(2):
            for q4 in range(2):
                off = slot * BUF + STAT + which * 256 + (32 * qb + 16 * m + 4 * q4) * 4
                out.append(f"ds_read_b128 {vr(acc + 8 * m + 4 * q4, 4)}, %[statb] offset:{off} ;@ld:ST")
    return out


def a_groups(slot: int, qb: int, tail: list[list[str]], ph: int = 0) -> list[list[str]]:
    """16 MFMAs: S += Q_frag x K'(ks), dP += dO_frag x V'(ks), alternating.  Group i waits for fragment i and requests i + 2; the last two carry tail[0 / 1]."""
    groups = []
    for i in range(2 * NKS):
        ks, which = i >> 1, i & 1
        head = [f"@wait:R{i}"] + (row_request(i + 2, slot, qb, ph) if i < 2 * NKS - 2 else tail[i - (2 * NKS - 2)])
        if i == 0:
            head = ["@wait:ST"] + head
        acc = DPACC if which else SACC
        opb = VFR(ks) if which else KFR(ks)
        groups.append(head + [f"{MFMA} {vr(acc, 16)}, {vr(RING(i, ph), 4)}, {opb}, {vr(acc, 16)}"])
    return groups


def b_ops() -> list[str]:
    """p = exp2(-scale2 * (lse / scale2 - s)) ; dS = p * -(delta - dP) ; P -> bf16 pairs in S[0:7], dS -> bf16 pairs in dP[0:7] (in place, ascending)"""
    ops = []
    if "nob" in DBG:
        return ops
    for r0 in range(0, 16, 4):
        rs = [SACC + r0 + i for i in range(4)]
        ps = [DPACC + r0 + i for i in range(4)]
        ops += [f"v_mul_f32_e32 {vr(r)}, %[nscale2], {vr(r)}" for r in rs]       # -scale2 * (lse / scale2 - s) = s * scale2 - lse
        ops += [f"v_exp_f32_e32 {vr(r)}, {vr(r)}" for r in rs]
        ops += [f"v_mul_f32_e64 {vr(p)}, {vr(r)}, -{vr(p)}" for r, p in zip(rs, ps)]
        ops += [f"v_cvt_pk_bf16_f32 {vr(SACC + (r0 >> 1) + i)}, {vr(rs[2 * i])}, {vr(rs[2 * i + 1])}" for i in range(2)]
        ops += [f"v_cvt_pk_bf16_f32 {vr(DPACC + (r0 >> 1) + i)}, {vr(ps[2 * i])}, {vr(ps[2 * i + 1])}" for i in range(2)]
    return ops


def c_groups(slot: int, qb: int, tail: list[list[str]], ph: int = 0) -> list[list[str]]:
    """16 MFMAs: dV^T(dt) += dO^T_frag x P(m), dK^T(dt) += Q^T_frag x dS(m).  Group i waits for fragment i and requests i + 2; the last two carry tail."""
    groups = []
    for i in range(4 * NDT):
        dt, m, which = i >> 2, (i >> 1) & 1, i & 1
        head = [f"@wait:C{i}"] + (tr_request(i + 2, slot, qb, ph) if i < 4 * NDT - 2 else tail[i - (4 * NDT - 2)])
        if which == 0:
            groups.append(head + [f"{MFMA} {DV(dt)}, {vr(RING(i, ph), 4)}, {vr(SACC + 4 * m, 4)}, {DV(dt)}"])
        else:
            groups.append(head + [f"{MFMA} {DK(dt)}, {vr(RING(i, ph), 4)}, {vr(DPACC + 4 * m, 4)}, {DK(dt)}"])
    return groups


def stage_ops(slot: int) -> list:
    """LDS-DMA of tile min(qt + 1, last) into `slot`: this wave's two Q pieces and two dO pieces (rows clamped to Sq - 1), waves 0 / 1 also the lse / delta row.
    S_QQ0 = first query of that tile, S_STQ = the same clamped for the padded statistics rows.  Returned as chunks for the weaver."""
    ch: list = []
    for p in range(2):
        t0, t1 = T[0], T[1]
        src = f"s{S_QQ0}" if p == 0 else f"s{S_T2}"           # S_T2 = S_QQ0 + 32: the wave's second piece starts 32 rows further down
        ch.append(([f"s_add_u32 s{S_T2}, s{S_QQ0}, 32"] if p else []) + [f"v_add_u32_e32 {vr(t0)}, {src}, %[drow]", f"v_min_u32_e32 {vr(t0)}, s{S_SQ1}, {vr(t0)}"])
        ch.append([f"v_mad_u32_u24 {vr(t1)}, {vr(t0)}, %[q2], %[dcol]",            # Q rows are head_dim * 2 bytes apart
                   f"s_add_u32 m0, s{S_T0}, {slot * BUF + QOFF + p * 8192}", "s_nop 0", f"global_load_lds_dwordx4 {vr(t1)}, s[{S_QP}:{S_QP + 1}]"])
        ch.append([f"v_mad_u32_u24 {vr(t1)}, {vr(t0)}, %[ldo2], %[dcol]",
                   f"s_add_u32 m0, s{S_T0}, {slot * BUF + GOFF + p * 8192}", "s_nop 0", f"global_load_lds_dwordx4 {vr(t1)}, s[{S_GP}:{S_GP + 1}]"])
    return ch


_uniq = [0]


def stat_stage(slot: int) -> list[str]:
    """waves 0 and 1: the 64 lse / delta values of the staged tile (256 bytes each) by one 4-byte LDS-DMA per lane"""
    _uniq[0] += 1
    L0, L1, LE = (f".Ldkv_st{_uniq[0]}_{k}_%=" for k in ("w0", "w1", "e"))
    return [f"s_cmp_eq_u32 %[wv], 0", f"s_cbranch_scc1 {L0}", f"s_cmp_eq_u32 %[wv], 1", f"s_cbranch_scc1 {L1}", f"s_branch {LE}",
            f"{L0}:", f"v_mbcnt_lo_u32_b32 {vr(T[0])}, -1, 0", f"v_mbcnt_hi_u32_b32 {vr(T[0])}, -1, {vr(T[0])}", f"v_add_lshl_u32 {vr(T[0])}, {vr(T[0])}, s{S_STQ}, 2", f"s_add_u32 m0, %[lds], {slot * BUF + STAT}", "s_nop 0",
            f"global_load_lds_dword {vr(T[0])}, s[{S_LP}:{S_LP + 1}]", f"s_branch {LE}",
            f"{L1}:", f"v_mbcnt_lo_u32_b32 {vr(T[0])}, -1, 0", f"v_mbcnt_hi_u32_b32 {vr(T[0])}, -1, {vr(T[0])}", f"v_add_lshl_u32 {vr(T[0])}, {vr(T[0])}, s{S_STQ}, 2", f"s_add_u32 m0, %[lds], {slot * BUF + STAT + 256}", "s_nop 0",
            f"global_load_lds_dword {vr(T[0])}, s[{S_DP}:{S_DP + 1}]", f"{LE}:"]


def build() -> str:
    st = Stream()
    o = st.op
    st.comment("---- prologue: K' = -K, V' = -V fragments; zero accumulators; stage tile 0")
    o(f"s_mov_b32 s{S_M0}, m0")
    for ks in range(NKS):
        o(f"global_load_dwordx4 {KFR(ks)}, %[koffs], %[kbase] offset:{32 * ks}")
        o(f"global_load_dwordx4 {VFR(ks)}, %[voffs], %[vbase] offset:{32 * ks}")
    for i in range(32 * NDT):
        o(f"v_accvgpr_write_b32 a{i}, 0")
    o(f"s_mov_b64 s[{S_QP}:{S_QP + 1}], %[qbase]")
    o(f"s_mov_b64 s[{S_GP}:{S_GP + 1}], %[gbase]")
    o(f"s_mov_b64 s[{S_LP}:{S_LP + 1}], %[lbase]")
    o(f"s_mov_b64 s[{S_DP}:{S_DP + 1}], %[dbase]")
    o(f"s_sub_u32 s{S_SQ1}, %[sq], 1")
    o(f"s_lshl_b32 s{S_T0}, %[wv], 10")
    o(f"s_add_u32 s{S_T0}, s{S_T0}, %[lds]")                  # LDS address of this wave's first piece inside an image
    o(f"s_mov_b32 s{S_QQ0}, 0")
    o(f"s_mov_b32 s{S_STQ}, 0")
    for chn in stage_ops(0):
        st.extend(chn)
    st.extend(stat_stage(0))
    o("s_waitcnt vmcnt(0)")
    for r in list(range(32, 32 + 4 * NKS)) + list(range(64, 64 + 4 * NKS)):
        o(f"v_xor_b32_e32 {vr(r)}, 0x80008000, {vr(r)}")       # K' = -K, V' = -V (both bf16 halves of every register)
    o(f"s_mov_b32 s{S_CNT}, %[nqt]")
    # static priority for the second-dispatched half of the workgroup (waves 4-7): it is the arbitration loser of every SIMD pair otherwise
    # (MI355X_MICROARCH.md 'Two waves per SIMD' item 4; lab, same box: 1280-1285 -> 1302-1303 TFLOP/s; "lo" = waves 0-3 instead: 1268)
    prio = os.environ.get("DKV_PRIO", "hi")
    if prio in ("hi", "lo"):
        o("s_cmp_ge_u32 %[wv], 4")
        o(f"s_cbranch_scc{0 if prio == 'hi' else 1} .Ldkv_noprio_%=")
        o("s_setprio 1")
        o(".Ldkv_noprio_%=:")
    o("s_barrier")

    def block(slot: int, qb: int, nxt: tuple[int, int] | None, fill: list | None = None) -> None:
        """statistics + first two row fragments of this block were requested by the previous one; nxt = (slot, qb) of the following block or None"""
        pa = PHASE[0]
        pc = (pa + 2 * NKS) % 4                               # C's ring position; the next block's A starts 4 NDT fragments further
        pn = (pc + 4 * NDT) % 4
        tail_a = [tr_request(0, slot, qb, pc), tr_request(1, slot, qb, pc)]
        ag = a_groups(slot, qb, tail_a, pa)
        if nxt is not None:
            tail_c = [row_request(0, nxt[0], nxt[1], pn), row_request(1, nxt[0], nxt[1], pn)]
        else:
            tail_c = [[], []]
        cg = c_groups(slot, qb, tail_c, pc)
        PHASE[0] = pn
        b = b_ops()
        # A | 12 states for the MFMA results | B | C.  B cannot ride in this wave's own A or C gaps (it needs A's results, C needs its results): the partner
        # wave's MFMAs cover it.  Fillers (LDS-DMA of the next tile) ride in A's and C's gaps.
        segs = [(fill, 0, 2 * NKS - 1)] if fill else []
        st.extend(weave_budget(ag, segs, CAP) if segs else [x for g in ag for x in g])
        o("s_nop 11")
        st.extend(b)
        o("s_nop 1")
        for g in cg:
            st.extend(g)
        if nxt is not None:                                   # the next block's chains start from its statistics: after C's last MFMA has taken P / dS
            st.extend(stat_request(nxt[0], nxt[1]))

    def tile(slot: int, last: bool) -> None:
        """one 64-query tile in `slot`; unless last: stage the next tile into the other slot (clamped), then wait + barrier and continue with its block 0"""
        fill = None
        if not last:
            o(f"s_add_u32 s{S_QQ0}, s{S_QQ0}, 64")
            o(f"s_min_u32 s{S_STQ}, s{S_QQ0}, %[stmax]")
            st.extend(stat_stage(slot ^ 1))
            fill = stage_ops(slot ^ 1)
        block(slot, 0, (slot, 1), fill)
        if last:
            block(slot, 1, None)
        else:
            # block 1 must not request from the other slot before the barrier: its C tail carries nothing, the requests follow the barrier
            block(slot, 1, None)
            o("s_waitcnt vmcnt(0)")
            o("s_barrier")
            st.extend(stat_request(slot ^ 1, 0))
            st.extend(row_request(0, slot ^ 1, 0, PHASE[0]))
            st.extend(row_request(1, slot ^ 1, 0, PHASE[0]))

    st.comment("---- first block's statistics and fragments")
    st.extend(stat_request(0, 0))
    st.extend(row_request(0, 0, 0))
    st.extend(row_request(1, 0, 0))
    st.comment("---- main loop: two tiles per trip (slot 0, slot 1) while at least three tiles are left (so that both stage a real next tile or the clamp)")
    o(f"s_cmp_lt_u32 s{S_CNT}, 3")
    o("s_cbranch_scc1 .Ldkv_tail_%=")
    o(".Ldkv_loop_%=:")
    assert PHASE[0] == 0
    tile(0, False)
    tile(1, False)
    assert PHASE[0] == 0, "four blocks per trip must bring the fragment ring back to its entry position"
    o(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 2")
    o(f"s_cmp_ge_u32 s{S_CNT}, 3")
    o("s_cbranch_scc1 .Ldkv_loop_%=")
    o(".Ldkv_tail_%=:")
    st.comment("---- tail: one or two tiles left")
    o(f"s_cmp_eq_u32 s{S_CNT}, 1")
    o("s_cbranch_scc1 .Ldkv_one_%=")
    PHASE[0] = 0                                              # (both tails are entered at the loop's entry position)
    tile(0, False)
    tile(1, True)
    o("s_branch .Ldkv_done_%=")
    o(".Ldkv_one_%=:")
    PHASE[0] = 0
    tile(0, True)
    o(".Ldkv_done_%=:")
    st.comment("---- park dK^T * scale and dV^T as bf16 token rows (rope_bwd_store's image): dK at lds + wave * 16384, dV 8 KiB behind it")
    o("s_nop 15")
    o("s_barrier")
    o(f"s_lshl_b32 s{S_T1}, %[wv], 14")
    o(f"s_add_u32 s{S_T1}, s{S_T1}, %[lds]")
    # park lane base: l31 * 256 + ((l31 & 15) << 4) + 8 h   from the lane id
    o(f"v_mbcnt_lo_u32_b32 {vr(T[0])}, -1, 0")
    o(f"v_mbcnt_hi_u32_b32 {vr(T[0])}, -1, {vr(T[0])}")
    o(f"v_and_b32_e32 {vr(T[1])}, 31, {vr(T[0])}")
    o(f"v_lshrrev_b32_e32 {vr(T[0])}, 5, {vr(T[0])}")                      # h
    o(f"v_lshlrev_b32_e32 {vr(T[0])}, 3, {vr(T[0])}")                      # 8 h
    o(f"v_lshl_add_u32 {vr(T[0])}, {vr(T[1])}, 8, {vr(T[0])}")             # + l31 * 256
    o(f"v_and_b32_e32 {vr(T[1])}, 15, {vr(T[1])}")
    o(f"v_lshl_add_u32 {vr(T[0])}, {vr(T[1])}, 4, {vr(T[0])}")             # + (l31 & 15) << 4
    o(f"v_add_u32_e32 {vr(T[0])}, s{S_T1}, {vr(T[0])}")
    for which in range(2):
        for dt in range(NDT):
            for a in range(4):
                base = 16 * NDT * which + 16 * dt + 4 * a
                t = 96 + 4 * ((4 * dt + a) & 3)
                for bb in range(4):
                    o(f"v_accvgpr_read_b32 {vr(t + bb)}, a{base + bb}")
                if which == 0:
                    for bb in range(4):
                        o(f"v_mul_f32_e32 {vr(t + bb)}, %[scale], {vr(t + bb)}")
                o(f"v_cvt_pk_bf16_f32 {vr(t)}, {vr(t)}, {vr(t + 1)}")
                o(f"v_cvt_pk_bf16_f32 {vr(t + 1)}, {vr(t + 2)}, {vr(t + 3)}")
                ch = 4 * dt + a
                o(f"v_xor_b32_e32 {vr(T[1])}, {hex(ch << 4)}, {vr(T[0])}" if ch else f"v_mov_b32_e32 {vr(T[1])}, {vr(T[0])}")
                o(f"ds_write_b64 {vr(T[1])}, {vr(t, 2)} offset:{which * 8192}")
    o("s_waitcnt lgkmcnt(0)")
    o("s_setprio 0")
    o(f"s_mov_b32 m0, s{S_M0}")
    lines = resolve_lgkm(st.lines)
    bad = check_hazards(lines)
    if bad:
        raise SystemExit("hazard check failed:\n" + "\n".join(bad[:20]))
    return "\n".join('"' + ln.replace("\\", "\\\\") + '\\n"' for ln in lines) + "\n"


def main() -> None:
    name = "attn_dkv4_body.inc" if HD == 128 else f"attn_dkv4_hd{HD}_body.inc"
    out = os.environ.get("DKV_OUT") or os.path.join(os.path.dirname(__file__), "..", "..", "simpletuner_amd", "csrc", "gen", name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    body = build()
    with open(out, "w") as f:
        f.write("// GENERATED by tools/kgen/dkv.py — do not edit; regenerate with  python -m tools.kgen.dkv\n")
        f.write(body)
    if not os.environ.get("DKV_OUT") and HD == 128:
        regs = [f'"v{i}"' for i in range(10, 128)] + [f'"a{i}"' for i in range(128)] + [f'"s{i}"' for i in range(40, 76)]
        with open(os.path.join(os.path.dirname(out), "attn_dkv4_clobbers.inc"), "w") as f:
            f.write("// GENERATED by tools/kgen/dkv.py — the registers the dkv4 body owns\n")
            f.write(",\n".join(", ".join(regs[i:i + 16]) for i in range(0, len(regs), 16)) + "\n")
    print(f"wrote {os.path.normpath(out)}: {body.count(chr(10))} lines", file=sys.stderr)


if __name__ == "__main__":
    main()
