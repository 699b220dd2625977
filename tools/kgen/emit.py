"""kgen.emit — tiny helpers for generating hand-scheduled gfx950 instruction streams (the bodies of the one-wave-per-SIMD attention kernels).

The generated text is the body of ONE `asm volatile` statement inside a HIP kernel (simpletuner_amd/csrc/gen/*.inc): the statement owns literal register
ranges (listed as clobbers by the kernel), everything else reaches it as named operands.  hipcc neither schedules nor pads what is inside such a
statement (cdna_hip_programming.md §5.7), so the generator is responsible for
  * every s_waitcnt (LDS reads: lgkmcnt; LDS-DMA: vmcnt),
  * the gfx950 wait-state rules between dependent instructions (measured from hipcc's own output, /tmp probes of round 4):
        MFMA result -> any reader / overwriter other than the next MFMA accumulating into it ........ >= 12 states
        VALU (or v_accvgpr_write) result -> MFMA A / B / C operand ..................................... >= 2 states
        transcendental (v_exp_f32) result -> its consumer ................................................ >= 1 state
        s_mov m0 -> LDS-DMA .............................................................................. >= 1 state
    The streams below keep these by construction (distance in instructions); `Stream.check()` re-verifies them on the final text.
"""
from __future__ import annotations

import re


def vr(lo: int, n: int = 1) -> str:
    return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"


def ar(lo: int, n: int = 1) -> str:
    return f"a{lo}" if n == 1 else f"a[{lo}:{lo + n - 1}]"


class Stream:
    """an ordered list of instruction lines; `fill()` weaves filler instructions into the gaps after MFMAs"""

    def __init__(self) -> None:
        self.lines: list[str] = []

    def op(self, text: str) -> None:
        self.lines.append(text)

    def extend(self, other: "Stream | list[str]") -> None:
        self.lines.extend(other.lines if isinstance(other, Stream) else other)

    def comment(self, text: str) -> None:
        self.lines.append(f"; {text}")

    def text(self) -> str:
        return "\n".join(self.lines) + "\n"


def weave(mfmas: list[list[str]], fillers: list[str], per_gap: list[int] | None = None) -> list[str]:
    """mfmas: list of groups, each group = lines that must stay together and end with one MFMA (leading waits / reads allowed);
    fillers are distributed over the gaps AFTER each group (per_gap[i] fillers after group i; default: spread evenly, remainder in the earliest gaps)."""
    n = len(mfmas)
    if per_gap is None:
        base, rem = divmod(len(fillers), n)
        per_gap = [base + (1 if i < rem else 0) for i in range(n)]
    assert sum(per_gap) == len(fillers), (sum(per_gap), len(fillers))
    out: list[str] = []
    k = 0
    for g, cnt in zip(mfmas, per_gap):
        out.extend(g)
        out.extend(fillers[k:k + cnt])
        k += cnt
    return out


_REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs_of(tok: str) -> set[tuple[str, int]]:
    s: set[tuple[str, int]] = set()
    for m in _REG.finditer(tok):
        if m.group(1):
            for i in range(int(m.group(2)), int(m.group(3)) + 1):
                s.add((m.group(1), i))
        else:
            s.add((m.group(4), int(m.group(5))))
    return s


def check_hazards(lines: list[str]) -> list[str]:
    """Re-derive the wait-state rules on straight-line text (labels / branches reset the window conservatively: the generator keeps hazards inside
    basic blocks anyway).  Returns a list of violations (empty = clean).  Instructions count one state each; `s_nop N` counts N + 1."""
    problems: list[str] = []
    hist: list[tuple[str, set, set, int]] = []   # (kind, writes, reads, states_since)
    def states(line: str) -> int:
        m = re.match(r"s_nop\s+(\d+)", line)
        return int(m.group(1)) + 1 if m else 1
    for ln, raw in enumerate(lines):
        line = raw.split(";")[0].strip()
        if not line or line.endswith(":"):
            continue
        parts = line.split(None, 1)
        opc = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if opc.startswith("v_mfma"):
            kind, writes = "mfma", regs_of(ops[0])
            reads_ab = regs_of(ops[1]) | regs_of(ops[2])
            reads_c = regs_of(ops[3]) if len(ops) > 3 else set()
            reads = reads_ab | reads_c
        elif opc.startswith("v_") or opc.startswith("ds_") or opc.startswith("global_") or opc.startswith("buffer_"):
            kind = "trans" if opc.startswith("v_exp") or opc.startswith("v_log") or opc.startswith("v_rcp") else "valu" if opc.startswith("v_") else "mem"
            if opc.startswith("ds_write") or opc.startswith("global_store") or (opc.startswith("global_load_lds")):
                writes, reads = set(), set().union(*[regs_of(o) for o in ops]) if ops else set()
            else:
                writes = regs_of(ops[0]) if ops else set()
                reads = set().union(*[regs_of(o) for o in ops[1:]]) if len(ops) > 1 else set()
            reads_ab, reads_c = set(), set()
        else:
            kind, writes, reads, reads_ab, reads_c = "other", set(), set(), set(), set()
        # look back
        dist = 0
        for pk, pw, pr, st in reversed(hist):
            if dist > 16:
                break
            need = 0
            if pk == "mfma":
                if kind == "mfma":
                    if pw & reads_ab: need = 12
                    elif pw & reads_c and not (pw == reads_c and pw == writes): need = 12     # partial / shifted overlap
                    elif pw & writes and not (pw == writes): need = 12
                else:
                    if pw & (reads | writes): need = 12
            elif pk in ("valu", "trans") and kind == "mfma" and pw & reads:
                need = 2
            elif pk in ("valu", "trans") and opc.startswith("v_permlane") and pw & (reads | writes):
                need = 2
            elif pk == "trans" and kind in ("valu", "trans") and pw & reads:
                need = 1
            if need and dist < need:
                problems.append(f"line {ln}: '{line}' needs {need} states after a {pk} writing {sorted(pw & (reads | writes))[:2]} (has {dist})")
            dist += st
        hist.append((kind, writes, reads, states(line)))
        if len(hist) > 40:
            hist.pop(0)
    return problems


def resolve_lgkm(lines: list[str], loop_label: str | None = None, loop_branch: str | None = None) -> list[str]:
    """Counted LDS waits.  A line ending in `;@ld:<tag>` is an LDS operation (lgkmcnt += 1, returns in order); a line `@wait:<tag>` becomes
    `s_waitcnt lgkmcnt(N)` with N = the number of LDS operations issued after the LAST outstanding one carrying <tag> (nothing if none is outstanding).
    ds_write / untagged ds_read lines also count.  An explicit `s_waitcnt lgkmcnt(0)` clears the queue.  The walk is linear; for the one loop of a body
    the queue at the back-branch must equal the queue at the loop label (checked), so the counts hold on every path."""
    out: list[str] = []
    q: list[str] = []
    at_label: list[str] | None = None
    for raw in lines:
        line = raw.strip()
        if line.startswith("@wait:"):
            tag = line[6:]
            idx = max((i for i, t in enumerate(q) if t == tag), default=-1)
            if idx >= 0:
                n = len(q) - 1 - idx
                assert n <= 15, (tag, n)
                out.append(f"s_waitcnt lgkmcnt({n})")
                q = q[idx + 1:]
            continue
        if loop_label and line == loop_label:
            at_label = list(q)
        if loop_branch and line == loop_branch:
            assert at_label is not None and q == at_label, f"LDS queue at the back-branch {q} != at the loop label {at_label}"
        if "lgkmcnt(0)" in line:
            q = []
        elif ";@ld:" in line:
            q.append(line.split(";@ld:")[1].strip())
            raw = raw.split(";@ld:")[0].rstrip()
        elif line.startswith("ds_"):
            q.append("?")
        out.append(raw)
    return out


def weave_budget(groups: list[list[str]], segments: list[tuple], cap: float = 5, cost=None) -> list[str]:
    """Budgeted weaving for a one-wave-per-SIMD stream: every MFMA gap may carry at most `cap` issues besides the MFMA (MI355X_MICROARCH: 'one wave per
    SIMD: single-issue instructions hidden per MFMA gap <= 5'; measured here: the gaps that also held the next group's wait + address + two reads ran 8-9
    issues and cost ~30 cycles each).  groups[k] = head lines + one MFMA; gap k lies between MFMA k and MFMA k+1 and already holds head(k+1).
    segments: (fillers, first gap, last gap[, "chain"]), consumed in order inside their window, earliest gap first; a filler is a line or a list of
    lines that must stay together.  A "chain" segment starts no earlier than the gap where the previous chain segment ended (program order between
    dependent segments is kept).  Raises if a segment does not fit."""
    n = len(groups)
    # scalar-unit instructions (waits, nops, SALU, branches) are issued by a different port than VALU / LDS / VMEM: counted as half an issue; labels are free
    cost = cost or (lambda line: 0 if line.startswith(";") or line.endswith(":") else 0.5 if line.startswith(("s_", "@wait")) else 1)
    used = [0.0] * n
    for k in range(n - 1):
        used[k] = sum(cost(x) for x in groups[k + 1][:-1])
    placed: list[list[str]] = [[] for _ in range(n)]
    chain_at = 0
    for seg in segments:
        fill, g0, g1 = seg[0], seg[1], seg[2]
        chained = len(seg) > 3 and seg[3] == "chain"
        k = max(g0, chain_at) if chained else g0
        for item in fill:
            lines = [item] if isinstance(item, str) else list(item)
            c = sum(cost(x) for x in lines)
            while k <= g1 and used[k] + c > cap and not (used[k] == 0 and c > cap):
                k += 1
            if k > g1:
                raise ValueError(f"segment of {len(fill)} fillers does not fit into gaps {g0}..{g1} (cap {cap}); used = {used[g0:g1 + 1]}")
            placed[k].extend(lines)
            used[k] += c
        if chained:
            chain_at = k
    out: list[str] = []
    for k in range(n):
        out.extend(groups[k])
        out.extend(placed[k])
    return out
