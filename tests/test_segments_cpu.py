"""Host logic of the segmented-row operands (st355_gemm_args.seg_rows): how a [B, rows, C] strided view of a joint [B, S, C] buffer turns into
(ld, seg_rows, seg_stride) and how the Flux engine decides between ONE segmented problem per stream and one problem per sample.
No device work: `_seg` only reads shapes / strides, `_problems` only slices views."""
from types import SimpleNamespace

import pytest
import torch

from simpletuner_amd import ops
from simpletuner_amd.flux.transformer import FluxTransformer2DModel as M
from simpletuner_amd.lib import St355Error


def test_seg_descriptor_of_views():
    B, S, C, lo, rows = 3, 768, 64, 256, 512
    joint = torch.zeros(B * S, C)
    v = joint.view(B, S, C)[:, lo:lo + rows]
    assert ops._seg(v, "a") == (B * rows, C, C, rows, S)                     # rows_total, cols, ld, seg_rows, stride between segments (rows)
    assert ops._seg(joint, "a") == (B * S, C, C, 0, 0)                       # plain 2-D operand: no segments
    cols = joint.view(B, S, C)[:, lo:lo + rows, 16:48]                        # a column block of the same rows: ld stays the buffer's row stride
    assert ops._seg(cols, "a") == (B * rows, 32, C, rows, S)
    compact = torch.zeros(B, rows, C)
    assert ops._seg(compact, "a") == (B * rows, C, C, rows, rows)             # compact 3-D: stride == seg_rows
    with pytest.raises(St355Error):
        ops._seg(joint.view(B, S, C)[:, :, ::2], "a")                         # inner stride 2
    with pytest.raises(St355Error):
        ops._seg(torch.zeros(2, 3, 5, 7), "a")
    assert ops._seg_join(0, 512, "t") == 512 and ops._seg_join(512, 0, "t") == 512 and ops._seg_join(512, 512, "t") == 512
    with pytest.raises(St355Error):
        ops._seg_join(512, 256, "t")


def _env(B, S, Si, St):
    return SimpleNamespace(B=B, S=S, Si=Si, St=St)


def test_rows_of_and_problem_expansion():
    B, Si, St, C = 4, 512, 256, 8
    S = Si + St
    env = _env(B, S, Si, St)
    joint = torch.arange(B * S * C, dtype=torch.float32).view(B * S, C)
    img = M._rows_of(joint, St, Si, env)
    assert img.shape == (B, Si, C) and img.data_ptr() == joint[St].data_ptr() and img.stride() == (S * C, C, 1)
    assert torch.equal(img[2], joint[2 * S + St:3 * S])
    compact = torch.zeros(B * Si, C)
    gate = torch.zeros(B, C)
    # tile-aligned rows: ONE segmented problem, operands untouched
    pr = dict(a=compact, w=torch.zeros(C, C), out=img, aux_in=compact, gate=gate, rows_per_batch=Si, epilogue=2)
    assert M._problems(env, Si, pr) == [pr]
    # rows not a multiple of the 256-row tile: one problem per sample, views / slices per sample, gate row per sample
    env2 = _env(B, 300 + 40, 300, 40)
    joint2 = torch.zeros(B * env2.S, C)
    out2 = M._rows_of(joint2, 40, 300, env2)
    c2 = torch.zeros(B * 300, C)
    ps = M._problems(env2, 300, dict(a=c2, w=torch.zeros(C, C), out=out2, gate=gate, rows_per_batch=300))
    assert len(ps) == B
    for b, q in enumerate(ps):
        assert q["a"].shape == (300, C) and q["a"].data_ptr() == c2[b * 300].data_ptr()
        assert q["out"].shape == (300, C) and q["out"].data_ptr() == joint2[b * env2.S + 40].data_ptr()
        assert q["gate"].shape == (1, C) and q["gate"].data_ptr() == gate[b].data_ptr() and q["rows_per_batch"] == 300
    # batch 1: 3-D views collapse to plain 2-D operands
    env1 = _env(1, S, Si, St)
    j1 = torch.zeros(S, C)
    (p1,) = M._problems(env1, Si, dict(a=torch.zeros(Si, C), w=torch.zeros(C, C), out=M._rows_of(j1, St, Si, env1)))
    assert p1["out"].dim() == 2 and p1["out"].data_ptr() == j1[St].data_ptr()
    # operands of the rank-space gradient kernels
    assert M._compact(img, env, Si) is img                                     # aligned: walked in place
    assert M._compact(out2, env2, 300).shape == (B * 300, C)                    # unaligned: compact copy
    assert M._compact(M._rows_of(j1, St, Si, env1), env1, Si).dim() == 2


def test_sd3_stream_problems_policy():
    """SD3's joint blocks: aligned image rows -> one segmented problem; the short unaligned text rows -> ONE compact problem with gather copies of the
    joint-buffer inputs and a scatter-back closure for a joint-buffer output; big unaligned blocks (odd aspect buckets) -> one problem per sample."""
    from simpletuner_amd.sd3.transformer import _rows3, _stream_problems

    B, Si, St, C = 3, 512, 154, 8
    S = Si + St
    joint = torch.arange(B * S * C, dtype=torch.float32).view(B * S, C)
    w = torch.zeros(C, C)
    # aligned: untouched, nothing to do afterwards
    after = []
    pr = dict(a=torch.zeros(B * Si, C), w=w, out=_rows3(joint, 0, Si, B, S))
    assert _stream_problems(B, S, Si, pr, after) == [pr] and after == []
    # text rows: compact gather of the joint input, temporary + scatter for the joint output
    out_joint = torch.zeros(B * S, C)
    a_view = _rows3(joint, Si, St, B, S)
    (q,) = _stream_problems(B, S, St, dict(a=a_view, w=w, out=_rows3(out_joint, Si, St, B, S), gate=torch.zeros(B, C), rows_per_batch=St), after)
    assert q["a"].shape == (B * St, C) and torch.equal(q["a"].view(B, St, C), a_view) and q["a"].data_ptr() != joint.data_ptr()
    assert q["out"].shape == (B * St, C) and len(after) == 1 and q["gate"].shape == (B, C)
    q["out"].copy_(torch.arange(B * St * C, dtype=torch.float32).view(B * St, C) + 1000.0)      # "the GEMM wrote its result"
    after[0]()
    assert torch.equal(out_joint.view(B, S, C)[:, Si:], q["out"].view(B, St, C)) and out_joint.view(B, S, C)[:, :Si].abs().max() == 0
    # compact operands of an unaligned stream need no copies at all
    after2 = []
    (q2,) = _stream_problems(B, S, St, dict(a=torch.zeros(B * St, C), w=w, out=torch.zeros(B * St, C)), after2)
    assert after2 == [] and q2["a"].dim() == 2
    # a big unaligned block stays per sample (no after-closures, views per sample)
    Sb = 1100 + St
    jb = torch.zeros(B * Sb, C)
    after3 = []
    ps = _stream_problems(B, Sb, 1100, dict(a=torch.zeros(B * 1100, C), w=w, out=_rows3(jb, 0, 1100, B, Sb)), after3)
    assert len(ps) == B and after3 == [] and ps[1]["out"].data_ptr() == jb[Sb].data_ptr()
    # batch 1: plain 2-D slices
    (p1,) = _stream_problems(1, S, St, dict(a=_rows3(joint[:S], Si, St, 1, S), w=w), [])
    assert p1["a"].dim() == 2 and p1["a"].data_ptr() == joint[Si].data_ptr()
