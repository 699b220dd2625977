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
