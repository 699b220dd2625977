"""Activation-checkpoint segment plans: FluxTransformer2DModel._checkpoint_segments must cut the block stacks exactly where the
reference's `checkpoint_sequential_state` (helpers/training/gradient_checkpointing_interval.py:69-120, executed here from where it lies — a pure
python module) opens / closes its checkpoint calls, for the published-table modes `layer`, `interval2`, `seg2-stride4` and a few others."""
import importlib.util
from pathlib import Path

import pytest

REF = Path("/root/reference/simpletuner/helpers/training/gradient_checkpointing_interval.py")


def _ref_plan(gci, n, interval, stride):
    log = []

    def run_block(i, blk, *st):
        log.append(("blk", i))
        return st

    def ck(fn, *st, **kw):
        log.append(("begin",))
        r = fn(*st)
        log.append(("end",))
        return r

    gci.checkpoint_sequential_state(list(range(n)), interval, (0,), run_block, ck, {}, stride)
    segs, cur = [], None
    for e in log:
        if e[0] == "begin":
            cur = []
        elif e[0] == "end":
            segs.append((cur[0], len(cur), True)); cur = None
        elif cur is not None:
            cur.append(e[1])
        else:
            segs.append((e[1], 1, False))
    return segs


def _model():
    from simpletuner_amd.flux.transformer import FluxTransformer2DModel
    return FluxTransformer2DModel.__new__(FluxTransformer2DModel)


@pytest.mark.skipif(not REF.exists(), reason="the reference tree is only present in the build container")
def test_segment_plans_equal_the_reference_helper():
    spec = importlib.util.spec_from_file_location("ref_gci", str(REF))
    gci = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gci)
    m = _model()
    m.gradient_checkpointing = True
    for n in (19, 38, 5, 1):
        for interval, stride in ((2, None), (2, 4), (3, 3), (4, 6), (2, 2)):
            m.gradient_checkpointing_interval, m.gradient_checkpointing_segment_stride = interval, stride
            assert m._checkpoint_segments(n) == _ref_plan(gci, n, interval, stride), (n, interval, stride)
        # non-segmented: should_checkpoint_block(interval None) == every block its own checkpoint
        m.gradient_checkpointing_interval, m.gradient_checkpointing_segment_stride = None, None
        assert m._checkpoint_segments(n) == [(i, 1, gci.should_checkpoint_block(i, True, None)) for i in range(n)]


def test_modes_off_layer_and_bad_stride():
    m = _model()
    m.gradient_checkpointing, m.gradient_checkpointing_interval, m.gradient_checkpointing_segment_stride = False, 2, 4
    assert m._checkpoint_segments(4) == [(0, 1, False), (1, 1, False), (2, 1, False), (3, 1, False)]
    m.gradient_checkpointing = True
    assert m._checkpoint_segments(6) == [(0, 2, True), (2, 1, False), (3, 1, False), (4, 2, True)]       # seg2-stride4
    m.gradient_checkpointing_segment_stride = 1
    with pytest.raises(ValueError, match="segment_stride must be at least"):
        m._checkpoint_segments(6)
    m.gradient_checkpointing_backend = "torch"
    with pytest.raises(NotImplementedError):
        m.set_gradient_checkpointing_backend("unsloth")
