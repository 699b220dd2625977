"""Aspect buckets across replicas: `split_buckets_between_processes` against the reference's own method executed at fixture-generation
time (tools/gen_bucket_golden.py -> tests/golden/bucket_split_vectors.pt: 3 datasets x batch / ranks / grad-accum / repeats / padding / oversubscription
grid, including the configurations where the reference raises), and the token-balanced shared schedule on top of it."""
import os
import tempfile
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simpletuner_amd.training.bucket_split import TokenBalancedSchedule, shared_counts, split_buckets_between_processes

GOLD = Path(__file__).resolve().parent / "golden" / "bucket_split_vectors.pt"


def test_split_equals_the_reference_method_on_every_recorded_case():
    g = torch.load(GOLD, weights_only=False)
    n_ok = n_err = 0
    for case in g["cases"]:
        c, want = case["cfg"], case["res"]
        kw = dict(batch_size=c["batch_size"], num_processes=c["world"], rank=c["rank"], gradient_accumulation_steps=c["ga"], repeats=c["repeats"], seed=c["seed"],
                  backend_id=c["backend_id"], apply_padding=c["apply_padding"], allow_oversubscription=c["oversub"], user_set_repeats=c["user_repeats"])
        if want["ok"]:
            got = split_buckets_between_processes(g["datasets"][c["dataset"]], **kw)
            assert got == want["buckets"], c
            n_ok += 1
        else:
            with pytest.raises(ValueError, match="Dataset configuration will produce zero usable batches"):
                split_buckets_between_processes(g["datasets"][c["dataset"]], **kw)
            n_err += 1
    assert n_ok > 300 and n_err > 50          # the grid exercises both outcomes


def test_ranks_partition_each_bucket_without_overlap():
    buckets = {"1.0": [f"a{i}" for i in range(64)], "0.75": [f"b{i}" for i in range(40)]}
    parts = [split_buckets_between_processes(buckets, 2, 4, r, seed=5, backend_id="x") for r in range(4)]
    for b in buckets:
        seen = [s for p in parts for s in p[b]]
        assert len(seen) == len(set(seen)) == len(buckets[b]) and set(seen) == set(buckets[b])
        assert all(len(p[b]) == len(buckets[b]) // 4 for p in parts)


def _worker(rank, world, init_file, out_dir):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    buckets = {"128x128": [f"s{i}" for i in range(40)], "96x168": [f"p{i}" for i in range(26)], "168x96": [f"l{i}" for i in range(18)]}
    tokens = {"128x128": 4096, "96x168": 4032, "168x96": 4032}
    local = split_buckets_between_processes(buckets, 2, world, rank, seed=42, backend_id="ds")
    counts = shared_counts(local, 2)
    sched = TokenBalancedSchedule(local, 2, seed=42, epoch=1, counts=counts, tokens_of=tokens)
    steps = [(b, list(samples)) for b, samples in sched]
    torch.save({"order": [b for b, _ in steps], "samples": [s for _, s in steps], "tokens": sched.step_tokens(), "counts": counts}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_walk_one_token_balanced_schedule():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)
        r0, r1 = (torch.load(os.path.join(d, f"r{r}.pt")) for r in range(2))
    assert r0["order"] == r1["order"] and r0["tokens"] == r1["tokens"] and r0["counts"] == r1["counts"]      # same bucket, same token count, every step
    assert len(set(r0["order"])) == 3                                                                          # mixed buckets inside the epoch
    flat0 = [s for b in r0["samples"] for s in b]; flat1 = [s for b in r1["samples"] for s in b]
    assert not set(flat0) & set(flat1) and len(flat0) == len(set(flat0))                                       # disjoint samples per rank, none repeated
    assert all(len(b) == 2 for b in r0["samples"] + r1["samples"])
    # a different epoch reshuffles the order identically on both ranks (seeded by (seed, epoch) only)
    local = {"a": list(range(8)), "b": list(range(8))}
    o1 = TokenBalancedSchedule(local, 2, seed=1, epoch=1).order; o2 = TokenBalancedSchedule(local, 2, seed=1, epoch=2).order
    assert sorted(o1) == sorted(o2) and TokenBalancedSchedule(local, 2, seed=1, epoch=1).order == o1
    with pytest.raises(ValueError, match="schedule asks for"):
        TokenBalancedSchedule(local, 2, counts={"a": 5, "b": 1})
