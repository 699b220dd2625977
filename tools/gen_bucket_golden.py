#!/usr/bin/env python3
"""tests/golden/bucket_split_vectors.pt: outputs of the reference's OWN `MetadataBackend.split_buckets_between_processes`
(/root/reference/simpletuner/helpers/metadata/backends/base.py:741-937), lifted by AST at generation time and executed on a stand-in `self` with the
module globals it needs stubbed (logger, StateTracker, the context-parallel helper, the dataset-type enum).  Nothing of the reference is copied into
the repo; only the resulting bucket -> sample-list maps are committed.

    python tools/gen_bucket_golden.py
"""
import ast
import itertools
import logging
import os
import random
from math import ceil
from pathlib import Path
from types import SimpleNamespace

import torch

REF = Path("/root/reference/simpletuner/helpers/metadata/backends/base.py")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "bucket_split_vectors.pt"


def lift():
    tree = ast.parse(REF.read_text())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and any(isinstance(m, ast.FunctionDef) and m.name == "split_buckets_between_processes" for m in n.body))
    fn = next(m for m in cls.body if isinstance(m, ast.FunctionDef) and m.name == "split_buckets_between_processes")
    fn.decorator_list = []
    mod = ast.Module(body=[fn], type_ignores=[])
    ast.fix_missing_locations(mod)
    return mod


def run_case(code, buckets, batch_size, world, rank, ga, repeats, seed, backend_id, apply_padding, oversub, user_repeats):
    args = SimpleNamespace(seed=seed, allow_dataset_oversubscription=oversub)

    class StateTracker:
        @staticmethod
        def get_data_backend_config(_id):
            return {"repeats": repeats if user_repeats else 0}

        @staticmethod
        def get_args():
            return args

    class DatasetType:
        IMAGE, EVAL = "image", "eval"

    ns = {"logger": logging.getLogger("ref"), "StateTracker": StateTracker, "DatasetType": DatasetType, "ensure_dataset_type": lambda v: DatasetType.IMAGE,
          "get_cp_aware_dp_info": lambda acc: (world, rank, 1), "should_log": lambda: False, "ceil": ceil, "os": os, "random": random,
          "broadcast_object_from_main": lambda v: v}
    exec(compile(code, str(REF), "exec"), ns)
    me = SimpleNamespace(aspect_ratio_bucket_indices={k: list(v) for k, v in buckets.items()}, accelerator=SimpleNamespace(num_processes=world, is_main_process=rank == 0),
                         id=backend_id, batch_size=batch_size, repeats=repeats, bucket_report=None, dataset_type="image", read_only=False)
    try:
        ns["split_buckets_between_processes"](me, gradient_accumulation_steps=ga, apply_padding=apply_padding)
        return {"ok": True, "buckets": me.aspect_ratio_bucket_indices}
    except ValueError as e:
        return {"ok": False, "error": str(e).splitlines()[0]}


def main():
    code = lift()
    rnd = random.Random(7)
    datasets = {
        "even": {"1.0": [f"/d/sq_{i:03d}.png" for i in range(64)], "0.75": [f"/d/p_{i:03d}.png" for i in range(48)], "1.33": [f"/d/l_{i:03d}.png" for i in range(32)]},
        "ragged": {"1.0": [f"/d/a{i}.jpg" for i in rnd.sample(range(1000), 37)], "0.56": [f"/d/b{i}.jpg" for i in rnd.sample(range(1000), 19)], "1.78": [f"/d/c{i}.jpg" for i in range(9)],
                   "empty": []},
        "tiny": {"1.0": ["/d/x1.png", "/d/x2.png", "/d/x3.png"], "2.0": [f"/d/y{i}.png" for i in range(11)]},
    }
    cases = []
    for name, buckets in datasets.items():
        for batch, world, ga, repeats, pad, oversub in itertools.product((1, 2, 4), (1, 2, 4, 8), (1, 2), (0, 3), (False, True), (False, True)):
            if name == "even" and (oversub or repeats):
                continue
            for rank in sorted({0, world - 1, world // 2}):
                cfg = dict(dataset=name, batch_size=batch, world=world, rank=rank, ga=ga, repeats=repeats, seed=42, backend_id="ds-" + name, apply_padding=pad,
                           oversub=oversub, user_repeats=repeats > 0)
                res = run_case(code, buckets, batch, world, rank, ga, repeats, 42, "ds-" + name, pad, oversub, repeats > 0)
                cases.append({"cfg": cfg, "res": res})
    OUT.parent.mkdir(parents=True, exist_ok=True)
    torch.save({"datasets": datasets, "cases": cases}, OUT)
    ok = sum(c["res"]["ok"] for c in cases)
    print(f"wrote {OUT}: {len(cases)} cases ({ok} splits, {len(cases) - ok} reference ValueErrors)")


if __name__ == "__main__":
    main()
