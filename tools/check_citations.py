#!/usr/bin/env python3
"""Sweep every `file.py:line[-line]` citation in DESIGN.md / INTEGRATION.md / include/st355.h and the docstrings under simpletuner_amd/ and oracle/:
the cited reference file must exist under /root/reference and the line range must lie inside it.  Build-container tool (reads /root/reference).

    python tools/check_citations.py        # exit status 1 when a citation does not resolve
"""
import glob
import os
import re
import sys

REF, ROOT = "/root/reference", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASES = ["", "simpletuner", "simpletuner/helpers", "simpletuner/helpers/models", "simpletuner/helpers/training", "simpletuner/helpers/data_backend",
         "simpletuner/helpers/caching", "simpletuner/helpers/training/optimizers", "simpletuner/helpers/training/quantisation"]
PAT = re.compile(r"([A-Za-z_][\w/\.]*\.(?:py|md)):(\d+)(?:-(\d+))?")


def main() -> int:
    if not os.path.isdir(REF):
        sys.exit("reference tree not present (this tool only runs in the build container)")
    files = ["DESIGN.md", "INTEGRATION.md", "include/st355.h"] + [os.path.relpath(p, ROOT) for d in ("simpletuner_amd", "oracle")
                                                                   for p in glob.glob(os.path.join(ROOT, d, "**", "*.py"), recursive=True)]
    lens, bad, total = {}, [], 0
    for f in files:
        for m in PAT.finditer(open(os.path.join(ROOT, f), errors="ignore").read()):
            path, a, b = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
            if os.path.exists(os.path.join(ROOT, path)) and not os.path.exists(os.path.join(REF, path)):
                continue                                             # a citation of this repo's own file
            total += 1
            hit = [c for c in (os.path.join(REF, base, path) for base in BASES) if os.path.isfile(c)] or glob.glob(f"{REF}/**/{path}", recursive=True)[:1]
            if not hit:
                bad.append((f, m.group(0), "no such reference file"))
                continue
            if hit[0] not in lens:
                lens[hit[0]] = sum(1 for _ in open(hit[0], errors="ignore"))
            if max(a, b) > lens[hit[0]]:
                bad.append((f, m.group(0), f"file has {lens[hit[0]]} lines"))
    print(f"{total} citations checked, {len(bad)} problems")
    for row in bad:
        print("  ", *row)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
