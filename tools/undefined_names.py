#!/usr/bin/env python3
"""A linter-free check for names that are read but never bound (the image has no pyflakes): every name a scope treats as an implicit global must be bound at module
level or be a builtin.  Catches the code paths no test on a 1-GPU box executes (world > 1 branches of bench.py, error paths).

    python tools/undefined_names.py [files...]        # default: bench.py, __graft_entry__.py, simpletuner_amd/**/*.py, tools/*.py
"""
import builtins
import symtable
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def check(path: Path):
    src = path.read_text()
    top = symtable.symtable(src, str(path), "exec")
    module_names = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}
    star = "import *" in src
    bad = []

    def walk(tab):
        for s in tab.get_symbols():
            n = s.get_name()
            if s.is_referenced() and s.is_global() and not s.is_declared_global() and tab.get_type() != "module":
                if n not in module_names and not hasattr(builtins, n) and not star and n not in ("__file__", "__builtins__", "__name__"):
                    bad.append((tab.get_name(), tab.get_lineno(), n))
            if tab.get_type() == "module" and s.is_referenced() and not (s.is_assigned() or s.is_imported() or s.is_namespace()):
                if not hasattr(builtins, n) and n not in ("__file__", "__name__", "__doc__", "__path__") and not star:
                    bad.append(("<module>", 0, n))
        for ch in tab.get_children():
            walk(ch)

    walk(top)
    return bad


def main():
    files = [Path(a) for a in sys.argv[1:]] or ([ROOT / "bench.py", ROOT / "__graft_entry__.py"] + sorted((ROOT / "simpletuner_amd").rglob("*.py")) + sorted((ROOT / "tools").glob("*.py"))
                                                 + sorted((ROOT / "tools" / "kgen").glob("*.py")) + sorted((ROOT / "oracle").glob("*.py")))
    n_bad = 0
    for f in files:
        for scope, line, name in check(f):
            print(f"{f.relative_to(ROOT) if (f.is_absolute() and ROOT in f.parents) else f}:{line}: name {name!r} is read in {scope} but never bound")
            n_bad += 1
    print(f"{len(files)} files, {n_bad} unbound names")
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
