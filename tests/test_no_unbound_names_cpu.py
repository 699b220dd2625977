"""Names that are read but never bound, in every product / bench / tool source (the image has no linter): tools/undefined_names.py walks the symbol tables.  The
code paths a 1-GPU box never executes — the world > 1 branches of bench.py above all — are the reason: round 4 found `comm_rep` computed in one function and read
in another this way."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_no_unbound_names_in_product_bench_and_tools():
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "undefined_names.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
