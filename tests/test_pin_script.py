"""tools/pin_against_gpytorch.py (the only path from GP parity 'partial' to 'green': it needs gpytorch, which neither
the build image nor the GPU box has) at least runs: --self-check puts the oracle in gpytorch's seat and must come out
'fresh' / 'jitter 1e-4' at round-off, through the same comparison code a real gpytorch run goes through."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tools", "pin_against_gpytorch.py")


def test_pin_script_self_check(tmp_path):
    out = tmp_path / "pin.json"
    r = subprocess.run([sys.executable, SCRIPT, "--self-check", "--kats", "m7_d6,m22_d32", "--json", str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(out.read_text())
    assert rep["u1"] == "fresh" and rep["u2"] == "1e-4" and rep["worst_var_rel"] < 1e-9
    assert "PINNED" in rep["verdict"]
    for row in rep["kats"].values():
        assert row["f64"]["var_vs_jitter1e-3"] > 1e-5  # the jitter alternative is distinguishable on these vectors
        assert row["f32"]["var_vs_fresh"] < 1e-2      # the reference's float32 / float64 split stays near


def test_pin_script_imports_gpytorch_only_never_the_reference():
    src = open(SCRIPT).read()
    assert "/root/reference" not in src and "import gen_ps" not in src and "gaussian_process_utils import" not in src
    r = subprocess.run([sys.executable, SCRIPT, "--kats", "m7_d6"], capture_output=True, text=True, timeout=120)
    try:
        import gpytorch  # noqa: F401
    except ImportError:
        assert r.returncode == 2 and "gpytorch is not installed" in r.stdout
