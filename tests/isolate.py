"""Run one test body in its own interpreter (tests/test_zz_dit_gpu.py): a GPU memory fault or a hang inside a kernel that has
never run before then costs that one test, not the pytest process that holds everyone else's results."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_isolated(module, func, args=(), timeout=600):
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import importlib; "
            f"m = importlib.import_module({module!r}); getattr(m, {func!r})(*{tuple(args)!r}); print('ISOLATED-OK')")
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    except subprocess.TimeoutExpired as e:
        raise AssertionError(f"{func}{tuple(args)} timed out after {timeout} s: {(e.stderr or '')[-2000:]}")
    assert r.returncode == 0 and "ISOLATED-OK" in r.stdout, \
        f"{func}{tuple(args)} exit {r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[-6000:]}"
