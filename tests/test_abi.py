"""CPU: the C-ABI library builds/loads and exports every symbol include/fdmi.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "fdmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fdmi_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from flash_diffusion_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    l = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(l, n), f"libfdmi.so does not export {n}"
    l.fdmi_version.restype = ctypes.c_int
    assert l.fdmi_version() >= 1


def test_ctypes_binding_covers_header():
    from flash_diffusion_amd import _lib
    import flash_diffusion_amd.unet  # noqa: F401  (registers the plan-API signatures)
    bound = set(_lib.declared_symbols())
    for n in _declared():
        assert n in bound, f"{n} declared in fdmi.h but not bound in _lib.py"


def test_missing_library_fails_loudly(monkeypatch):
    import pytest
    from flash_diffusion_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libfdmi.so")
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.lib()


def test_allreduce_entry_points_fail_cleanly_without_a_communicator():
    """fdmi_allreduce* (RCCL bound at run time): before fdmi_allreduce_init the collective returns an error code with a
    message, the library has no load-time dependency on librccl, and world size reads 0"""
    import subprocess
    from flash_diffusion_amd import _lib
    L = _lib.lib()
    assert L.fdmi_allreduce_world() == 0
    assert L.fdmi_allreduce(None, 0, 0, None) != 0 and b"fdmi_allreduce_init" in L.fdmi_last_error()
    assert L.fdmi_allreduce_init(3, 2, None) != 0 and b"bad rank" in L.fdmi_last_error()
    assert L.fdmi_allreduce_destroy() == 0
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in needed.lower()
