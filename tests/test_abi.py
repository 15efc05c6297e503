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


def _plan(M, N, K, splitk=1, **kw):
    import ctypes as C
    from flash_diffusion_amd import _lib
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.lda, d.ldw, d.splitk, d.use_glds, d.alpha = M, N, K, K, K, splitk, 1, 1.0
    for k, v in kw.items():
        setattr(d, k, v)
    out = [C.c_int32() for _ in range(4)]
    assert _lib.lib().fdmi_gemm_plan(C.byref(d), *[C.byref(x) for x in out]) == 0
    return tuple(x.value for x in out)     # (kernel, BM, BN, splitk)


def test_gemm_planner_choices_for_the_benchmark_shapes():
    """host-only planner query (fdmi_gemm_plan): the 256x320 LDS-DMA kernel takes the C2 linears whose N divides by 320, its
    256x192 variant every large DiT / MMDiT linear (N = 1152 / 1536 / 4608 / 6144 divide by 192, not by 320), the small
    tiles the per-sample-vector GEMMs; LoRA weight gradients (long contraction, tiny output) are split 16 ways"""
    assert _plan(65536, 320, 320)[:3] == (2, 256, 320) and _plan(65536, 2560, 320)[:3] == (2, 256, 320)   # SD1.5 level 0
    assert _plan(16384, 640, 640)[0] == 1 and _plan(4096, 1280, 1280)[0] == 1
    # round 6: the wide 256 x 384 tile where it quantises at least as well (its K loop is 9 - 16 % faster: profiles/r6_kloop_dit_widths.txt);
    # N = 1152 at 32 768 rows is 384 tiles = 1.5 rounds of 256 CUs against 3 full rounds of 256 x 192 tiles: the narrow tile stays
    from flash_diffusion_amd import _lib
    wide = {(32768, 4608, 1152), (32768, 3456, 1152), (16384, 1536, 1536), (16384, 4608, 1536), (16384, 6144, 1536), (16384, 1536, 6144)}
    narrow = {(32768, 1152, 1152), (32768, 1152, 4608)}
    for shape in sorted(wide | narrow):
        assert _plan(*shape)[:3] == (2, 256, 384 if shape in wide else 192), shape
    assert _plan(32768, 4608, 1152, act=5)[:3] == (2, 256, 384) and _plan(32768, 4608, 1152, act=1)[:3] == (2, 256, 192)   # tanh-GELU yes, SiLU no
    _lib.lib().fdmi_tune_set(51, 1)                               # A/B switch: the planner without the wide tile
    try:
        for shape in sorted(wide | narrow):
            assert _plan(*shape)[:3] == (2, 256, 192), shape
    finally:
        _lib.lib().fdmi_tune_set(51, 0)
    assert _plan(8, 6912, 1152)[0] == 0 and _plan(32768, 32, 1152)[0] == 0
    k, bm, bn, sk = _plan(1152, 64, 32768, splitk=0, accum_atomic=1, out_f32=1)
    assert sk == 16 and k == 0


def test_planner_offers_the_256x192_tile_for_the_transformer_widths():
    from flash_diffusion_amd import _lib
    L = _lib.lib()
    assert _plan(32768, 1152, 1152)[:3] == (2, 256, 192) and _plan(16384, 6144, 1536)[:3] == (2, 256, 384)
    assert _plan(65536, 320, 320)[:3] == (2, 256, 320)            # widths that 320 divides keep the larger tile
    L.fdmi_tune_set(12, 1)                                        # A/B switch: the planner without the tile
    try:
        assert _plan(32768, 1152, 1152)[:3] == (1, 256, 128)
    finally:
        L.fdmi_tune_set(12, 0)


def test_ctypes_argument_types_match_the_header_prototypes():
    """every prototype of include/fdmi.h, parsed here, against the argtypes / restype _lib.py binds: a float bound as a double or
    an int64 as an int32 is silent at call time and garbage in the kernel (capi.hip includes the same header under extern "C",
    so a definition that disagrees with it does not compile)"""
    import ctypes as C
    from flash_diffusion_amd import _lib
    import flash_diffusion_amd.unet  # noqa: F401
    src = open(os.path.join(ROOT, "include", "fdmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = re.findall(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(fdmi_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src)
    scalars = {"int": "i32", "int32_t": "i32", "int64_t": "i64", "float": "f32", "double": "f64", "void": "void"}

    def from_c(t):
        t = re.sub(r"\bconst\b", "", t).strip()
        if "*" in t:
            return "char*" if t.replace("*", "").strip() == "char" else "ptr"
        return scalars[t]

    def from_ctypes(ct):
        if ct is None:
            return "void"
        if ct is C.c_char_p:
            return "char*"
        if ct is C.c_void_p or (hasattr(ct, "_type_") and not isinstance(ct._type_, str)):
            return "ptr"
        return {C.c_int32: "i32", C.c_int64: "i64", C.c_float: "f32", C.c_double: "f64"}[ct]

    sigs = dict(_lib._SIGS)
    sigs.update(_lib.EXTRA_SIGS)
    assert len(protos) >= 55
    for ret, name, args in protos:
        if name == "fdmi_last_error":
            continue
        hdr = []
        for a in ([x.strip() for x in args.split(",")] if args.strip() not in ("", "void") else []):
            arr = re.search(r"\[\d*\]$", a) is not None
            typ = a if a.endswith("*") else re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*\s*(\[\d*\])?$", "", a).strip()
            hdr.append(from_c(typ + ("*" if arr else "")))
        res, at = sigs[name]
        assert [from_ctypes(x) for x in at] == hdr, (name, hdr, [from_ctypes(x) for x in at])
        assert from_ctypes(res) == from_c(ret), (name, ret)


def test_developer_knobs_cover_the_keys_in_use():
    """fdmi_tune_set / fdmi_tune_value: keys 0..63; unknown keys fail to set and read as 0; every key the sources read is inside the
    table (round 3: switches 32..34 were read by the kernels' launchers while the table had 32 entries -- they silently read 0)"""
    import glob
    import re
    from flash_diffusion_amd import _lib
    L = _lib.lib()
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash_diffusion_amd", "csrc")
    used = set()
    for f in glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")):
        used |= {int(m) for m in re.findall(r"fdmi_tune_get\((\d+)\)", open(f).read())}
    assert used and max(used) < 64, sorted(used)
    for k in sorted(used | {31, 63}):
        assert L.fdmi_tune_value(k) == 0
        assert L.fdmi_tune_set(k, 3) == 0 and L.fdmi_tune_value(k) == 3
        assert L.fdmi_tune_set(k, 0) == 0
    assert L.fdmi_tune_set(64, 1) != 0 and L.fdmi_tune_value(64) == 0 and L.fdmi_tune_value(-1) == 0


def test_grouped_weight_gradient_entry_point_validates_before_any_launch():
    """fdmi_wgrad_tn_group (round 5): argument errors are reported through the return code and fdmi_last_error before any device
    work -- a null array, an empty or over-long group, and a problem that breaks fdmi_wgrad_tn's operand rules (checked per
    problem) -- so a host binding can probe it on a machine without a GPU"""
    import ctypes as C
    from flash_diffusion_amd import _lib
    L = _lib.lib()
    arr = (_lib.WgradProblem * 7)()
    assert C.sizeof(_lib.WgradProblem) == 64                    # fdmi_wgrad_problem: 4 pointers / int64 + 2 int32 + pointer + int64
    assert L.fdmi_wgrad_tn_group(None, 2, None) != 0 and b"wgrad_tn_group" in L.fdmi_last_error()
    assert L.fdmi_wgrad_tn_group(C.cast(arr, C.c_void_p), 0, None) != 0
    assert L.fdmi_wgrad_tn_group(C.cast(arr, C.c_void_p), 7, None) != 0 and b"1 ... 6" in L.fdmi_last_error()
    # two problems with null operands: the per-problem check names the single-product kernel's rule
    assert L.fdmi_wgrad_tn_group(C.cast(arr, C.c_void_p), 2, None) != 0 and b"null operand" in L.fdmi_last_error()
