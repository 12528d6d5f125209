"""CPU: libkd6d.so loads and exports every symbol include/kd6d.h declares; argument checks of the
C ABI reject bad input before touching a device (no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "kd6d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kd6d_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from kd6d import _lib
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(_lib.lib, n), "libkd6d.so does not export %s" % n
    bound = set(_lib.SIGNATURES) | {"kd6d_last_error"}
    assert set(names) <= bound, "header symbols without a ctypes binding: %s" % sorted(set(names) - bound)
    assert bound <= set(names), "bindings without a header declaration: %s" % sorted(bound - set(names))
    assert _lib.lib.kd6d_abi_version() == _lib.ABI_VERSION


def test_argument_checks_fail_loudly_without_gpu():
    from kd6d import _lib
    lib = _lib.lib
    g = _lib.ConvGeom()
    g.nseg, g.batch, g.cin, g.cout, g.ksize, g.stride, g.pad = 1, 1, 12, 16, 3, 1, 1   # cin % 8 != 0
    g.seg[0].in_h = g.seg[0].in_w = g.seg[0].out_h = g.seg[0].out_w = 4
    rc = lib.kd6d_conv2d_fwd(ctypes.byref(g), _lib.KD6D_BF16, None, None, None, None, None, 0, None, None, 0, None, 0, None, 0, None)
    assert rc == -1 and b"cin=12" in lib.kd6d_last_error()
    g.cin = 16
    rc = lib.kd6d_conv2d_fwd(ctypes.byref(g), _lib.KD6D_BF16, None, None, None, None, None, 0, None, None, 0, None, 0, None, 0, None)
    assert rc == -1 and b"null tensor" in lib.kd6d_last_error()
    g.seg[0].out_h = 5
    rc = lib.kd6d_conv2d_dgrad(ctypes.byref(g), _lib.KD6D_F32, None, None, None, 0, None)
    assert rc == -1 and b"inconsistent" in lib.kd6d_last_error()
    rc = lib.kd6d_sinkhorn_div_fwd_bwd(*([None] * 8), 4, 1.0, 0.001, 0.5, 0.5, *([None] * 6))
    assert rc == -1
    try:
        _lib.check(rc, "kd6d_sinkhorn_div_fwd_bwd")
        raise AssertionError("check() must raise")
    except _lib.Kd6dError as e:
        assert "null pointer" in str(e)
    assert lib.kd6d_sinkhorn_max_points() >= 64


def test_kernel_selection_options_table():
    """kd6d_set_option / kd6d_get_option / kd6d_reset_options (host-only state): the documented names exist with the
    documented defaults, unknown names are refused, and no KD6D_* environment variable is read by the library or the
    host package any more."""
    import pytest
    from kd6d import ops
    defaults = {"conv.halo": -1, "conv.smallc": -1, "conv.splitk": -1, "conv.tile": -1, "wgrad.small": -1,
                "bn.onepass": 1, "bn.onepass_max": 65536, "gn.onepass": 1, "sinkhorn.lanes": 1,
                "conv.halo_pairing": 1, "conv.fuse_norm": 3, "sinkhorn.dense_mfma": 1, "conv.halo_wide": 1,
                "conv.smallc_wmax": 640, "sinkhorn.dense_screen": 1, "sinkhorn.dense_rows": -1}
    ops.lib.kd6d_reset_options()
    for k, v in defaults.items():
        assert ops.get_option(k) == v, k
    with ops.option("bn.onepass_max", 1 << 40):
        assert ops.get_option("bn.onepass_max") == 1 << 40
    assert ops.get_option("bn.onepass_max") == 65536
    with pytest.raises(RuntimeError, match="unknown option"):
        ops.set_option("conv.nonsense", 1)
    header = open(os.path.join(ROOT, "include", "kd6d.h")).read()
    assert all(k in header for k in defaults)
    pkg = os.path.join(ROOT, "kd-6d-pose-adlp_amd")
    for d, _, files in os.walk(pkg):
        if os.sep + "build" in d:
            continue
        for f in files:
            if f.endswith((".hip", ".h", ".py")):
                src = open(os.path.join(d, f)).read()
                assert "getenv(" not in src, f
                assert not re.search(r"environ[^\n]*KD6D_", src), f


def test_no_float_atomics_in_the_product_sources():
    """Every cross-workgroup sum of the path is order-independent (include/kd6d.h, "reproducible reductions"): the
    sources hold no floating-point atomic -- integer atomics on fixed-point images (kd6d_det.h), ordered partials
    and slabs instead.  A float atomic slipping back in would make two executions of a step differ in the last bits."""
    csrc = os.path.join(ROOT, "kd-6d-pose-adlp_amd", "csrc")
    allowed_int = re.compile(r"atomicAdd\s*\(\s*(timeouts|reinterpret_cast<unsigned long long\*>)")
    offenders = []
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".h")):
            continue
        for no, line in enumerate(open(os.path.join(csrc, name)), 1):
            code = line.split("//")[0]
            if "atomicAdd" in code and not allowed_int.search(code):
                offenders.append(f"{name}:{no}: {line.strip()}")
            if "unsafeAtomicAdd" in code or "atomicAdd_system" in code:
                offenders.append(f"{name}:{no}: {line.strip()}")
            m = re.search(r"__hip_atomic_fetch_add\s*\(([^,]+),", code)
            if m and re.search(r"float|double", code):
                offenders.append(f"{name}:{no}: {line.strip()}")
    assert not offenders, "\n".join(offenders)
