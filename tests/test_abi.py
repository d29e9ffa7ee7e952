"""CPU tests of the drop-in boundary: libkpilqr.so loads, exports exactly what include/kpilqr.h
declares, and refuses to run without a HIP device (there is no CPU fallback in the product path)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import trajoptkp_amd
from trajoptkp_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "kpilqr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kpilqr_[a-z_A-Z0-9]+)\s*\(", src)))


def test_header_and_binding_list_agree():
    assert _header_functions() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = trajoptkp_amd.load()
    for name in _header_functions():
        assert hasattr(L, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (kpilqr_[a-z_A-Z0-9]+)", out))
    assert exported == set(_header_functions())


def test_header_compiles_as_plain_c(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "kpilqr.h"\nint main(void){ kpilqr_dims d = {7,7,3000,14,1,6,0,0}; (void)d; return KPILQR_VERSION > 0 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(c),
                           "-o", str(tmp_path / "t.o")])


def test_version():
    v = trajoptkp_amd.load().kpilqr_version()
    header = int(re.search(r"#define KPILQR_VERSION (\d+)", open(os.path.join(ROOT, "include", "kpilqr.h")).read()).group(1))
    assert v == header == 410 and v // 100 == _lib.ABI_MAJOR


def test_null_context_calls_are_harmless():
    """kpilqr_host_free(NULL, NULL) (bindings release pinned blocks after the context is gone) and the name queries on a NULL
    context return without touching a device."""
    L = trajoptkp_amd.load()
    assert L.kpilqr_host_free(None, None) == 0
    assert L.kpilqr_last_launch(None, 0) == b"" and L.kpilqr_backward_variant(None) == b""
    assert L.kpilqr_upload_residual_jacobians_const(None, None, None) == _lib.ERR_ARG


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful on a machine without a GPU")
def test_create_fails_loudly_without_device():
    with pytest.raises(trajoptkp_amd.KpilqrError) as ei:
        trajoptkp_amd.Engine(7, 7, 100, 14)
    assert ei.value.code == _lib.ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_bad_dims_rejected_before_touching_the_device():
    L = trajoptkp_amd.load()
    h = C.c_void_p()
    for bad in (dict(dof=0), dict(m=0), dict(T=1), dict(nr=0), dict(batch=0), dict(n_alpha=0)):
        kw = dict(dof=7, m=7, T=100, nr=14, batch=1, n_alpha=6, device=0, flags=0)
        kw.update(bad)
        d = _lib.Dims(**kw)
        assert L.kpilqr_create(C.byref(d), None, C.byref(h)) == _lib.ERR_ARG
        assert h.value is None
    assert L.kpilqr_create(None, None, C.byref(h)) == _lib.ERR_ARG


def test_product_package_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under trajoptkp_amd/ may import, include, call or link it."""
    pkg = os.path.join(ROOT, "trajoptkp_amd")
    bad = re.compile(r"^\s*(from\s+oracle|import\s+oracle|#\s*include\s*[\"<].*oracle)|\borc_[a-z_]+\s*\(|liboracle|libkpilqr_oracle", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not bad.search(txt), os.path.join(dirpath, f)
    out = subprocess.check_output(["ldd", _lib.LIB_PATH], text=True)
    assert "oracle" not in out


def test_every_environment_switch_the_library_reads_is_listed_in_the_header():
    """Round-4 verdict: run-time dispatch switches inside the product library must be visible at the boundary."""
    import re
    src = ""
    for f in os.listdir(os.path.join(ROOT, "trajoptkp_amd", "csrc")):
        if f.endswith((".cpp", ".hip", ".h")):
            src += open(os.path.join(ROOT, "trajoptkp_amd", "csrc", f)).read()
    read = set(re.findall(r'env_int\("(KPILQR_[A-Z0-9_]+)"', src)) | set(re.findall(r'getenv\("(KPILQR_[A-Z0-9_]+)"', src))
    header = open(os.path.join(ROOT, "include", "kpilqr.h")).read()
    assert len(read) >= 10
    missing = [n for n in sorted(read) if n not in header]
    assert not missing, missing
