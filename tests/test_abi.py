"""The C-ABI library loads and exports every symbol include/bpmsm.h declares; without a GPU the
product fails loudly (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import pytest

import bulletproofs_amd as bpa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "bpmsm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bp_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(bpa.lib_path()):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "bulletproofs_amd", "csrc"), "../libbpmsm.so"])
    return bpa.lib()


def test_header_matches_binding_list():
    assert _header_symbols() == sorted(bpa.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(built_lib):
    for sym in _header_symbols():
        assert getattr(built_lib, sym) is not None, sym


def test_no_gpu_fails_loudly(built_lib):
    if built_lib.bp_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(bpa.BpError) as e:
        bpa.Context(0)
    assert e.value.code == -3
    assert b"no CPU fallback" in built_lib.bp_last_error()


def test_product_does_not_import_oracle():
    """SURVEY/prompt rule: only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "bulletproofs_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, fn


def test_rust_sys_crate_declares_every_export():
    """rust/bpmsm-sys/src/lib.rs (the FFI crate a Rust deployment binds; not compilable here: no cargo)
    declares every function include/bpmsm.h does, with the same number of parameters."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "bpmsm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    rs = open(os.path.join(root, "rust", "bpmsm-sys", "src", "lib.rs")).read()
    rs = re.sub(r"//.*", "", rs)
    cdecl = {m.group(1): m.group(2) for m in re.finditer(r"\b(bp_[a-z0-9_]+)\s*\(([^;{]*)\)\s*;", hdr)}
    rdecl = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (bp_[a-z0-9_]+)\s*\(([^;]*?)\)\s*(?:->\s*[^;]+)?;", rs, flags=re.S)}
    assert cdecl, "no declarations parsed"
    for name, args in cdecl.items():
        assert name in rdecl, name
        nc = 0 if args.strip() in ("", "void") else args.count(",") + 1
        nr = 0 if not rdecl[name].strip() else rdecl[name].count(",") + 1
        assert nc == nr, (name, nc, nr)
