"""The C-ABI library builds for gfx950 (hipcc cross-compiles without a GPU), loads, and exports every symbol
include/vg_kernels.h declares; the ctypes signature table covers all of them.  No compute is launched."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    src = open(os.path.join(ROOT, "include", "vg_kernels.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vg_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as g

    g.build()
    from videoglamm_amd import _lib

    lib = _lib.load()
    names = declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in vg_kernels.h but not exported by libvgkernels.so"
    assert set(names) == set(_lib.EXPORTS), set(names) ^ set(_lib.EXPORTS)
    assert lib.vg_version() >= 100
    assert lib.vg_last_error() is not None


def test_product_fails_loudly_without_the_extension(monkeypatch, tmp_path):
    import pytest
    from videoglamm_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(_lib.VGKernelError):
        _lib.load()


def test_product_never_imports_the_oracle():
    """only tests/, smoke and bench.py's cpu_baseline may touch oracle/ (the product path must not)."""
    pkg = os.path.join(ROOT, "videoglamm_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py") and f != "smoke.py":
            assert "oracle" not in open(os.path.join(pkg, f)).read().replace("oracle/seeded.py", ""), f
