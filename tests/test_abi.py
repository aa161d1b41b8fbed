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
    """only tests/, smoke_check.py (repo root) and bench.py's cpu_baseline may touch oracle/: no file of the product package mentions it."""
    pkg = os.path.join(ROOT, "videoglamm_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            assert "oracle" not in open(os.path.join(pkg, f)).read().replace("oracle/seeded.py", ""), f


def test_gemm_routing_rules():
    """vg_gemm_route launches nothing: the shape rules of the GEMM launcher on the C1 / C2 shapes."""
    import __graft_entry__ as g

    g.build()
    from videoglamm_amd import _lib, ops

    lib = _lib.load()
    route = lambda M, N, K, glu=0, win=0: lib.vg_gemm_route(M, N, K, 1, glu, win)      # bf16
    assert route(1, 4096, 4096) == 0                       # decode row: skinny kernel
    assert route(1697, 4096, 4096) == 1                    # C1 o_proj: 112 tiles of 256^2 do not fill the chip -> 128x128 kernel
    assert route(1697, 14336, 4096, glu=1) == 3            # C1 gate|up + SwiGLU: 256x256 kernel
    assert route(3361, 4096, 14336) == 3                   # C2 down
    assert route(32768, 1728, 576) == 3                    # Hiera stage 3 (K x 2 B = 1152): the phase-split 256x256 kernel since r04
    assert route(32768, 1728, 576, win=1) == 4             # ... its window-gathering form stays on the single-stage whole-line kernel
    assert route(131072, 1152, 288) == 7 and route(524288, 432, 144) == 7     # Hiera stages 1 / 2 (K = 144 / 288, >= 65536 rows): the row-register kernel (r05)
    assert route(32768, 1152, 288) == 2 and route(4096, 432, 144) == 2        # ... fewer rows: the 64-byte-step kernel
    assert lib.vg_gemm_route(524288, 432, 144, 0, 0, 0) == 2                  # ... and fp32 always
    assert route(65536, 2304, 576) == 3 and route(9232, 4096, 1024) == 3    # Hiera stage 3 fc1 and CLIP's fc1 (K = 1024): the 256x256 kernel
    # r05: 256x192 tiles where whole rounds x tile width say so — N = 576 is three exact tiles (Hiera stage 3 fc2 / proj), Llama's q|k|v at
    # M = 3361 two rounds of the narrow tile against two of the wide one, CLIP's fc2 one round of 222 narrow tiles against 148 wide ones
    assert route(65536, 576, 2304) == 6 and route(65536, 576, 576) == 6 and route(3361, 6144, 4096) == 6 and route(9232, 1024, 4096) == 6
    assert route(16384, 1152, 4608) == 6 and route(16384, 4608, 1152) == 3 and route(3361, 4096, 4096) == 3 and route(3361, 14336, 4096, glu=1) == 3
    assert lib.vg_gemm_route(1697, 4096, 4096, 0, 0, 0) == 1   # fp32 parity mode never takes the bf16-only kernels
    # split-K (ops.linear): under-filled grids with a long K only
    assert ops._splitk(213, 4096, 14336, 2) >= 2 and ops._splitk(2050, 1408, 6144, 2) == 2
    assert ops._splitk(1697, 4096, 4096, 2) == 0 and ops._splitk(213, 4096, 576, 2) == 0 and ops._splitk(8, 4096, 14336, 2) == 0
