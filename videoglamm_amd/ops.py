"""Host-side operator layer: torch tensors in, C-ABI calls (include/vg_kernels.h) out.

PyTorch is used for device memory and streams only; every arithmetic op on the VideoGLaMM hot path
goes through a hand-written gfx950 kernel in libvgkernels.so.  All tensors are token-major
(channels last).  There is no CPU fallback: calling any op with a CPU tensor raises.
"""
import ctypes

import torch

from . import _lib

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_QUICK_GELU, ACT_RELU, ACT_SILU, ACT_SIGMOID = 0, 1, 2, 3, 4, 5


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype} (float32 / bfloat16 only)")


def _tdt(code):
    return torch.float32 if code == F32 else torch.bfloat16


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.VGKernelError("VideoGLaMM ops run on the MI355X HIP kernels only; got a CPU tensor")
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


import contextlib as _contextlib


_capture_lock = __import__("threading").Lock()
_capture_depth = [0, True]      # [captures in flight, the collector was enabled when the first one began]


@_contextlib.contextmanager
def graph_capture(g):
    """torch.cuda.graph(g) with Python's cyclic garbage collector held off for the duration of the capture.  A capture of the decode step or of
    the SAM2 propagation creates tens of thousands of Python objects, so a generation-2 collection WILL run inside it in a long-lived process —
    and if that collection finalises device objects of an earlier clip that sit in reference cycles (an evicted CUDAGraph, events), the runtime
    calls their destructors make are illegal while a stream is capturing and abort the process (seen r03: tests/test_kernels_gpu.py's graph
    tests followed by an end-to-end capture in one process)."""
    import gc
    # the collector is process-wide state: a lock + depth counter make nested / concurrent captures restore it exactly once
    with _capture_lock:
        if _capture_depth[0] == 0:
            gc.collect()
            _capture_depth[1] = gc.isenabled()
            gc.disable()
        _capture_depth[0] += 1
    try:
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            yield
    finally:
        with _capture_lock:
            _capture_depth[0] -= 1
            if _capture_depth[0] == 0 and _capture_depth[1]:
                gc.enable()


def graph_node_counts(g):
    """(kernel nodes, memcpy nodes, other nodes) of a torch.cuda.CUDAGraph created with keep_graph=True: launches per replay, read off the
    captured hipGraph_t itself (vg_graph_node_counts)."""
    import ctypes
    counts = (ctypes.c_int64 * 3)()
    _lib.check(_lib.load().vg_graph_node_counts(ctypes.c_void_p(int(g.raw_cuda_graph())), counts), "vg_graph_node_counts")
    return tuple(int(c) for c in counts)


def _f32(t):
    if t is None:
        return None
    assert t.dtype == torch.float32 and t.is_contiguous(), "bias/gamma/norm vectors are fp32 contiguous"
    return t


def _rows2d(x):
    """View x[..., K] as 2-D [M, K] with a single row stride (no copy when possible)."""
    K = x.shape[-1]
    if x.dim() == 2 and x.stride(1) == 1:
        return x, x.shape[0], x.stride(0) if x.shape[0] > 1 else max(x.stride(0), K)
    if not x.is_contiguous():
        x = x.contiguous()
    x2 = x.view(-1, K)
    return x2, x2.shape[0], K


def _splitk(M, N, K, es):
    """K slices for a GEMM whose 128x128 tiles leave most of the 256 CUs (two workgroups each) idle while K is long:
    measured r01 (tools/bench_gemm.py): the Llama down projection at M = 213 streams 117 MB of weights in 157 us on 64 tiles."""
    import os
    if M <= 16 or N % 8 or os.environ.get("VG_GEMM_SPLITK", "1") == "0":
        return 0
    tiles = -(-M // 128) * -(-N // 128)
    steps = K * es // 128
    if tiles > int(os.environ.get("VG_GEMM_SPLITK_TILES", "256")) or steps < 32:
        return 0
    ks = min(8, 512 // tiles, steps // 16)
    return ks if ks >= 2 else 0


def _linear_splitk(lib, x, x2, M, lda, w, bias, act, gamma, residual, odt, ks, out=None):
    N, K = w.shape
    y = out if out is not None else torch.empty(*x.shape[:-1], N, dtype=odt, device=x.device)
    assert y.numel() == M * N
    ws = torch.empty(ks * M * N, dtype=torch.float32, device=x.device)
    r2, ldr = None, 0
    if residual is not None:
        r2, mr, ldr = _rows2d(residual)
        assert mr == M and r2.shape[1] == N and r2.dtype == odt
    rc = lib.vg_gemm_splitk(_p(x2), lda, _p(w), w.stride(0), _p(y), N, _p(_f32(bias)), _p(_f32(gamma)), _p(r2), ldr, M, N, K, _dt(x2), _dt(y),
                            int(act), ks, _p(ws), ws.numel(), _stream())
    _lib.check(rc, "vg_gemm_splitk")
    return y


import os as _os
_TAILSPLIT = True


def linear(x, w, bias=None, act=ACT_NONE, gamma=None, residual=None, out_dtype=None, out=None, glu=False):
    """y[..., N] = ((act(x @ w^T + bias)) * gamma) + residual ; w: [N, K] (row stride may exceed K).
    glu: w is [2N, K] = gate rows | up rows and y = silu(x @ gate^T) * (x @ up^T) — done in the GEMV epilogue for
    <= 16 rows (decode), as GEMM + vg_swiglu otherwise."""
    return _linear(x, w, bias, act, gamma, residual, out_dtype, out, glu)


def _linear(x, w, bias=None, act=ACT_NONE, gamma=None, residual=None, out_dtype=None, out=None, glu=False):
    lib = _lib.load()
    if glu and x.numel() // x.shape[-1] > 16 and ((w.shape[0] // 2) % 8 != 0 or residual is not None or gamma is not None
                                                  or (out is not None and not (out.is_contiguous() and out.data_ptr() % 16 == 0))):
        return swiglu(_linear(x, w, bias))     # shapes the fused epilogue does not take (odd widths)
    x2, M, lda = _rows2d(x)
    N, K = w.shape
    if glu:
        N //= 2
    assert x2.shape[1] == K, (x.shape, w.shape)
    assert w.stride(1) == 1
    odt = out_dtype if out_dtype is not None else x.dtype
    # a few rows past a multiple of 256 (the C2 prompt: 3361 = 13 x 256 + 33) cost the 256x256-tile kernel a whole extra row of tiles;
    # when dropping that row of tiles saves a ROUND of 256 workgroups, the tail rows go through their own (128-tile / skinny) launch:
    # Llama gate|up at M = 3361: 1568 tiles = 7 rounds -> 1456 tiles = 6 rounds + a 33-row launch
    rem = M % 256
    if (_TAILSPLIT and M > 1024 and 0 < rem <= 64 and x2.dtype == torch.bfloat16 and x2.is_contiguous() and (out is None or out.is_contiguous())
            and lib.vg_gemm_route(M, N, K, BF16, 1 if glu else 0, 0) == 3):
        ntw = -(-N // (128 if glu else 256))
        if -(-((M // 256) * ntw) // 256) < -(-((M // 256 + 1) * ntw) // 256):
            M0 = M - rem
            y = out if out is not None else torch.empty(*x.shape[:-1], N, dtype=odt, device=x.device)
            y2 = y.view(M, N)
            r2 = None if residual is None else residual.reshape(M, N)
            for a, b_ in ((0, M0), (M0, M)):
                _linear(x2[a:b_], w, bias, act, gamma, None if r2 is None else r2[a:b_], out_dtype, y2[a:b_], glu)
            return y
    if (out is not None and out.dim() == 3 and not out.is_contiguous() and out.stride(2) == 1 and x.dim() == 3 and x.is_contiguous()
            and residual is None and not glu and out.shape[:2] == x.shape[:2]):
        # [B, rows, N] rows of a larger buffer (the memory bank's slot of each object): one batched launch, batch stride = out.stride(0)
        B, Mb = x.shape[0], x.shape[1]
        assert out.dtype == odt and out.shape[2] == N and _dt(w) == _dt(x)
        rc = lib.vg_gemm(_p(x), K, Mb * K, _p(w), w.stride(0), 0, _p(out), out.stride(1), out.stride(0), _p(_f32(bias)), _p(_f32(gamma)),
                         None, 0, 0, Mb, N, K, B, _dt(x), _dt(out), act, 0, _stream())
        _lib.check(rc, "vg_gemm(batched rows)")
        return out
    ks = _splitk(M, N, K, x2.element_size()) if (not glu and (out is None or (out.is_contiguous() and out.dtype == odt))) else 0
    if ks:
        return _linear_splitk(lib, x, x2, M, lda, w, bias, act, gamma, residual, odt, ks, out)
    if out is None:
        out = torch.empty(*x.shape[:-1], N, dtype=odt, device=x.device)
    o2, Mo, ldc = _rows2d(out)
    assert Mo == M and out.dtype == odt and out.shape[-1] == N, (tuple(out.shape), M, N, out.dtype, odt)      # (a wider / narrower `out` would be written with a wrong row stride)
    assert o2.data_ptr() == out.data_ptr() and (o2.is_contiguous() or out.dim() == 2), "out must be contiguous or a 2-D row-strided view"
    r2, ldr = None, 0
    if residual is not None:
        assert residual.dtype == odt
        r2, Mr, ldr = _rows2d(residual)
        assert Mr == M and r2.shape[1] == N
    assert _dt(w) == _dt(x2)
    rc = lib.vg_gemm(_p(x2), lda, 0, _p(w), w.stride(0), 0, _p(o2), ldc, 0, _p(_f32(bias)), _p(_f32(gamma)),
                     _p(r2), ldr, 0, M, N, K, 1, _dt(x2), _dt(out), act, int(bool(glu)), _stream())
    _lib.check(rc, "vg_gemm")
    return out


def linear_rows(x, w, bias=None, act=ACT_NONE, residual=None, out=None, ln=None, add=None, rope=None, force_fused=False):
    """y = act(pro(x) @ w^T + bias) [RoPE] [+ residual] on short rows, ONE launch in bf16 (vg_gemm_rows) while the problem is small — the fp32 parity
    mode, shapes the kernel does not take and large row counts (see below; force_fused overrides) run the same arithmetic as the separate launches.
    ln = (weight, bias, eps): LayerNorm over the last dim first;  add = a2 [rows2, K]: x + a2, a2 repeated over blocks of rows2 rows;
    rope = (cos, sin, cols, ch, rows_per_block, r0, r1, grid): axial RoPE (rope_axial_heads_) on the first `cols` output columns, heads of `ch`
    channels, rows [r0, r1) of every block of rows_per_block rows, token = (row - r0) % grid.  out: optional [.., N] destination (row stride free)."""
    lib = _lib.load()
    N, K = w.shape
    sA = 0
    if x.dim() == 3 and x.stride(2) == 1 and not x.is_contiguous() and x.dtype == torch.bfloat16:
        # a strided [B, rows, K] view (the memory bank's valid rows): addressed in place by block stride
        x2, M, lda, sA = x, x.shape[0] * x.shape[1], x.stride(1), x.stride(0)
        assert rope is None or rope[4] == x.shape[1]
    else:
        x2, M, lda = _rows2d(x)
    assert x2.shape[-1] == K and w.stride(1) == 1 and not (ln is not None and add is not None)
    fused = x2.dtype == torch.bfloat16 and K in (64, 128, 192, 256) and N % 64 == 0 and (rope is None or (act == ACT_NONE and residual is None)) \
        and (act == ACT_NONE or residual is None)
    if fused and ln is not None and not force_fused:
        # the 64 x 64-tile kernel pays while the problem is small (every workgroup re-stages W, and the LayerNorm is redone per column group): measured
        # r05 (tools/lab/rows_bench.py, K = 256): norm -> linear -> ReLU at M = 4096: N = 1024 17.9 us fused / 25.5 separate, N = 2048 29.0 / 25.2;
        # M = 32768 (8 objects), N = 2048: 158 / 69; with the RoPE epilogue the separate path has a third launch: M = 8192, N = 768 24.6 / 33.7,
        # M = 32768: 80.5 / 48.5.  (The A + A2 prologue of the memory keys replaces nine launches and always pays.)
        tiles = -(-M // 64) * (N // 64)
        fused = tiles <= 2048 if rope is not None else tiles < 2048
    if not fused:
        h = x
        if ln is not None:
            h = layernorm(x, ln[0], ln[1], ln[2])
        elif add is not None:
            h = axpby(x.reshape(-1, add.shape[0], K), add, 1.0, 1.0).view(x.shape)
        y = _linear(h, w, bias, act, None, residual, None, out if rope is None else None)
        if rope is not None:
            cos, sin, cols, ch, rpb, r0, r1, grid = rope
            heads, B = cols // ch, M // rpb
            yv = y.view(B, rpb, N)
            if y.dtype == torch.bfloat16 and r0 == 0:
                rope_axial_heads_(yv, heads, cos, sin, r1, grid)        # (the first r1 rows of every block, strided heads: one launch)
            else:
                # (parity mode only: the strided head slices go through a contiguous copy — torch plumbing around vg_rope_axial)
                part = yv[:, r0:r1, :cols].reshape(B, r1 - r0, heads, ch).permute(0, 2, 1, 3).reshape(B * heads, r1 - r0, ch).clone(memory_format=torch.contiguous_format)
                rope_axial_(part, cos, sin, r1 - r0, grid)
                yv[:, r0:r1, :cols] = part.view(B, heads, r1 - r0, ch).permute(0, 2, 1, 3).reshape(B, r1 - r0, cols)
            if out is not None:
                out.copy_(y.view(out.shape))
                y = out
        return y
    if out is None:
        out = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
    o2, Mo, ldc = _rows2d(out)
    assert o2.data_ptr() == out.data_ptr(), "out must be contiguous or a 2-D row-strided view (anything else would be written through a copy)"
    assert Mo == M and out.dtype == x.dtype and out.shape[-1] == N
    r2, ldr = None, 0
    if residual is not None:
        r2, Mr, ldr = _rows2d(residual)
        assert Mr == M and r2.shape[1] == N and residual.dtype == x.dtype
    lw = lb = a2 = None
    eps, lda2, rows2 = 0.0, 0, 0
    if ln is not None:
        lw, lb, eps = _f32(ln[0]), _f32(ln[1]), float(ln[2])
        assert lw.numel() == K and lb.numel() == K
    if add is not None:
        a2 = add
        assert a2.dim() == 2 and a2.shape[1] == K and a2.stride(1) == 1 and a2.dtype == x.dtype and M % a2.shape[0] == 0
        lda2, rows2 = a2.stride(0), a2.shape[0]
    cs = sn = None
    cols = ch = rpb = r0 = r1 = grid = 0
    if rope is not None:
        cs, sn, cols, ch, rpb, r0, r1, grid = rope
        cs, sn = _f32(cs), _f32(sn)
        assert M % rpb == 0 and cs.shape[1] * 2 == ch and cs.shape[0] >= grid and cs.is_contiguous() and sn.is_contiguous()
    if sA:
        rpb = x.shape[1]
    rc = lib.vg_gemm_rows(_p(x2), lda, _p(w), w.stride(0), _p(o2), ldc, _p(_f32(bias)), _p(r2), ldr, M, N, K, act, _p(lw), _p(lb), eps,
                          _p(a2), lda2, rows2, _p(cs), _p(sn), int(cols), int(ch), int(rpb), int(sA), int(r0), int(r1), int(grid), BF16, _stream())
    _lib.check(rc, "vg_gemm_rows")
    return out


def mlp_rows(x, ln, w1, b1, w2, b2, force=False):
    """y = x + w2 @ gelu(w1 @ LayerNorm(x) + b1) + b2 — a pre-norm MLP block (Hiera's x + mlp(norm2(x))); ONE launch at the narrow bf16 width
    vg_mlp_rows is routed for (C = 144), the separate launches otherwise (wider rows, the fp32 parity mode).  ln = (weight, bias, eps).
    force: run the fused kernel on any width it has an instantiation for (C = 288: correct, measured slower than the three launches)."""
    lib = _lib.load()
    C, H = x.shape[-1], w1.shape[0]
    ok = lib.vg_mlp_rows_supported(C, H) or (force and C in (144, 288) and H % 32 == 0)
    if not (x.dtype == torch.bfloat16 and ok and w1.shape == (H, C) and w2.shape == (C, H)
            and w1.is_contiguous() and w2.is_contiguous()):
        h = linear_ln(x, ln, w1, b1, ACT_GELU)          # (one launch at stage 2's width: vg_gemm_ln; norm + linear otherwise)
        return linear(h, w2, b2, residual=x)
    x2, M, ldx = _rows2d(x)
    out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    rc = lib.vg_mlp_rows(_p(x2), ldx, _p(out), C, _p(_f32(ln[0])), _p(_f32(ln[1])), float(ln[2]), _p(w1), _p(_f32(b1)), _p(w2), _p(_f32(b2)),
                         M, C, H, BF16, _stream())
    _lib.check(rc, "vg_mlp_rows")
    return out


def linear_ln(x, ln, w, bias=None, act=ACT_NONE, window=None):
    """act(LayerNorm(x) @ w^T + bias); ln = (weight, bias, eps).  window = (B, H, W, ws): x is image-order [B,H,W,K] and the output window-order
    [Bw, ws*ws, N] (window_partition behind the norm, as linear_window's gather).  ONE launch where the row-register kernel is built for the shape
    (vg_gemm_ln: bf16, K = 144 / 288, >= 65536 rows — Hiera stages 1 and 2), the norm and the projection as two launches otherwise."""
    lib = _lib.load()
    N, K = w.shape
    x2, M, lda = _rows2d(x)
    if window is not None:
        B, H, W, ws = window
        Bw = B * (-(-H // ws)) * (-(-W // ws))
        M = Bw * ws * ws
    fused = (x.dtype == torch.bfloat16 and x.is_cuda and w.dtype == x.dtype and w.stride(1) == 1 and act in (ACT_NONE, ACT_GELU)
             and lib.vg_gemm_route(M, N, K, BF16, 0, 1 if window is not None else 0) == 7)
    if not fused:
        xn = layernorm(x, ln[0], ln[1], ln[2])
        return linear_window(xn, w, bias, *window, scatter=False, act=act) if window is not None else linear(xn, w, bias, act=act)
    out = torch.empty((Bw, ws * ws, N) if window is not None else x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
    o2, _, ldc = _rows2d(out)
    wa = window if window is not None else (0, 0, 0, 0)
    rc = lib.vg_gemm_ln(_p(x2), lda, _p(w), w.stride(0), _p(o2), ldc, _p(_f32(bias)), _p(_f32(ln[0])), _p(_f32(ln[1])), float(ln[2]), M, N, K, act,
                        1 if window is not None else 0, wa[0], wa[1], wa[2], wa[3],
                        _p(_zero_row(K, x.dtype, x.device)) if window is not None else None, BF16, _stream())
    if rc == -3:      # VG_ERR_UNSUPPORTED (include/vg_kernels.h)
        # the launcher's own eligibility (alignment, output stride, LDS) is stricter than the route query: take the two-launch form
        xn = layernorm(x, ln[0], ln[1], ln[2])
        return linear_window(xn, w, bias, *window, scatter=False, act=act) if window is not None else linear(xn, w, bias, act=act)
    _lib.check(rc, "vg_gemm_ln")
    return out


def mlp3_pack(w):
    """[G, out, in] nn.Linear weights -> vg_mlp3_grouped's fragment order [G, ceil(out / 32), in / 16, 64, 8] (lane = 32 * k-half + row; rows past `out` zero)."""
    G, n_out, n_in = w.shape
    T = -(-n_out // 32)
    wp = torch.zeros(G, T * 32, n_in, dtype=w.dtype, device=w.device)
    wp[:, :n_out] = w
    return wp.view(G, T, 32, n_in // 16, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous().view(G, T, n_in // 16, 64, 8)


def mlp3_grouped(x, G, w0, b0, w1, b1, w2, b2, out, sigmoid_mask=0):
    """G three-layer MLP heads in one launch (vg_mlp3_grouped).  x: a bf16 [R, >= G, K] view (head g reads x[:, g, :]); w0, w1, w2: mlp3_pack of the stacked
    [G, Hd, K], [G, Hd, Hd], [G, No, Hd] bf16 weights, b0 / b1 / b2 fp32 [G, Hd] / [G, Hd] / [G, No]; out: a bf16 / fp32 [R, G, >= No] view whose leading No
    columns are written."""
    lib = _lib.load()
    R, K = x.shape[0], x.shape[2]
    Hd, No = b0.shape[1], b2.shape[1]
    assert x.dtype == torch.bfloat16 and x.stride(2) == 1 and out.stride(2) == 1 and x.shape[1] >= G and out.shape[1] >= G and out.shape[2] >= No
    assert w0.shape == (G, -(-Hd // 32), K // 16, 64, 8) and w1.shape == (G, -(-Hd // 32), Hd // 16, 64, 8) and w2.shape == (G, -(-No // 32), Hd // 16, 64, 8)
    assert all(t.is_contiguous() for t in (w0, b0, w1, b1, w2, b2)) and b0.shape == (G, Hd) and b1.shape == (G, Hd) and b2.shape == (G, No)
    rc = lib.vg_mlp3_grouped(_p(x), x.stride(0), x.stride(1), _p(w0), _p(b0), _p(w1), _p(b1), _p(w2), _p(b2), _p(out), out.stride(0), out.stride(1),
                             _dt(out), G, R, K, Hd, No, int(sigmoid_mask), _stream())
    _lib.check(rc, "vg_mlp3_grouped")
    return out


def heads_blockdiag(x, TP):
    """[N, nt, 128] -> block-diagonal [N * 8 * TP, 128]: row (h, t) holds head h's 16 channels of token t, zeros elsewhere (vg_heads_blockdiag)."""
    x = x.contiguous()
    N, nt, C = x.shape
    assert C == 128 and nt <= TP
    out = torch.empty(N * 8 * TP, 128, dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().vg_heads_blockdiag(_p(x), _p(out), N, nt, TP, 0, _dt(x), _stream()), "vg_heads_blockdiag")
    return out


def heads_blockdiag_gather(full, N, nt, TP):
    """the inverse read: full [N * 8 * TP, 128] (rows (h, t)) -> [N, nt, 128], head h's channels of token t from row (h, t)."""
    full = full.contiguous()
    assert full.numel() == N * 8 * TP * 128
    out = torch.empty(N, nt, 128, dtype=full.dtype, device=full.device)
    _lib.check(_lib.load().vg_heads_blockdiag(_p(full), _p(out), N, nt, TP, 1, _dt(full), _stream()), "vg_heads_blockdiag")
    return out


_ZROWS = {}


def _zero_row(K, dtype, device):
    key = (dtype, str(device))
    z = _ZROWS.get(key)
    if z is None or z.numel() < K:
        z = _ZROWS[key] = torch.zeros(max(K, 8192), dtype=dtype, device=device)
    return z


def linear_window(x, w, bias, B, H, W, ws, scatter, act=ACT_NONE, gamma=None, residual=None):
    """Hiera's windowed projections with the partition folded into the GEMM (vg_gemm_window).
    scatter=False: x image-order [B,H,W,K] -> window-order [Bw, ws*ws, N]   (window_partition + linear)
    scatter=True : x window-order [Bw, ws*ws, K] -> image-order [B,H,W,N] (+ residual [B,H,W,N])  (linear + unpartition + add)"""
    lib = _lib.load()
    N, K = w.shape
    nH, nW = -(-H // ws), -(-W // ws)
    Bw = B * nH * nW
    x2, Mx, lda = _rows2d(x)
    assert x2.shape[1] == K and w.stride(1) == 1 and _dt(w) == _dt(x2)
    if scatter:
        assert Mx == Bw * ws * ws, (x.shape, B, H, W, ws)
        out = torch.empty(B, H, W, N, dtype=x.dtype, device=x.device)
    else:
        assert Mx == B * H * W, (x.shape, B, H, W)
        assert residual is None
        out = torch.empty(Bw, ws * ws, N, dtype=x.dtype, device=x.device)
    o2, _, ldc = _rows2d(out)
    r2, ldr = None, 0
    if residual is not None:
        assert residual.dtype == out.dtype
        r2, Mr, ldr = _rows2d(residual)
        assert Mr == B * H * W and r2.shape[1] == N
    rc = lib.vg_gemm_window(_p(x2), lda, _p(w), w.stride(0), _p(o2), ldc, _p(_f32(bias)), _p(_f32(gamma)), _p(r2), ldr, N, K,
                            _dt(x2), _dt(out), act, 2 if scatter else 1, B, H, W, ws, _p(_zero_row(K, x.dtype, x.device)), _stream())
    _lib.check(rc, "vg_gemm_window")
    return out


def bmm_nt(a, w, out_dtype=None, shared_a=False):
    """c[b] = a[b] @ w[b]^T ; a: [B,M,K], w: [B,N,K] (or [N,K] shared) -> [B,M,N].  shared_a: a is ONE [M, >= K] matrix for every batch entry (batch
    stride 0; its leading K columns are used — a K-padded weight), w: [B,N,K]."""
    lib = _lib.load()
    a = a.contiguous()
    w = w.contiguous()
    if shared_a:
        assert a.dim() == 2 and w.dim() == 3 and a.shape[1] >= w.shape[2]
        B, (M, lda), K, sA = w.shape[0], a.shape, w.shape[2], 0
    else:
        B, M, K = a.shape
        lda, sA = K, M * K
    N = w.shape[-2]
    sW = 0 if w.dim() == 2 else N * K
    odt = out_dtype if out_dtype is not None else a.dtype
    out = torch.empty(B, M, N, dtype=odt, device=a.device)
    rc = lib.vg_gemm(_p(a), lda, sA, _p(w), K, sW, _p(out), N, M * N, None, None, None, 0, 0, M, N, K, B,
                     _dt(a), _dt(out), ACT_NONE, 0, _stream())
    _lib.check(rc, "vg_gemm(batched)")
    return out


def _causal_code(causal, window):
    """the kernel's mask code: 0 none, 1 causal, c >= 2 causal with a sliding window of c visible keys (own position included)."""
    if not causal:
        assert not window, "a sliding window needs the causal mask"
        return 0
    assert window == 0 or window >= 2, "a window of one key cannot be expressed (and is not attention)"
    return int(window) if window else 1


_SPLIT_WG_D256 = 256   # workgroups the KV split aims for at head dim 256: one 8-wave workgroup per CU (measured r02: 256 beats 384 / 512 / 1024 by 1.2-2.5x, the fp32 partials are the cost)


def attention(q, k, v, scale, causal=False, window=0):
    """q: [B,Sq,Hq,D], k/v: [B,Skv,Hkv,D] (arbitrary batch/token/head strides, D contiguous) -> [B,Sq,Hq,D].
    window > 0 (with causal): every query sees its own position and the window - 1 before it (sliding-window LLMs)."""
    lib = _lib.load()
    B, Sq, Hq, D = q.shape
    Skv, Hkv = k.shape[1], k.shape[2]
    assert q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1
    out = torch.empty(B, Sq, Hq, D, dtype=q.dtype, device=q.device)
    # few query rows against a long KV range (LLM decode, mask-decoder tokens -> image): split the KV range over
    # workgroups so the chip is filled, merge the partial softmaxes afterwards
    nsplit, ws = 1, None
    blocks = -(-Sq // (128 if q.dtype == torch.bfloat16 else 64)) * Hq * B   # workgroups without splitting
    if Skv >= 512 and blocks < 384:                                           # < 1.5 workgroups per CU: split KV
        nsplit = max(1, min(64, (_SPLIT_WG_D256 if D > 128 else 512) // blocks, Skv // 128))
    if nsplit > 1:
        ws = torch.empty(B * Hq * nsplit * Sq * (D + 2), dtype=torch.float32, device=q.device)
    rc = lib.vg_attention_splitkv(_p(q), _p(k), _p(v), _p(out), B, Hq, Hkv, Sq, Skv, D,
                                  q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                                  v.stride(0), v.stride(1), v.stride(2), out.stride(0), out.stride(1), out.stride(2),
                                  float(scale), _causal_code(causal, window), _dt(q), _p(ws), 0 if ws is None else ws.numel(), nsplit,
                                  None, _stream())
    _lib.check(rc, "vg_attention")
    return out


def attention_dv(q, k, v, scale):
    """softmax(q k^T * scale) v with values of another width than the keys: q [B,Sq,H,D], k [B,Skv,H,D], v [B,Skv,H,DV] -> [B,Sq,H,DV]
    (vg_attention_dv: SAM2's memory cross-attention on the un-projected memory, D = 256, DV = 64)."""
    lib = _lib.load()
    B, Sq, H, D = q.shape
    Skv, DV = k.shape[1], v.shape[3]
    assert k.shape == (B, Skv, H, D) and v.shape[:3] == (B, Skv, H) and q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1
    if q.dtype != torch.bfloat16:
        # fp32 parity mode: the values zero-padded to the key width through vg_attention (exact fp32 MFMA chain), the first DV columns kept —
        # the same arithmetic; the DV-wide kernel is built for bf16 only (vg_attention.hip: attention_impl)
        vp = torch.zeros(B, Skv, H, D, dtype=q.dtype, device=q.device)
        vp[..., :DV].copy_(v)
        return attention(q, k, vp, scale)[..., :DV].contiguous()
    out = torch.empty(B, Sq, H, DV, dtype=q.dtype, device=q.device)
    nsplit, ws = 1, None
    # r06: the LDS-DMA form (vg_attention_dma.hip) takes D = 256 / DV = 64 with 256-row query tiles; attn_kernel's key-split form has 128-row tiles
    dma = D == 256 and DV == 64 and Sq >= 256 and q.dtype == torch.bfloat16 and _os.environ.get("VG_ATTN_DMA", "1") != "0"
    blocks = -(-Sq // (256 if dma else 128)) * H * B
    if Skv >= 512 and blocks < 384:
        nsplit = max(1, min(64, _SPLIT_WG_D256 // blocks, Skv // 128))
    if nsplit > 1:
        ws = torch.empty(B * H * nsplit * Sq * (DV + 2), dtype=torch.float32, device=q.device)
    rc = lib.vg_attention_dv(_p(q), _p(k), _p(v), _p(out), B, H, Sq, Skv, D, DV,
                             q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                             v.stride(0), v.stride(1), v.stride(2), out.stride(0), out.stride(1), out.stride(2),
                             float(scale), _dt(q), _p(ws), 0 if ws is None else ws.numel(), nsplit, _stream())
    _lib.check(rc, "vg_attention_dv")
    return out


_WIN_SHAPES = {(256, 256), (16, 16), (64, 64), (4, 16), (16, 64)}


def window_attention(q, k, v, scale):
    """Hiera's windowed attention on the dedicated kernels (vg_window_attention): q [Bw, wq, H, D], k / v [Bw, wk, H, D] strided
    views (head dim contiguous).  Returns None for shapes those kernels do not take (the caller falls back to vg_attention)."""
    Bw, wq, H, D = q.shape
    wk = k.shape[1]
    ok = (q.dtype == torch.bfloat16 and (wq, wk) in _WIN_SHAPES and k.shape == (Bw, wk, H, D) and v.shape == k.shape
          and all(t.stride(3) == 1 for t in (q, k, v)) and (D == 72 or (wq == 256 and D in (64, 80))) and (wq != 256 or Bw <= 65535))
    if not ok:
        return None
    lib = _lib.load()
    out = torch.empty(Bw, wq, H, D, dtype=q.dtype, device=q.device)
    rc = lib.vg_window_attention(_p(q), _p(k), _p(v), _p(out), Bw, H, wq, wk, D, q.stride(0), q.stride(1), q.stride(2),
                                 k.stride(0), k.stride(1), k.stride(2), v.stride(0), v.stride(1), v.stride(2),
                                 out.stride(0), out.stride(1), out.stride(2), float(scale), _dt(q), _stream())
    _lib.check(rc, "vg_window_attention")
    return out


def attention_windows(q, k, v, scale):
    """Self-attention inside many small independent windows: q/k/v [Bw, wtok, H, D] views of one fused projection.
    Windows are packed back to back into 128-token sequences under a block-diagonal mask (vg_attention, causal = -wtok),
    so a 16-token window costs 1/8 of a query tile instead of a whole padded one.  Falls back to attention() when the
    windows do not pack (wtok >= 128, odd strides, or a window count that is not a multiple of the pack)."""
    Bw, wtok, H, D = q.shape
    out = window_attention(q, k, v, scale)
    if out is not None:
        return out
    pack = 128 // wtok if wtok > 0 else 0
    ok = (pack >= 2 and Bw % pack == 0 and k.shape == q.shape and v.shape == q.shape
          and all(t.stride(0) == wtok * t.stride(1) and t.stride(3) == 1 for t in (q, k, v)))
    if not ok:
        return attention(q, k, v, scale)
    lib = _lib.load()
    B, S = Bw // pack, pack * wtok
    out = torch.empty(Bw, wtok, H, D, dtype=q.dtype, device=q.device)
    rc = lib.vg_attention_splitkv(_p(q), _p(k), _p(v), _p(out), B, H, H, S, S, D,
                                  S * q.stride(1), q.stride(1), q.stride(2), S * k.stride(1), k.stride(1), k.stride(2),
                                  S * v.stride(1), v.stride(1), v.stride(2), S * H * D, H * D, D,
                                  float(scale), -wtok, _dt(q), None, 0, 1, None, _stream())
    _lib.check(rc, "vg_attention(windows)")
    return out


def attention_decode(q, k_cache, v_cache, pos_dev, scale, window=0):
    """One decode step against a growing KV cache: q [1,1,Hq,D]; k_cache/v_cache [max_len,Hkv,D]; the number of valid
    keys is *pos_dev + 1, read on the device (graph-replayable).  The split geometry is fixed by max_len."""
    lib = _lib.load()
    _, Sq, Hq, D = q.shape
    max_len, Hkv = k_cache.shape[0], k_cache.shape[1]
    out = torch.empty(1, Sq, Hq, D, dtype=q.dtype, device=q.device)
    nsplit = max(1, min(64, -(-max_len // 64)))   # one 64-key tile per workgroup x Hkv KV heads (GQA-folded rows)
    ws = torch.empty(Hq * nsplit * Sq * (D + 2), dtype=torch.float32, device=q.device)
    rc = lib.vg_attention_splitkv(_p(q), _p(k_cache), _p(v_cache), _p(out), 1, Hq, Hkv, Sq, max_len, D,
                                  q.stride(0), q.stride(1), q.stride(2), 0, k_cache.stride(0), k_cache.stride(1),
                                  0, v_cache.stride(0), v_cache.stride(1), out.stride(0), out.stride(1), out.stride(2),
                                  float(scale), _causal_code(True, window), _dt(q), _p(ws), ws.numel(), nsplit, _p(pos_dev), _stream())
    _lib.check(rc, "vg_attention_splitkv(decode)")
    return out


def rope_kv_append_(qkv, k_cache, v_cache, cos, sin, H, Hkv, D, pos0=0, pos_dev=None):
    """in place on the fused projection qkv [S,(H+2Hkv)*D]: rotate q, rotate k -> k_cache[pos+s], v -> v_cache[pos+s]."""
    lib = _lib.load()
    S = qkv.shape[0]
    assert qkv.stride(1) == 1 and k_cache.is_contiguous() and v_cache.is_contiguous()
    rc = lib.vg_rope_kv_append(_p(qkv), qkv.stride(0), _p(k_cache), _p(v_cache), _p(_f32(cos)), _p(_f32(sin)), S, H, Hkv, D,
                               int(pos0), _p(pos_dev), _dt(qkv), _stream())
    _lib.check(rc, "vg_rope_kv_append")
    return qkv


def decode_gemv(x, w, norm_w=None, eps=0.0, residual=None, glu=False, out_dtype=None, out=None):
    """Decode-step projection of ONE row: y[N] = rmsnorm(x; norm_w, eps) (or x) @ w^T, optional SwiGLU over w = [gate|up]
    rows and optional residual — HF LlamaRMSNorm + nn.Linear (+ LlamaMLP act) in one launch (vg_decode_gemv)."""
    lib = _lib.load()
    K = x.shape[-1]
    assert x.numel() == K and x.is_contiguous() and w.stride(1) == 1 and w.shape[1] == K
    N = w.shape[0] // 2 if glu else w.shape[0]
    odt = out_dtype or x.dtype
    y = out if out is not None else torch.empty(1, N, dtype=odt, device=x.device)
    assert y.is_contiguous() and y.numel() == N
    if residual is not None:
        assert residual.is_contiguous() and residual.numel() == N and residual.dtype == y.dtype
    rc = lib.vg_decode_gemv(_p(x), _p(w), w.stride(0), _p(y), _p(None if norm_w is None else _f32(norm_w)), float(eps),
                            _p(residual), N, K, int(bool(glu)), _dt(x), _dt(y), _stream())
    _lib.check(rc, "vg_decode_gemv")
    return y


def quantize_fp8_rows(w):
    """[N,K] weight -> (uint8 [N,K] holding OCP e4m3 codes, fp32 [N] scales): w ~ scale[n] * fp8.  One-time load work (torch)."""
    wf = w.float()
    scale = (wf.abs().amax(dim=1).clamp_min(1e-12) / 448.0).contiguous()
    q = (wf / scale[:, None]).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
    return q, scale


def quantize_fp8(x):
    """activations [M,K] (bf16 / fp32, rows contiguous) -> (uint8 [M,K] e4m3 codes, fp32 [M] row scales) on the device kernel."""
    lib = _lib.load()
    x2, M, ldx = _rows2d(x)
    K = x2.shape[1]
    q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    sc = torch.empty(M, dtype=torch.float32, device=x.device)
    _lib.check(lib.vg_quantize_fp8_rows(_p(x2), ldx, _p(q), K, _p(sc), M, K, _dt(x2), _stream()), "vg_quantize_fp8_rows")
    return q, sc


def linear_f8(q, qs, w8, ws, bias=None, residual=None, glu=False, out_dtype=torch.bfloat16):
    """fp8 x fp8 GEMM with row scales on both operands (vg_gemm_f8): q uint8 [M,K] + qs [M]; w8 uint8 [N or 2N, K] + ws."""
    lib = _lib.load()
    M, K = q.shape
    N = w8.shape[0] // 2 if glu else w8.shape[0]
    assert q.dtype == torch.uint8 and w8.dtype == torch.uint8 and w8.shape[1] == K and q.is_contiguous() and w8.stride(1) == 1
    y = torch.empty(M, N, dtype=out_dtype, device=q.device)
    r2 = None
    ldr = 0
    if residual is not None:
        r2, mr, ldr = _rows2d(residual)
        assert r2.dtype == out_dtype and mr == M and r2.shape[1] == N
    rc = lib.vg_gemm_f8(_p(q), K, _p(qs), _p(w8), w8.stride(0), _p(ws), _p(y), N, _p(_f32(bias)), _p(r2), ldr,
                        M, N, K, _dt(y), 1 if glu else 0, _stream())
    _lib.check(rc, "vg_gemm_f8")
    return y


def decode_gemv_w8(x, w8, wscale, norm_w=None, eps=0.0, residual=None, glu=False, out_dtype=None, out=None):
    """decode_gemv with fp8 (e4m3) weights + per-row scales (vg_decode_gemv_w8): x bf16 [1,K]; w8 uint8 [N or 2N, K]."""
    lib = _lib.load()
    K = x.shape[-1]
    assert x.dtype == torch.bfloat16 and x.numel() == K and x.is_contiguous() and w8.dtype == torch.uint8 and w8.stride(1) == 1 and w8.shape[1] == K
    assert wscale.dtype == torch.float32 and wscale.numel() == w8.shape[0] and wscale.is_contiguous()
    N = w8.shape[0] // 2 if glu else w8.shape[0]
    odt = out_dtype or x.dtype
    y = out if out is not None else torch.empty(1, N, dtype=odt, device=x.device)
    assert y.is_contiguous() and y.numel() == N
    if residual is not None:
        assert residual.is_contiguous() and residual.numel() == N and residual.dtype == y.dtype
    rc = lib.vg_decode_gemv_w8(_p(x), _p(w8), w8.stride(0), _p(wscale), _p(y), _p(None if norm_w is None else _f32(norm_w)), float(eps),
                               _p(residual), N, K, int(bool(glu)), _dt(y), _stream())
    _lib.check(rc, "vg_decode_gemv_w8")
    return y


def decode_attention_workspace(H, Hkv, D, max_len, device):
    """zero-filled once: the workspace ends with per-KV-head arrival counters that the kernel resets itself."""
    n = _lib.load().vg_decode_attention_ws_floats(H, Hkv, D, max_len)
    if n < 0:
        raise _lib.VGKernelError(f"vg_decode_attention_ws_floats: bad shape H={H} Hkv={Hkv} D={D} max_len={max_len}")
    return torch.zeros(n, dtype=torch.float32, device=device)


def decode_attention(qkv, k_cache, v_cache, cos, sin, H, Hkv, D, pos_dev, scale, ws, window=0, keys_per_wg=0):
    """Fused RoPE + KV append + attention of the one new token (vg_decode_attention): qkv [1,(H+2Hkv)*D] -> [1,H*D].
    keys_per_wg=128: two 64-key blocks per workgroup (long caches: half the partials to merge)."""
    lib = _lib.load()
    assert qkv.is_contiguous() and k_cache.is_contiguous() and v_cache.is_contiguous() and pos_dev.dtype == torch.int32
    max_len = k_cache.shape[0]
    out = torch.empty(1, H * D, dtype=qkv.dtype, device=qkv.device)
    rc = lib.vg_decode_attention(_p(qkv), _p(k_cache), _p(v_cache), _p(_f32(cos)), _p(_f32(sin)), _p(out), H, Hkv, D, max_len,
                                 int(window), float(scale), _p(pos_dev), _p(ws), ws.numel(), int(keys_per_wg), _dt(qkv), _stream())
    _lib.check(rc, "vg_decode_attention")
    return out


def decode_rope_path(H, Hkv, D, K, dtype):
    """True when vg_decode_qkv_rope + vg_decode_attention2 cover this shape (bf16, head_dim 128: Llama-3)."""
    lib = _lib.load()
    dt = F32 if dtype == torch.float32 else BF16
    return bool(lib.vg_decode_qkv_rope_supported(H, Hkv, D, K, dt) and lib.vg_decode_attention2_supported(H, Hkv, D, dt))


def decode_qkv_rope(x, wqkv, norm_w, eps, k_cache, v_cache, rope_cs, pos_dev, H, Hkv, D, out=None):
    """RMSNorm -> q|k|v GEMV -> RoPE -> KV append of ONE row (vg_decode_qkv_rope): -> rotated q [1, H*D]; k_cache / v_cache row *pos_dev written."""
    lib = _lib.load()
    K = x.shape[-1]
    assert x.numel() == K and x.is_contiguous() and wqkv.stride(1) == 1 and wqkv.shape == ((H + 2 * Hkv) * D, K)
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.dtype == x.dtype and pos_dev.dtype == torch.int32
    assert rope_cs.dtype == torch.float32 and rope_cs.numel() == D and rope_cs.is_contiguous()
    q = out if out is not None else torch.empty(1, H * D, dtype=x.dtype, device=x.device)
    rc = lib.vg_decode_qkv_rope(_p(x), _p(wqkv), wqkv.stride(0), _p(_f32(norm_w)), float(eps), _p(q), _p(k_cache), _p(v_cache), _p(rope_cs),
                                _p(pos_dev), H, Hkv, D, K, _dt(x), _stream())
    _lib.check(rc, "vg_decode_qkv_rope")
    return q


def decode_attention2(q, k_cache, v_cache, H, Hkv, D, pos_dev, scale, ws, window=0, keys_per_wg=256):
    """attention of the one new (pre-rotated) query row against caches that already hold its key / value row (vg_decode_attention2)."""
    lib = _lib.load()
    assert q.is_contiguous() and k_cache.is_contiguous() and v_cache.is_contiguous() and pos_dev.dtype == torch.int32
    out = torch.empty(1, H * D, dtype=q.dtype, device=q.device)
    rc = lib.vg_decode_attention2(_p(q), _p(k_cache), _p(v_cache), _p(out), H, Hkv, D, k_cache.shape[0], int(window), float(scale), _p(pos_dev),
                                  _p(ws), ws.numel(), int(keys_per_wg), _dt(q), _stream())
    _lib.check(rc, "vg_decode_attention2")
    return out


def decode_advance_(pos_dev, inc, tok_dev=None, step_dev=None, forced=None, hist=None, raw=None, rope=None):
    """device-side bookkeeping of the decode loop (vg_decode_advance).  tok_dev + step_dev: record / force the emitted token and count the step;
    inc: added to *pos_dev; rope = (cos, sin, rope_cs): refresh the cos / sin row of the (new) position."""
    lib = _lib.load()
    assert pos_dev.dtype == torch.int32 and (tok_dev is None) == (step_dev is None)
    assert tok_dev is None or (tok_dev.dtype == torch.int64 and step_dev.dtype == torch.int32)
    cap = 0 if hist is None else hist.numel()
    assert (raw is None or raw.numel() == cap) and all(t is None or t.dtype == torch.int64 for t in (forced, hist, raw))
    cos, sin, rope_cs = rope if rope is not None else (None, None, None)
    hd = 0
    if rope is not None:
        hd = cos.shape[1]
        assert cos.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous() and rope_cs.numel() == 2 * hd and rope_cs.dtype == torch.float32
    rc = lib.vg_decode_advance(_p(tok_dev), _p(pos_dev), _p(step_dev), _p(forced), 0 if forced is None else forced.numel(), _p(hist), _p(raw), cap,
                               _p(cos), _p(sin), _p(rope_cs), hd, int(inc), _stream())
    _lib.check(rc, "vg_decode_advance")


def decode_step_begin(tok_dev, table, pos_dev, rope=None):
    """x = table[*tok_dev] [1, D] and (rope = (cos, sin, rope_cs)) the cos / sin row of *pos_dev: the head of a captured decode step (vg_decode_step_begin)."""
    lib = _lib.load()
    assert tok_dev.dtype == torch.int64 and table.is_contiguous() and pos_dev.dtype == torch.int32
    D = table.shape[1]
    x = torch.empty(1, D, dtype=table.dtype, device=table.device)
    cos, sin, rope_cs = rope if rope is not None else (None, None, None)
    rc = lib.vg_decode_step_begin(_p(tok_dev), _p(table), _p(x), D, _dt(table), _p(pos_dev), _p(cos), _p(sin), _p(rope_cs), 0 if rope is None else cos.shape[1], _stream())
    _lib.check(rc, "vg_decode_step_begin")
    return x


def argmax_partial(x, acc):
    """first stage of argmax over one long row into the zeroed uint64 accumulator `acc` (int64 tensor [1]); decode_step_end decodes it."""
    lib = _lib.load()
    x = x.contiguous()
    assert acc.dtype == torch.int64 and acc.numel() >= 1
    _lib.check(lib.vg_argmax_partial(_p(x), 1, x.numel(), _p(acc), _dt(x), _stream()), "vg_argmax_partial")


def decode_step_end(acc, tok_dev, pos_dev, step_dev, row, hid_all, forced=None, hist=None, raw=None):
    """the tail of a captured decode step (vg_decode_step_end): token from the argmax accumulator, final-norm row into hid_all[*pos], bookkeeping, *pos += 1."""
    lib = _lib.load()
    assert row.is_contiguous() and hid_all.is_contiguous() and row.dtype == hid_all.dtype and row.numel() == hid_all.shape[1]
    cap = 0 if hist is None else hist.numel()
    rc = lib.vg_decode_step_end(_p(acc), _p(tok_dev), _p(pos_dev), _p(step_dev), _p(forced), 0 if forced is None else forced.numel(), _p(hist), _p(raw), cap,
                                _p(row), _p(hid_all), row.numel(), _dt(row), _stream())
    _lib.check(rc, "vg_decode_step_end")


def decode_layer_roles(H, Hkv, D, hidden, inter, dtype):
    """0 / 1 / 3: which roles vg_decode_layer covers for this shape (0: use decode_attention + decode_gemv)."""
    return int(_lib.load().vg_decode_layer_roles(H, Hkv, D, hidden, inter, F32 if dtype == torch.float32 else BF16))


def decode_layer_flags(n_layers, device):
    """[n_layers, vg_decode_layer_flag_ints()] int32 zeros: one flag region per layer; the caller zero-fills ALL of it once per token."""
    return torch.zeros(n_layers, int(_lib.load().vg_decode_layer_flag_ints()), dtype=torch.int32, device=device)


def decode_layer(qkv, k_cache, v_cache, cos, sin, H, Hkv, D, pos_dev, scale, ws, flags, w_o, resid, window=0, mlp=None):
    """vg_decode_layer: attention + o_proj (+ residual) and, with mlp = (norm_w fp32, eps, W gate|up [2I,hidden], W down [hidden,I]), the
    whole MLP as roles of one launch.  flags: one row of decode_layer_flags(), zeroed since its last use.  -> the layer's output row [1, hidden] (mlp) or the post-attention row (mlp=None)."""
    lib = _lib.load()
    assert qkv.is_contiguous() and k_cache.is_contiguous() and v_cache.is_contiguous() and pos_dev.dtype == torch.int32
    assert w_o.stride(1) == 1 and resid.is_contiguous() and flags.dtype == torch.int32 and flags.is_contiguous()
    hidden = w_o.shape[0]
    o = torch.empty(1, H * D, dtype=qkv.dtype, device=qkv.device)
    y_o = torch.empty(1, hidden, dtype=qkv.dtype, device=qkv.device)
    if mlp is None:
        nw, eps, wgu, wd, act, y, inter = None, 0.0, None, None, None, None, 0
    else:
        nw, eps, wgu, wd = mlp
        inter = wd.shape[1]
        assert wgu.shape == (2 * inter, hidden) and wgu.stride(1) == 1 and wd.stride(1) == 1 and nw.dtype == torch.float32
        act = torch.empty(1, inter, dtype=qkv.dtype, device=qkv.device)
        y = torch.empty(1, hidden, dtype=qkv.dtype, device=qkv.device)
    rc = lib.vg_decode_layer(_p(qkv), _p(k_cache), _p(v_cache), _p(_f32(cos)), _p(_f32(sin)), _p(o), H, Hkv, D, k_cache.shape[0],
                             int(window), float(scale), _p(pos_dev), _p(ws), ws.numel(), _p(flags),
                             _p(w_o), w_o.stride(0), _p(resid), _p(y_o),
                             _p(nw), float(eps), _p(wgu), wgu.stride(0) if wgu is not None else 0, _p(act),
                             _p(wd), wd.stride(0) if wd is not None else 0, _p(y),
                             hidden, inter, _dt(qkv), _stream())
    _lib.check(rc, "vg_decode_layer")
    return y_o if mlp is None else y


def store_row_(src, dst, idx_dev, idx_off=0):
    """dst[*idx_dev + idx_off] = src (one row), index read on the device."""
    lib = _lib.load()
    n = src.numel()
    assert src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype
    _lib.check(lib.vg_store_row(_p(src), _p(dst), n, _p(idx_dev), int(idx_off), _dt(src), _stream()), "vg_store_row")
    return dst


def add_int_(p, v):
    lib = _lib.load()
    assert p.dtype == torch.int32
    _lib.check(lib.vg_add_int(_p(p), int(v), _stream()), "vg_add_int")
    return p


def twoway_image_update(xpe, x, u2, c2, w2t, bo, ln_w, ln_b, eps, pe, nt, TP):
    """Image side of a two-way block's image -> token cross-attention, fused (vg_twoway_image_update; the algebra is in sam2.py: _i2t_fused):
    xpe, x [Nx, P, 256] bf16 (instance n reads slot n % Nx); u2 [N, 8 TP, 256] bf16; c2 [N, 8 TP] fp32; w2t [N, 256, 8 TP] bf16; pe [P, 256]
    -> (x', x' + pe), each [N, P, 256]."""
    lib = _lib.load()
    Nx, P, C = x.shape
    N = u2.shape[0]
    assert N % Nx == 0 and xpe.shape == x.shape
    assert C == 256 and x.dtype == torch.bfloat16 and TP in (8, 16) and nt <= TP, "vg_twoway_image_update: bf16, 256 channels, <= 16 tokens"
    for t_ in (xpe, x, u2, w2t, pe):
        assert t_.is_contiguous() and t_.dtype == torch.bfloat16
    c2 = _f32(c2).contiguous()
    xo = torch.empty(N, P, C, dtype=x.dtype, device=x.device)
    xpo = torch.empty_like(xo)
    rc = lib.vg_twoway_image_update(_p(xpe), _p(x), _p(u2), _p(c2), _p(w2t), _p(_f32(bo)), _p(_f32(ln_w)), _p(_f32(ln_b)), float(eps), _p(pe),
                                    _p(xo), _p(xpo), N, Nx, P, int(nt), int(TP), _dt(x), _stream())
    _lib.check(rc, "vg_twoway_image_update")
    return xo, xpo


def mask_upscale(x, w0, b0, s1, ln_w, ln_b, eps, w1, b1, s0, hyper, es):
    """Mask decoder output upscaling + hypernetwork product, fused (vg_mask_upscale): x [N, es*es, 256], s1 [Bi, (2es)^2, 64], s0 [Bi, (4es)^2, 32]
    (instance n -> image n % Bi), hyper [N, 4, 32] -> masks fp32 [N, 4, 4es, 4es]."""
    lib = _lib.load()
    N, Bi = x.shape[0], s1.shape[0]
    assert x.dtype == torch.bfloat16 and x.shape[1:] == (es * es, 256) and s1.shape[1:] == (4 * es * es, 64) and s0.shape == (Bi, 16 * es * es, 32)
    assert hyper.shape == (N, 4, 32) and w0.shape == (256, 256) and w1.shape == (128, 64) and N % Bi == 0
    for t_ in (x, w0, s1, w1, s0):
        assert t_.is_contiguous() and t_.dtype == torch.bfloat16
    hyper = hyper.to(torch.bfloat16).contiguous()
    out = torch.empty(N, 4, 4 * es, 4 * es, dtype=torch.float32, device=x.device)
    rc = lib.vg_mask_upscale(_p(x), _p(w0), _p(_f32(b0)), _p(s1), _p(_f32(ln_w)), _p(_f32(ln_b)), float(eps), _p(w1), _p(_f32(b1)), _p(s0), _p(hyper),
                             _p(out), N, Bi, int(es), _dt(x), _stream())
    _lib.check(rc, "vg_mask_upscale")
    return out


def layernorm(x, w, b, eps, out_dtype=None):
    lib = _lib.load()
    x2, M, ldx = _rows2d(x)
    C = x2.shape[1]
    out = torch.empty(*x.shape, dtype=out_dtype or x.dtype, device=x.device)
    rc = lib.vg_layernorm(_p(x2), ldx, _p(_f32(w)), _p(_f32(b)), _p(out), C, M, C, float(eps), _dt(x2), _dt(out), _stream())
    _lib.check(rc, "vg_layernorm")
    return out


def rmsnorm(x, w, eps, out_dtype=None):
    lib = _lib.load()
    x2, M, ldx = _rows2d(x)
    C = x2.shape[1]
    out = torch.empty(*x.shape, dtype=out_dtype or x.dtype, device=x.device)
    rc = lib.vg_rmsnorm(_p(x2), ldx, _p(_f32(w)), _p(out), C, M, C, float(eps), _dt(x2), _dt(out), _stream())
    _lib.check(rc, "vg_rmsnorm")
    return out


def axpby(a, b, alpha=1.0, beta=1.0, out_dtype=None):
    """alpha*a + beta*b, b broadcast over the leading dims of a (b.numel() divides a.numel())."""
    lib = _lib.load()
    a = a.contiguous()
    out = torch.empty(a.shape, dtype=out_dtype or a.dtype, device=a.device)
    if b is None:
        rc = lib.vg_axpby(_p(a), None, _p(out), a.numel(), float(alpha), float(beta), 1, _dt(a), F32, _dt(out), _stream())
    else:
        b = b.contiguous()
        assert a.numel() % max(b.numel(), 1) == 0, (a.shape, b.shape)
        rc = lib.vg_axpby(_p(a), _p(b), _p(out), a.numel(), float(alpha), float(beta), b.numel(), _dt(a), _dt(b),
                          _dt(out), _stream())
    _lib.check(rc, "vg_axpby")
    return out


def add(a, b):
    return axpby(a, b, 1.0, 1.0)


def activation(x, act, out_dtype=None):
    lib = _lib.load()
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    _lib.check(lib.vg_activation(_p(x), _p(out), x.numel(), act, _dt(x), _dt(out), _stream()), "vg_activation")
    return out


def swiglu(gu):
    lib = _lib.load()
    gu = gu.contiguous()
    Fh = gu.shape[-1] // 2
    M = gu.numel() // (2 * Fh)
    out = torch.empty(*gu.shape[:-1], Fh, dtype=gu.dtype, device=gu.device)
    _lib.check(lib.vg_swiglu(_p(gu), _p(out), M, Fh, _dt(gu), _stream()), "vg_swiglu")
    return out


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    lib = _lib.load()
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _lib.check(lib.vg_cast(_p(x), _p(out), x.numel(), _dt(x), _dt(out), _stream()), "vg_cast")
    return out


def where_rows(cond, a, b=None, fill=0.0, out=None):
    """out[n,...] = cond[n] > 0 ? a[n,...] : (b broadcast | fill); cond fp32 [rows].  out: optional destination whose rows (one per cond entry,
    each `inner` contiguous elements) may be strided — a [rows, ...] view into a larger buffer."""
    lib = _lib.load()
    a = a.contiguous()
    cond = cond.contiguous().view(-1)
    rows = cond.numel()
    inner = a.numel() // rows
    ld_out = 0
    if out is None:
        out = torch.empty_like(a)
    else:
        assert out.dtype == a.dtype and out.shape[0] == rows and out.numel() == a.numel() and out[0].is_contiguous()
        ld_out = out.stride(0) if rows > 1 else inner
    bp = 1
    if b is not None:
        b = b.contiguous()
        assert b.dtype == a.dtype
        bp = b.numel()
    rc = lib.vg_where_rows(_p(_f32(cond)), _p(a), _p(b), _p(out), rows, inner, bp, float(fill), ld_out, _dt(a), _stream())
    _lib.check(rc, "vg_where_rows")
    return out


def mask_for_mem(x, binarize, scale, bias, out_dtype):
    lib = _lib.load()
    x = x.contiguous()
    assert x.dtype == torch.float32
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    rc = lib.vg_mask_for_mem(_p(x), _p(out), x.numel(), int(bool(binarize)), float(scale), float(bias), _dt(out), _stream())
    _lib.check(rc, "vg_mask_for_mem")
    return out


def threshold(x):
    lib = _lib.load()
    x = x.contiguous()
    assert x.dtype == torch.float32
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _lib.check(lib.vg_threshold(_p(x), _p(out), x.numel(), _stream()), "vg_threshold")
    return out


def rope_half_(x, cos, sin, pos0):
    """in place; x: [S,H,D] view (head dim contiguous); cos/sin fp32 [max_pos, D/2]."""
    lib = _lib.load()
    S, H, D = x.shape
    assert x.stride(2) == 1
    rc = lib.vg_rope_half(_p(x), x.stride(0), x.stride(1), _p(_f32(cos)), _p(_f32(sin)), S, H, D, int(pos0), _dt(x), _stream())
    _lib.check(rc, "vg_rope_half")
    return x


def rope_axial_(x, cos, sin, n_rope, n_grid):
    """in place; x: [B,N,C] contiguous; cos/sin fp32 [n_grid, C/2]."""
    lib = _lib.load()
    assert x.is_contiguous()
    B, N, C = x.shape
    rc = lib.vg_rope_axial(_p(x), _p(_f32(cos)), _p(_f32(sin)), B, N, C, int(n_rope), int(n_grid), _dt(x), _stream())
    _lib.check(rc, "vg_rope_axial")
    return x


def rope_axial_heads_(x, heads, cos, sin, n_rope, n_grid):
    """in place on the first heads * Ch columns of x [B, N, >= heads*Ch] (row / batch strides free, columns contiguous; bf16): every head of
    Ch = 2 * cos.shape[1] channels rotated with the one table — q | k of a fused q|k|v projection in one launch."""
    lib = _lib.load()
    B, N, _ = x.shape
    Ch = 2 * cos.shape[1]
    assert x.stride(2) == 1 and x.shape[2] >= heads * Ch and x.dtype == torch.bfloat16
    assert 0 <= int(n_rope) <= N and cos.shape[0] >= int(n_grid) > 0, "vg_rope_axial_heads has no row count to check n_rope against: the wrapper does"
    rc = lib.vg_rope_axial_heads(_p(x), x.stride(1), x.stride(0), _p(_f32(cos)), _p(_f32(sin)), B, heads, Ch, int(n_rope), int(n_grid), _dt(x), _stream())
    _lib.check(rc, "vg_rope_axial_heads")
    return x


def embed(ids, table):
    lib = _lib.load()
    ids = ids.contiguous().view(-1)
    assert ids.dtype == torch.int64 and table.is_contiguous()
    D = table.shape[1]
    out = torch.empty(ids.numel(), D, dtype=table.dtype, device=table.device)
    _lib.check(lib.vg_embed(_p(ids), _p(table), _p(out), ids.numel(), D, _dt(table), _stream()), "vg_embed")
    return out


def argmax(x, out=None):
    lib = _lib.load()
    x = x.contiguous()
    n = x.shape[-1]
    rows = x.numel() // n
    if out is None:
        out = torch.empty(x.shape[:-1], dtype=torch.int64, device=x.device)
    assert out.dtype == torch.int64 and out.numel() == rows and out.is_contiguous()
    _lib.check(lib.vg_argmax(_p(x), rows, n, _p(out), _dt(x), _stream()), "vg_argmax")
    return out


def multimask_select(masks, ious, tokens, mode, delta=0.05, thresh=0.98):
    """masks fp32 [N,4,h,w], ious fp32 [N,4], tokens [N,4,C] -> (mask [N,1,h,w] fp32, iou [N], token [N,C], idx [N] int32)."""
    lib = _lib.load()
    masks, ious, tokens = masks.contiguous(), ious.contiguous(), tokens.contiguous()
    assert masks.dtype == torch.float32 and ious.dtype == torch.float32
    N, _, h, w = masks.shape
    C = tokens.shape[-1]
    om = torch.empty(N, 1, h, w, dtype=torch.float32, device=masks.device)
    oi = torch.empty(N, dtype=torch.float32, device=masks.device)
    ot = torch.empty(N, C, dtype=tokens.dtype, device=masks.device)
    ox = torch.empty(N, dtype=torch.int32, device=masks.device)
    rc = lib.vg_multimask_select(_p(masks), _p(ious), _p(tokens), _p(om), _p(oi), _p(ot), _p(ox), N, h * w, C,
                                 float(delta), float(thresh), int(mode), _dt(tokens), _stream())
    _lib.check(rc, "vg_multimask_select")
    return om, oi, ot, ox


def permute5(x, dims, strides):
    """out (contiguous, shape dims[5]) gathered from x's storage at element strides[5]."""
    lib = _lib.load()
    out = torch.empty(*dims, dtype=x.dtype, device=x.device)
    d = (ctypes.c_int64 * 5)(*dims)
    s = (ctypes.c_int64 * 5)(*strides)
    _lib.check(lib.vg_permute5(_p(x), _p(out), d, s, _dt(x), _stream()), "vg_permute5")
    return out


def im2col(x, kh, kw, stride, pad, kpad):
    lib = _lib.load()
    x = x.contiguous()
    B, H, W, C = x.shape
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    out = torch.empty(B * Ho * Wo, kpad, dtype=x.dtype, device=x.device)
    _lib.check(lib.vg_im2col(_p(x), _p(out), B, H, W, C, kh, kw, stride, pad, kpad, _dt(x), _stream()), "vg_im2col")
    return out, Ho, Wo


def dwconv(x, w, bias, k):
    lib = _lib.load()
    x = x.contiguous()
    B, H, W, C = x.shape
    out = torch.empty_like(x)
    _lib.check(lib.vg_dwconv(_p(x), _p(_f32(w)), _p(_f32(bias)), _p(out), B, H, W, C, k, _dt(x), _stream()), "vg_dwconv")
    return out


def conv3s2_ln_gelu(x, w, bias, ln_w, ln_b, eps):
    """Conv2d(k 3, stride 2, pad 1) + LayerNorm2d + GELU in one pass (vg_conv3s2_ln_gelu): x [B,H,W,Cin] channels-last, w [Cout, Kpad] in
    vg_im2col's column order -> [B,(H+1)/2,(W+1)/2,Cout]; None when the channel pair is not one the kernel is built for."""
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    if (Cin, Cout) not in ((1, 4), (4, 16)) or w.dtype != x.dtype:
        return None
    lib = _lib.load()
    x = x.contiguous()
    out = torch.empty(B, (H + 1) // 2, (W + 1) // 2, Cout, dtype=x.dtype, device=x.device)
    assert w.stride(1) == 1 and w.shape[1] >= 9 * Cin
    rc = lib.vg_conv3s2_ln_gelu(_p(x), _p(w), w.stride(0), _p(_f32(bias)), _p(_f32(ln_w)), _p(_f32(ln_b)), float(eps), _p(out), B, H, W, Cin, Cout,
                                _dt(x), _stream())
    _lib.check(rc, "vg_conv3s2_ln_gelu")
    return out


def pixel_shuffle2(g, bias, B, H, W, C):
    """g: GEMM output [B*H*W, 4*C] (tap-major) -> [B, 2H, 2W, C] (+bias)."""
    lib = _lib.load()
    g = g.contiguous()
    out = torch.empty(B, 2 * H, 2 * W, C, dtype=g.dtype, device=g.device)
    _lib.check(lib.vg_pixel_shuffle2(_p(g), _p(_f32(bias)), _p(out), B, H, W, C, _dt(g), _stream()), "vg_pixel_shuffle2")
    return out


def pool2(x, is_max):
    """x: [B,H,W,C] view whose pixel stride may exceed C (e.g. the q slice of a fused qkv) -> [B,H/2,W/2,C]."""
    lib = _lib.load()
    B, H, W, C = x.shape
    assert x.stride(3) == 1 and x.stride(1) == W * x.stride(2) and x.stride(0) == H * x.stride(1)
    out = torch.empty(B, H // 2, W // 2, C, dtype=x.dtype, device=x.device)
    _lib.check(lib.vg_pool2(_p(x), _p(out), B, H, W, C, x.stride(2), int(bool(is_max)), _dt(x), _stream()), "vg_pool2")
    return out


def window_partition(x, ws):
    lib = _lib.load()
    x = x.contiguous()
    B, H, W, C = x.shape
    nH, nW = -(-H // ws), -(-W // ws)
    out = torch.empty(B * nH * nW, ws * ws, C, dtype=x.dtype, device=x.device)
    _lib.check(lib.vg_window_partition(_p(x), _p(out), B, H, W, C, ws, _dt(x), _stream()), "vg_window_partition")
    return out


def window_unpartition(win, ws, B, H, W):
    lib = _lib.load()
    win = win.contiguous()
    C = win.shape[-1]
    out = torch.empty(B, H, W, C, dtype=win.dtype, device=win.device)
    _lib.check(lib.vg_window_unpartition(_p(win), _p(out), B, H, W, C, ws, _dt(win), _stream()), "vg_window_unpartition")
    return out


def bilinear(x, Ho, Wo):
    """x: fp32 [N,Hi,Wi] -> [N,Ho,Wo], align_corners=False."""
    lib = _lib.load()
    x = x.contiguous()
    assert x.dtype == torch.float32
    N, Hi, Wi = x.shape
    out = torch.empty(N, Ho, Wo, dtype=torch.float32, device=x.device)
    _lib.check(lib.vg_bilinear(_p(x), _p(out), N, Hi, Wi, Ho, Wo, _stream()), "vg_bilinear")
    return out


def bilinear_mask(x, Ho, Wo):
    """threshold(bilinear(x, Ho, Wo)) in one pass: x fp32 [N,Hi,Wi] -> uint8 [N,Ho,Wo] (1 where the interpolated logit > 0)."""
    lib = _lib.load()
    x = x.contiguous()
    assert x.dtype == torch.float32
    N, Hi, Wi = x.shape
    out = torch.empty(N, Ho, Wo, dtype=torch.uint8, device=x.device)
    _lib.check(lib.vg_bilinear_mask(_p(x), _p(out), N, Hi, Wi, Ho, Wo, _stream()), "vg_bilinear_mask")
    return out


def upsample2_add(lateral, top, out=None):
    lib = _lib.load()
    lateral = lateral.contiguous()
    top = top.contiguous()
    B, H, W, C = top.shape
    if out is None:
        out = torch.empty_like(lateral)
    assert out.is_contiguous() and out.shape == lateral.shape and out.dtype == lateral.dtype
    _lib.check(lib.vg_upsample2_add(_p(lateral), _p(top), _p(out), B, H, W, C, _dt(top), _stream()), "vg_upsample2_add")
    return out


# ---------------------------------------------------------------- mask post-processing / evaluation counts (§8f-2, §8f-4)
def _u8(x):
    """bool / uint8 device tensor -> contiguous uint8 view (bool shares its storage)."""
    if x.dtype == torch.bool:
        x = x.view(torch.uint8)
    assert x.dtype == torch.uint8, f"masks are bool / uint8, got {x.dtype}"
    return x.contiguous()


def connected_components(mask, connectivity=8):
    """mask [..., H, W] bool/uint8 -> (labels, areas) int32 of the same shape (labels: 1 + smallest pixel index)."""
    lib = _lib.load()
    m = _u8(mask)
    H, W = m.shape[-2:]
    N = m.numel() // (H * W)
    labels = torch.empty(m.shape, dtype=torch.int32, device=m.device)
    counts = torch.empty(m.shape, dtype=torch.int32, device=m.device)
    _lib.check(lib.vg_connected_components(_p(m), _p(labels), _p(counts), N, H, W, int(connectivity), _stream()), "vg_connected_components")
    return labels, counts


def remove_small_blobs(mask, min_size):
    lib = _lib.load()
    m = _u8(mask)
    H, W = m.shape[-2:]
    N = m.numel() // (H * W)
    ws = torch.empty((2,) + tuple(m.shape), dtype=torch.int32, device=m.device)
    out = torch.empty_like(m)
    _lib.check(lib.vg_remove_small_blobs(_p(m), _p(out), _p(ws[0]), _p(ws[1]), N, H, W, int(min_size), _stream()), "vg_remove_small_blobs")
    return out


def fill_holes(scores, max_area):
    lib = _lib.load()
    x = scores.contiguous()
    assert x.dtype == torch.float32
    H, W = x.shape[-2:]
    N = x.numel() // (H * W)
    ws = torch.empty((2,) + tuple(x.shape), dtype=torch.int32, device=x.device)
    out = torch.empty_like(x)
    _lib.check(lib.vg_fill_holes(_p(x), _p(out), _p(ws[0]), _p(ws[1]), N, H, W, int(max_area), _stream()), "vg_fill_holes")
    return out


def mask_pair_counts(a, b, diagonal=False):
    """a [P, ...], b [G, ...] bool/uint8 with equal trailing shapes -> (inter, union) int64 [P, G]
    ([P] of the pairs (i, i) when diagonal)."""
    lib = _lib.load()
    a, b = _u8(a), _u8(b)
    P, G = a.shape[0], b.shape[0]
    L = a.numel() // P
    assert b.numel() // G == L, "masks of a pair must have the same number of pixels"
    shape = (P,) if diagonal else (P, G)
    inter = torch.empty(shape, dtype=torch.int64, device=a.device)
    uni = torch.empty(shape, dtype=torch.int64, device=a.device)
    _lib.check(lib.vg_mask_pair_counts(_p(a), _p(b), _p(inter), _p(uni), P, G, L, int(bool(diagonal)), _stream()), "vg_mask_pair_counts")
    return inter, uni


def boundary_counts(fg, gt, radius):
    """fg, gt [..., H, W] bool/uint8 -> int64 [N, 4] = (n_fg, n_gt, fg_match, gt_match) per image."""
    lib = _lib.load()
    fg, gt = _u8(fg), _u8(gt)
    assert fg.shape == gt.shape
    H, W = fg.shape[-2:]
    N = fg.numel() // (H * W)
    out = torch.empty((N, 4), dtype=torch.int64, device=fg.device)
    _lib.check(lib.vg_boundary_counts(_p(fg), _p(gt), _p(out), N, H, W, int(radius), _stream()), "vg_boundary_counts")
    return out


# ---------------------------------------------------------------- image pre-processing on the device (§8f-1)
def resample_u8(x, out_size, axis, bounds, coeffs):
    """one pass of Pillow's 8-bit resampler over x [N,H,W,C] uint8; bounds [out,2] / coeffs [out,ksize] int32 on device."""
    lib = _lib.load()
    x = x.contiguous()
    assert x.dtype == torch.uint8 and x.dim() == 4
    assert bounds.dtype == torch.int32 and coeffs.dtype == torch.int32 and bounds.shape == (out_size, 2) and coeffs.shape[0] == out_size
    N, H, W, C = x.shape
    shape = (N, H, out_size, C) if axis == 1 else (N, out_size, W, C)
    out = torch.empty(shape, dtype=torch.uint8, device=x.device)
    rc = lib.vg_resample_u8(_p(x), _p(out), N, H, W, C, int(out_size), int(axis), _p(bounds.contiguous()), _p(coeffs.contiguous()),
                            coeffs.shape[1], _stream())
    _lib.check(rc, "vg_resample_u8")
    return out


def resize_cv_linear_u8(x, Ho, Wo, xi=None, xa=None, yi=None, yb=None):
    """OpenCV's 8-bit INTER_LINEAR resize of x [N,H,W,C] uint8 -> [N,Ho,Wo,C]; index / tap tables int32 [Wo,2] / [Ho,2] on the device
    (None for the exact 2x down-scale, which is OpenCV's area fast path)."""
    lib = _lib.load()
    x = x.contiguous()
    assert x.dtype == torch.uint8 and x.dim() == 4
    N, H, W, C = x.shape
    for t, n in ((xi, Wo), (xa, Wo), (yi, Ho), (yb, Ho)):
        assert t is None or (t.dtype == torch.int32 and t.shape == (n, 2) and t.is_contiguous())
    out = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x.device)
    rc = lib.vg_resize_cv_linear_u8(_p(x), _p(out), N, H, W, C, int(Ho), int(Wo), _p(xi), _p(xa), _p(yi), _p(yb), _stream())
    _lib.check(rc, "vg_resize_cv_linear_u8")
    return out


def normalize_u8(x, mean, std, mode, crop=None, out_dtype=torch.float32):
    """x [N,H,W,3] uint8 -> [N,3,h,w]; crop = (top, left, h, w); mode 0: fp32 (x - mean) / std, 1: fp64 (x / 255 - mean) / std."""
    lib = _lib.load()
    x = x.contiguous()
    assert x.dtype == torch.uint8 and x.dim() == 4 and x.shape[-1] == 3
    N, H, W, _ = x.shape
    top, left, h, w = crop if crop is not None else (0, 0, H, W)
    out = torch.empty((N, 3, h, w), dtype=out_dtype, device=x.device)
    m = (ctypes.c_double * 3)(*[float(v) for v in mean])
    sd = (ctypes.c_double * 3)(*[float(v) for v in std])
    rc = lib.vg_normalize_u8(_p(x), _p(out), N, H, W, int(top), int(left), int(h), int(w), m, sd, int(mode), _dt(out), _stream())
    _lib.check(rc, "vg_normalize_u8")
    return out
