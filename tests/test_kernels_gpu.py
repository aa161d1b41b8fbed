"""Per-kernel parity: every C-ABI entry point vs the plain-PyTorch fp32 statement in _cpu_ops.py.

fp32 kernels must agree to ~1e-5 relative (MFMA f32 is an exact fp32 FMA chain, only the summation
order differs); bf16 kernels are compared against the fp32 reference evaluated on the same
bf16-rounded inputs, with a bf16-sized tolerance.
"""
import pytest
import torch

import _cpu_ops as ref

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def tol(dtype, k=1):
    if dtype == torch.float32:
        return dict(rtol=2e-4, atol=2e-5 * max(1, k) ** 0.5)
    return dict(rtol=2e-2, atol=2e-2 * max(1, k / 64) ** 0.5)


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def close(a, b, **kw):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.isfinite(a).all(), "non-finite output"
    torch.testing.assert_close(a, b, **kw)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (257, 130, 72), (1, 512, 256), (7, 256, 2048), (16, 33, 64),
                                   (4096, 384, 144), (1025, 1408, 1408), (300, 4, 32), (17, 200, 8), (513, 96, 152),
                                   (8192, 3072, 256), (5001, 4999 + 1, 72),   # these two: many tiles, ragged M and N
                                   # few tiles and a short K (64 ... 256): gemm_small64_kernel in bf16 (memory attention / mask decoder
                                   # shapes, ragged M, an N whose last 64-wide tile is partial, a single segment)
                                   (4096, 256, 256), (4096, 128, 256), (300, 72, 64), (1000, 200, 192), (65, 64, 128)])
def test_gemm(cuda, dtype, M, N, K):
    from videoglamm_amd import ops
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2)
    bias, gamma = rnd(N, seed=3), rnd(N, seed=4)
    res = rnd(M, N, dtype=dtype, seed=5)
    # every activation with LayerScale + residual (only "none" has a straight-line epilogue variant there: the others take the general path) ...
    for act in (ops.ACT_NONE, ops.ACT_GELU, ops.ACT_RELU, ops.ACT_QUICK_GELU, ops.ACT_SILU, ops.ACT_SIGMOID):
        y = ops.linear(x.to(cuda), w.to(cuda), bias.to(cuda), act, gamma.to(cuda), res.to(cuda))
        close(y, ref.linear(x, w, bias, act, gamma, res), **tol(dtype, K))
    # ... and bias + activation alone (the epilogues the model runs: GELU / quick-GELU / ReLU variants, SiLU / sigmoid through the general path)
    for act in (ops.ACT_GELU, ops.ACT_RELU, ops.ACT_QUICK_GELU, ops.ACT_SILU, ops.ACT_SIGMOID):
        y = ops.linear(x.to(cuda), w.to(cuda), bias.to(cuda), act)
        close(y, ref.linear(x, w, bias, act), **tol(dtype, K))
    y = ops.linear(x.to(cuda), w.to(cuda))
    close(y, ref.linear(x, w), **tol(dtype, K))
    if dtype == torch.bfloat16:
        y = ops.linear(x.to(cuda), w.to(cuda), out_dtype=torch.float32)
        close(y, ref.linear(x, w, out_dtype=torch.float32), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,K,F_", [(1, 328, 96), (3, 328, 96), (40, 328, 96), (300, 256, 200), (1697, 512, 1408), (129, 64, 176), (77, 72, 36)])
def test_gemm_glu_epilogue(cuda, dtype, M, K, F_):
    """silu(x.gate^T + b_g) * (x.up^T + b_u) with the SwiGLU in the epilogue: skinny kernel (M <= 16), tile kernel with the
    gate | up halves of a tile gathered by row pointers (M > 16), GEMM + vg_swiglu for widths the fused form skips (36)."""
    from videoglamm_amd import ops
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(2 * F_, K, dtype=dtype, seed=2, scale=K ** -0.5)
    bias = rnd(2 * F_, seed=3)
    for b in (None, bias):
        y = ops.linear(x.to(cuda), w.to(cuda), None if b is None else b.to(cuda), glu=True)
        assert y.shape == (M, F_)
        close(y, ref.linear(x, w, b, glu=True), **tol(dtype, K))
        if M > 16:   # same roundings as the two-kernel path, bit for bit
            assert torch.equal(y, ops.swiglu(ops.linear(x.to(cuda), w.to(cuda), None if b is None else b.to(cuda))))


def test_decode_step_kernels(cuda):
    """rope+kv-append, device-indexed attention length, row store and counter bump used by the graph-replayed decode."""
    from videoglamm_amd import ops
    H, Hkv, D, max_len, pos = 8, 2, 64, 640, 517
    for dtype in DT:
        qkv = rnd(1, (H + 2 * Hkv) * D, dtype=dtype, seed=1)
        kc, vc = rnd(max_len, Hkv, D, dtype=dtype, seed=2), rnd(max_len, Hkv, D, dtype=dtype, seed=3)
        ang = torch.arange(max_len)[:, None].float() * (1.0 / (10000 ** (torch.arange(0, D, 2).float() / D)))[None]
        cos, sin = ang.cos(), ang.sin()
        pos_dev = torch.tensor([pos], dtype=torch.int32)
        g_qkv, g_kc, g_vc, g_pos = qkv.to(cuda), kc.to(cuda), vc.to(cuda), pos_dev.to(cuda)
        ops.rope_kv_append_(g_qkv, g_kc, g_vc, cos.to(cuda), sin.to(cuda), H, Hkv, D, 0, g_pos)
        r_qkv, r_kc, r_vc = qkv.clone(), kc.clone(), vc.clone()
        ref.rope_kv_append_(r_qkv, r_kc, r_vc, cos, sin, H, Hkv, D, 0, pos_dev)
        close(g_qkv, r_qkv, **tol(dtype)); close(g_kc, r_kc, **tol(dtype)); close(g_vc, r_vc, **tol(dtype))
        q = g_qkv[:, : H * D].view(1, 1, H, D)
        o = ops.attention_decode(q, g_kc, g_vc, g_pos, D ** -0.5)
        t = dict(rtol=1e-3, atol=2e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=2e-2)
        close(o, ref.attention_decode(r_qkv[:, : H * D].view(1, 1, H, D), r_kc, r_vc, pos_dev, D ** -0.5), **t)
        hid = torch.zeros(max_len, 32, dtype=dtype, device=cuda)
        row = rnd(1, 32, dtype=dtype, seed=4)
        ops.store_row_(row.to(cuda), hid, g_pos)
        assert torch.equal(hid[pos].cpu(), row[0]) and float(hid.float().abs().sum()) == float(row.float().abs().sum())
        ops.add_int_(g_pos, 1)
        assert int(g_pos[0]) == pos + 1


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("N,K", [(96, 64), (6144, 4096), (33, 176), (4096, 14336), (257, 2048), (9216, 3072), (3072, 8192)])
def test_decode_gemv(cuda, dtype, N, K):
    """fused [RMSNorm ->] GEMV [-> SwiGLU] [+ residual] of one row vs the fp32 statement, and vs the unfused HIP path."""
    from videoglamm_amd import ops
    x = rnd(1, K, dtype=dtype, seed=1)
    w = rnd(2 * N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    nw = (1.0 + 0.1 * rnd(K, seed=3)).to(dtype).float()
    res = rnd(1, N, dtype=dtype, seed=4)
    gx, gw, gnw, gres = x.to(cuda), w.to(cuda), nw.to(cuda), res.to(cuda)
    t = tol(dtype, K)
    close(ops.decode_gemv(gx, gw[:N]), ref.linear(x, w[:N]), **t)
    close(ops.decode_gemv(gx, gw[:N], residual=gres), ref.linear(x, w[:N], residual=res), **t)
    close(ops.decode_gemv(gx, gw[:N], norm_w=gnw, eps=1e-5), ref.linear(ref.rmsnorm(x, nw, 1e-5), w[:N]), **t)
    close(ops.decode_gemv(gx, gw, norm_w=gnw, eps=1e-5, glu=True), ref.linear(ref.rmsnorm(x, nw, 1e-5), w, glu=True), **t)
    close(ops.decode_gemv(gx, gw[:N], out_dtype=torch.float32), ref.linear(x, w[:N], out_dtype=torch.float32), **t)
    # same arithmetic, same order as the stand-alone kernels: the fused launch must not move a single bit
    assert torch.equal(ops.decode_gemv(gx, gw[:N], norm_w=gnw, eps=1e-5, residual=gres),
                       ops.linear(ops.rmsnorm(gx, gnw, 1e-5), gw[:N], residual=gres))
    assert torch.equal(ops.decode_gemv(gx, gw, glu=True), ops.linear(gx, gw, glu=True))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("H,Hkv,D,max_len", [(8, 2, 64, 640), (32, 8, 128, 2048), (4, 2, 16, 128), (4, 4, 96, 256), (8, 1, 32, 192), (32, 32, 96, 2048)])
def test_decode_attention(cuda, dtype, H, Hkv, D, max_len):
    """fused RoPE + KV append + split attention + last-workgroup merge vs rope_kv_append + attention of the fp32
    statement; replayed at several positions (split boundaries, first and last slot) on ONE workspace."""
    from videoglamm_amd import ops
    kc, vc = rnd(max_len, Hkv, D, dtype=dtype, seed=2), rnd(max_len, Hkv, D, dtype=dtype, seed=3)
    ang = torch.arange(max_len)[:, None].float() * (1.0 / (10000 ** (torch.arange(0, D, 2).float() / D)))[None]
    cos, sin = ang.cos(), ang.sin()
    g_cos, g_sin = cos.to(cuda), sin.to(cuda)
    ws = ops.decode_attention_workspace(H, Hkv, D, max_len, cuda)
    t = dict(rtol=1e-3, atol=2e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=2e-2)
    for pos in (0, 1, 63, 64, 65, max_len // 2 + 5, max_len - 1):
        qkv = rnd(1, (H + 2 * Hkv) * D, dtype=dtype, seed=10 + pos)
        pos_dev = torch.tensor([pos], dtype=torch.int32)
        g_kc, g_vc = kc.to(cuda), vc.to(cuda)
        o = ops.decode_attention(qkv.to(cuda), g_kc, g_vc, g_cos, g_sin, H, Hkv, D, pos_dev.to(cuda), D ** -0.5, ws)
        r_qkv, r_kc, r_vc = qkv.clone(), kc.clone(), vc.clone()
        ref.rope_kv_append_(r_qkv, r_kc, r_vc, cos, sin, H, Hkv, D, 0, pos_dev)
        close(o, ref.attention_decode(r_qkv[:, : H * D].view(1, 1, H, D), r_kc, r_vc, pos_dev, D ** -0.5).view(1, H * D), **t)
        close(g_kc, r_kc, **tol(dtype)); close(g_vc, r_vc, **tol(dtype))
        # the appended rows are bit-identical to the stand-alone rope+append kernel
        u_qkv, u_kc, u_vc = qkv.to(cuda), kc.to(cuda), vc.to(cuda)
        ops.rope_kv_append_(u_qkv, u_kc, u_vc, g_cos, g_sin, H, Hkv, D, 0, pos_dev.to(cuda))
        if dtype == torch.bfloat16:   # (fp32: the compiler contracts x1*c - x2*s differently in the two kernels, 1 ulp)
            assert torch.equal(g_kc, u_kc) and torch.equal(g_vc, u_vc)
        else:
            close(g_kc, u_kc, rtol=1e-6, atol=1e-6)
    assert int(ws[-Hkv:].view(torch.int32).abs().sum()) == 0   # arrival counters reset themselves
    # keys_per_wg = 128 (two 64-key blocks per workgroup; honoured for the bf16 Llama-3 / Phi-3 head shapes, the default kernel elsewhere): same
    # answers at the block boundaries of both granularities, same appended rows, on its own replayed workspace
    ws2 = ops.decode_attention_workspace(H, Hkv, D, max_len, cuda)
    for pos in (0, 63, 64, 127, 128, 129, 191, 192, max_len // 2 + 5, max_len - 1):
        if pos >= max_len:
            continue
        qkv = rnd(1, (H + 2 * Hkv) * D, dtype=dtype, seed=10 + pos).to(cuda)
        pos_dev = torch.tensor([pos], dtype=torch.int32, device=cuda)
        a_kc, a_vc, b_kc, b_vc = kc.to(cuda), vc.to(cuda), kc.to(cuda), vc.to(cuda)
        o1 = ops.decode_attention(qkv, a_kc, a_vc, g_cos, g_sin, H, Hkv, D, pos_dev, D ** -0.5, ws)
        o2 = ops.decode_attention(qkv, b_kc, b_vc, g_cos, g_sin, H, Hkv, D, pos_dev, D ** -0.5, ws2, keys_per_wg=128)
        close(o2, o1, **(dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=1.6e-2, atol=2e-3)))   # fp32 merge order; one bf16 ulp
        assert torch.equal(a_kc, b_kc) and torch.equal(a_vc, b_vc)
    assert int(ws2[-Hkv:].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("dtype,H,Hkv,D,inter", [(torch.bfloat16, 32, 8, 128, 14336), (torch.bfloat16, 32, 8, 128, 11008), (torch.float32, 32, 8, 128, 14336),
                                                   (torch.bfloat16, 32, 32, 96, 8192)])
def test_decode_layer_chain(cuda, dtype, H, Hkv, D, inter):
    """vg_decode_layer (attention, o_proj and the MLP as roles of one launch that wait on device-side arrival counters) moves no bit against
    the separate launches, at split boundaries and long caches, replayed on ONE workspace; its counters end at zero, no wait gave up."""
    from videoglamm_amd import ops
    hidden, max_len = H * D, 4096
    roles = ops.decode_layer_roles(H, Hkv, D, hidden, inter, dtype)
    assert roles == (3 if (dtype == torch.bfloat16 and D == 128 and inter == 14336) else 1)
    assert ops.decode_layer_roles(8, 2, 64, 512, 1024, dtype) == 0
    kc, vc = rnd(max_len, Hkv, D, dtype=dtype, seed=2), rnd(max_len, Hkv, D, dtype=dtype, seed=3)
    ang = torch.arange(max_len)[:, None].float() * (1.0 / (10000 ** (torch.arange(0, D, 2).float() / D)))[None]
    g_cos, g_sin = ang.cos().to(cuda), ang.sin().to(cuda)
    w_o = rnd(hidden, hidden, dtype=dtype, seed=4, scale=hidden ** -0.5).to(cuda)
    w_gu = rnd(2 * inter, hidden, dtype=dtype, seed=5, scale=hidden ** -0.5).to(cuda)
    w_d = rnd(hidden, inter, dtype=dtype, seed=6, scale=inter ** -0.5).to(cuda)
    nw = (1.0 + 0.1 * rnd(hidden, seed=7)).to(cuda)
    ws_a = ops.decode_attention_workspace(H, Hkv, D, max_len, cuda)
    ws_b = ops.decode_attention_workspace(H, Hkv, D, max_len, cuda)
    flags = ops.decode_layer_flags(2, cuda)
    for pos in (0, 63, 64, 1700, 3400, max_len - 1):
        qkv = rnd(1, (H + 2 * Hkv) * D, dtype=dtype, seed=10 + pos).to(cuda)
        x = rnd(1, hidden, dtype=dtype, seed=20 + pos).to(cuda)
        pos_dev = torch.tensor([pos], dtype=torch.int32, device=cuda)
        a_kc, a_vc, b_kc, b_vc = kc.to(cuda), vc.to(cuda), kc.to(cuda), vc.to(cuda)
        o = ops.decode_attention(qkv, a_kc, a_vc, g_cos, g_sin, H, Hkv, D, pos_dev, D ** -0.5, ws_a)
        y_o = ops.decode_gemv(o, w_o, residual=x)
        flags.zero_()
        got = ops.decode_layer(qkv, b_kc, b_vc, g_cos, g_sin, H, Hkv, D, pos_dev, D ** -0.5, ws_b, flags[0], w_o, x)
        assert torch.equal(got, y_o) and torch.equal(a_kc, b_kc) and torch.equal(a_vc, b_vc)
        if roles == 3:
            y = ops.decode_gemv(ops.decode_gemv(y_o, w_gu, norm_w=nw, eps=1e-5, glu=True), w_d, residual=y_o)
            got = ops.decode_layer(qkv, b_kc, b_vc, g_cos, g_sin, H, Hkv, D, pos_dev, D ** -0.5, ws_b, flags[1], w_o, x, mlp=(nw, 1e-5, w_gu, w_d))
            assert torch.equal(got, y)
        assert int(ws_b[-Hkv:].view(torch.int32).abs().sum()) == 0 and int(flags[:, 1].sum()) == 0   # per-head tickets reset; no wait gave up


def test_gemm_small64_routing_and_batch(cuda):
    """the small-problem kernel is what runs the memory-attention projections (vg_gemm_route == 5), batched launches included."""
    from videoglamm_amd import _lib, ops
    lib = _lib.load()
    assert lib.vg_gemm_route(4096, 256, 256, 1, 0, 0) == 5 and lib.vg_gemm_route(4096, 128, 256, 1, 0, 0) == 5
    assert lib.vg_gemm_route(4096, 2048, 256, 1, 0, 0) != 5 and lib.vg_gemm_route(4096, 256, 320, 1, 0, 0) != 5    # 512 tiles; K not in {64..256}
    assert lib.vg_gemm_route(4096, 256, 256, 0, 0, 0) != 5                                                            # fp32 parity mode
    a, w = rnd(3, 200, 128, dtype=torch.bfloat16, seed=1), rnd(3, 136, 128, dtype=torch.bfloat16, seed=2)
    close(ops.bmm_nt(a.to(cuda), w.to(cuda)), ref.bmm_nt(a, w), **tol(torch.bfloat16, 128))


ROWS = [  # (B, rows, N, K, prologue, rope (cols, ch, r0, r1, grid) | None, act, residual)
    (2, 4096, 768, 256, "ln", (512, 256, 0, 4096, 4096), 0, False),     # norm1 -> q|k|v -> RoPE(q | k): memory self-attention
    (2, 4096, 256, 256, "ln", (256, 256, 0, 4096, 4096), 0, False),     # norm2 -> q -> RoPE
    (2, 3 * 1024 + 16, 1024, 64, "add", (1024, 256, 16, 3 * 1024 + 16, 1024), 0, False),   # (memory + pos) -> four layers' k -> RoPE, pointer rows first
    (1, 4096, 2048, 256, "ln", None, 3, False),                         # norm3 -> linear1 -> ReLU
    (1, 4096, 1024, 256, "ln", None, 1, False),                         # fuser: norm -> pwconv1 -> GELU
    (3, 100, 256, 64, None, None, 0, True),                             # plain + residual, ragged rows (300 = 4 tiles + 44)
    (1, 77, 64, 128, "ln", (64, 64, 5, 70, 13), 0, False),              # odd geometry: one N tile, K = 128, a token grid that wraps
    (2, 130, 128, 192, "add", None, 0, True),                           # K = 192, A + A2 and a residual
]


@pytest.mark.parametrize("case", ROWS)
def test_gemm_rows(cuda, case):
    """vg_gemm_rows (bf16): LayerNorm / A + A2 in front of, axial RoPE behind a short-row GEMM, in one launch — against the fp32 statement of the
    separate modules AND against the separate HIP launches it replaces (vg_layernorm / vg_axpby -> vg_gemm -> vg_rope_axial_heads), which it
    follows rounding for rounding: at most a few bf16 steps apart where the fp32 summation order differs."""
    from videoglamm_amd import ops
    B, rows, N, K, pro, rope, act, res = case
    dt = torch.bfloat16
    x = rnd(B, rows, K, dtype=dt, seed=1)
    w, bias = rnd(N, K, dtype=dt, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ln = (1.0 + 0.1 * rnd(K, seed=4), 0.1 * rnd(K, seed=5), 1e-5) if pro == "ln" else None
    add = rnd(rows, K, dtype=dt, seed=6) if pro == "add" else None
    r = rnd(B, rows, N, dtype=dt, seed=7) if res else None
    rp = None
    if rope is not None:
        cols, ch, r0, r1, grid = rope
        ang = rnd(grid, ch // 2, seed=8, scale=3.0)
        rp = (ang.cos(), ang.sin(), cols, ch, rows, r0, r1, grid)
    dev = lambda t: None if t is None else t.to(cuda)      # noqa: E731
    tup = lambda t: None if t is None else tuple(dev(v) if torch.is_tensor(v) else v for v in t)      # noqa: E731
    y = ops.linear_rows(dev(x), dev(w), dev(bias), act, dev(r), ln=tup(ln), add=dev(add), rope=tup(rp), force_fused=True)
    want = ref.linear_rows(x, w, bias, act, r, ln=ln, add=add, rope=rp)
    close(y, want, **tol(dt, K))
    # the launches it replaces
    h = dev(x)
    if ln is not None:
        h = ops.layernorm(h, dev(ln[0]), dev(ln[1]), ln[2])
    elif add is not None:
        h = ops.axpby(h, dev(add), 1.0, 1.0)
    sep = ops.linear(h, dev(w), dev(bias), act, None, dev(r))
    if rp is not None:
        cols, ch, r0, r1, grid = rope
        v = sep[:, r0:, :]                      # rope_axial_heads_ ropes the first n_rope rows of a (strided) view
        ops.rope_axial_heads_(v, cols // ch, dev(rp[0]), dev(rp[1]), r1 - r0, grid)
    d = (y.float() - sep.float()).abs()
    step = sep.float().abs().clamp_min(1e-2) * 2.0 ** -7      # two bf16 rounding steps
    assert float((d > step).float().mean()) < 1e-3, (float(d.max()), float((d > step).float().mean()))
    # a strided [B, rows, K] view (the memory bank's valid rows inside a larger buffer) is addressed in place
    big = torch.zeros(B, rows + 40, K, dtype=dt, device=cuda)
    big[:, 24:24 + rows] = dev(x)
    y2 = ops.linear_rows(big[:, 24:24 + rows], dev(w), dev(bias), act, dev(r), ln=tup(ln), add=dev(add), rope=tup(rp), force_fused=True)
    assert torch.equal(y2, y)
    out = torch.full((B * rows, N + 64), 7.0, dtype=dt, device=cuda)                   # a 2-D row-strided destination
    ops.linear_rows(dev(x), dev(w), dev(bias), act, dev(r), ln=tup(ln), add=dev(add), rope=tup(rp), out=out[:, 64:], force_fused=True)
    assert torch.equal(out[:, 64:], y.view(B * rows, N)) and float((out[:, :64] - 7.0).abs().max()) == 0.0
    with pytest.raises(AssertionError):                                                 # a 3-D strided destination would be written through a copy: refused
        ops.linear_rows(dev(x), dev(w), dev(bias), act, dev(r), ln=tup(ln), add=dev(add), rope=tup(rp), out=out.view(B, rows, N + 64)[..., 64:], force_fused=True)


def test_gemm_rows_large_m_route(cuda):
    """above ~2048 tiles of 64 x 64 ops.linear_rows sends the LayerNorm forms through the separate launches (vg_layernorm -> vg_gemm ->
    vg_rope_axial_heads: measured faster there) — same values as the fused kernel up to bf16 rounding steps."""
    from videoglamm_amd import ops
    dt = torch.bfloat16
    x = rnd(8, 1024, 256, dtype=dt, seed=1).to(cuda)
    w, bias = rnd(768, 256, dtype=dt, seed=2, scale=1 / 16).to(cuda), rnd(768, seed=3).to(cuda)
    ln = ((1.0 + 0.1 * rnd(256, seed=4)).to(cuda), (0.1 * rnd(256, seed=5)).to(cuda), 1e-5)
    ang = rnd(1024, 128, seed=8, scale=3.0)
    rp = (ang.cos().to(cuda), ang.sin().to(cuda), 512, 256, 1024, 0, 1024, 1024)
    w2 = rnd(2048, 256, dtype=dt, seed=6, scale=1 / 16).to(cuda)
    for kw in (dict(rope=rp), dict(act=ops.ACT_RELU)):
        ww = w if "rope" in kw else w2
        bb = bias if "rope" in kw else None
        a = ops.linear_rows(x, ww, bb, ln=ln, force_fused=True, **kw)
        b = ops.linear_rows(x.repeat(5, 1, 1), ww, bb, ln=ln, **kw)[:8]          # 40 x 1024 rows: past the tile rule -> separate launches
        d = (a.float() - b.float()).abs()
        assert float((d > b.float().abs().clamp_min(1e-2) * 2.0 ** -7).float().mean()) < 1e-3


def test_gemm_rows_fp32_composition(cuda):
    """fp32 parity mode: linear_rows runs the separate entry points (no fused fp32 kernel) — same statement, fp32 tolerance."""
    from videoglamm_amd import ops
    x, w, bias = rnd(2, 100, 64, seed=1), rnd(128, 64, seed=2, scale=0.125), rnd(128, seed=3)
    ang = rnd(16, 32, seed=8, scale=3.0)
    rp = (ang.cos(), ang.sin(), 128, 64, 100, 4, 100, 16)
    add = rnd(100, 64, seed=6)
    y = ops.linear_rows(x.to(cuda), w.to(cuda), bias.to(cuda), add=add.to(cuda), rope=tuple(v.to(cuda) if torch.is_tensor(v) else v for v in rp))
    close(y, ref.linear_rows(x, w, bias, add=add, rope=rp), **tol(torch.float32, 64))


@pytest.mark.parametrize("hw", [(256, 256), (16, 16), (7, 9)])
def test_multimask_select(cuda, hw):
    """SAM2's mask selection (mask_decoder.py:257-295, sam2_base.py:378-389): mode 0 = token 0 unless its stability score falls under the
    threshold, mode 1 = the best-IoU token of 1..3 (the many-workgroup copy kernel when the mask is a whole number of 16-byte groups)."""
    from videoglamm_amd import ops
    N, C = 5, 256
    masks = rnd(N, 4, *hw, seed=1) * 3.0
    masks[1, 0] = masks[1, 0].abs() + 1.0           # a stable token-0 mask: every pixel clear of +-delta
    ious = rnd(N, 4, seed=2)
    ious[2, 1:] = ious[2, 2]                       # ties: the first maximum wins
    toks = rnd(N, 4, C, dtype=torch.bfloat16, seed=3)
    for mode in (0, 1):
        got = ops.multimask_select(masks.to(cuda), ious.to(cuda), toks.to(cuda), mode)
        want = ref.multimask_select(masks, ious, toks, mode)
        for g, w in zip(got, want):
            assert torch.equal(g.cpu(), w), mode


@pytest.mark.parametrize("C,M", [(144, 4096 + 77), (288, 1024 + 5), (144, 100), (288, 128)])
def test_mlp_rows(cuda, C, M):
    """vg_mlp_rows (bf16): LayerNorm -> fc1 -> exact-erf GELU -> fc2 -> + x in one launch, against the fp32 statement and against the three
    HIP launches it replaces (same roundings of the LayerNorm output and of the hidden activation: a few bf16 steps apart at most)."""
    from videoglamm_amd import _lib, ops
    assert _lib.load().vg_mlp_rows_supported(144, 576) == 1 and _lib.load().vg_mlp_rows_supported(576, 2304) == 0       # (288: instantiated, not routed)
    dt = torch.bfloat16
    H = 4 * C
    x = rnd(M, C, dtype=dt, seed=1, scale=2.0) + 0.5
    ln = (1.0 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3), 1e-6)
    w1, b1 = rnd(H, C, dtype=dt, seed=4, scale=C ** -0.5), rnd(H, seed=5)
    w2, b2 = rnd(C, H, dtype=dt, seed=6, scale=H ** -0.5), rnd(C, seed=7)
    d = lambda t: t.to(cuda)      # noqa: E731
    lnd = (d(ln[0]), d(ln[1]), ln[2])
    y = ops.mlp_rows(d(x), lnd, d(w1), d(b1), d(w2), d(b2), force=True)
    close(y, ref.mlp_rows(x, ln, w1, b1, w2, b2), **tol(dt, H))
    h = ops.linear(ops.layernorm(d(x), lnd[0], lnd[1], lnd[2]), d(w1), d(b1), ops.ACT_GELU)
    sep = ops.linear(h, d(w2), d(b2), residual=d(x))
    dd = (y.float() - sep.float()).abs()
    step = sep.float().abs().clamp_min(1e-2) * 2.0 ** -7
    assert float((dd > step).float().mean()) < 1e-3, (float(dd.max()), float((dd > step).float().mean()))
    # a row-strided input view (a channel slice of a wider tensor)
    wide = torch.zeros(M, C + 16, dtype=dt, device=cuda)
    wide[:, 8:8 + C] = d(x)
    assert torch.equal(ops.mlp_rows(wide[:, 8:8 + C], lnd, d(w1), d(b1), d(w2), d(b2), force=True), y)
    # one-hot probe of the column / hidden bookkeeping: W1 = W2^T = scaled identity blocks -> y = x + gelu(LN(x)[:, perm]) summed back
    # (a swapped fragment order would move whole 4-column groups)
    eye1 = torch.zeros(H, C, dtype=dt)
    eye1[torch.arange(H), torch.arange(H) % C] = 1.0
    eye2 = torch.zeros(C, H, dtype=dt)
    eye2[torch.arange(H) % C, torch.arange(H)] = 0.25
    y2 = ops.mlp_rows(d(x), lnd, d(eye1), d(b1 * 0), d(eye2), d(b2 * 0), force=True)
    close(y2, ref.mlp_rows(x, ln, eye1, b1 * 0, eye2, b2 * 0), **tol(dt, 4))


@pytest.mark.parametrize("dtype", DT)
def test_heads_blockdiag(cuda, dtype):
    """the token side's per-head block-diagonal layout and its inverse read (vg_heads_blockdiag), against the torch.diagonal statement"""
    from videoglamm_amd import ops
    for N, nt, TP in ((3, 9, 16), (1, 7, 8), (5, 8, 8), (2, 16, 16)):
        x = rnd(N, nt, 128, dtype=dtype, seed=N + nt)
        bd = ops.heads_blockdiag(x.to(cuda), TP)
        assert torch.equal(bd.cpu(), ref.heads_blockdiag(x, TP))
        full = rnd(N * 8 * TP, 128, dtype=dtype, seed=7)
        assert torch.equal(ops.heads_blockdiag_gather(full.to(cuda), N, nt, TP).cpu(), ref.heads_blockdiag_gather(full, N, nt, TP))
        assert torch.equal(ops.heads_blockdiag_gather(bd, N, nt, TP).cpu(), x)


def test_gemm_transpose_detect(cuda):
    """A = I against an asymmetric W catches a swapped C layout (guide §3)."""
    from videoglamm_amd import ops
    n = 160
    w = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 1000.0
    y = ops.linear(torch.eye(n).to(cuda), w.to(cuda))
    close(y, w.t().contiguous(), rtol=0, atol=1e-6)


@pytest.mark.parametrize("dtype", DT)
def test_gemm_strided_views(cuda, dtype):
    from videoglamm_amd import ops
    big = rnd(300, 3 * 64, dtype=dtype, seed=7).to(cuda)
    x = big[:, 64:128]                      # row stride 192, K=64
    w = rnd(40, 64, dtype=dtype, seed=8)
    out = torch.zeros(300, 80, dtype=dtype, device=cuda)
    ops.linear(x, w.to(cuda), out=out[:, 40:])
    close(out[:, 40:], ref.linear(big.cpu()[:, 64:128], w), **tol(dtype, 64))
    assert float(out[:, :40].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DT)
def test_bmm(cuda, dtype):
    from videoglamm_amd import ops
    a, w = rnd(3, 4, 32, dtype=dtype, seed=1), rnd(3, 1000, 32, dtype=dtype, seed=2)
    close(ops.bmm_nt(a.to(cuda), w.to(cuda), out_dtype=torch.float32), ref.bmm_nt(a, w, torch.float32), **tol(dtype, 32))
    a, w = rnd(2, 70, 64, dtype=dtype, seed=3), rnd(2, 50, 64, dtype=dtype, seed=4)
    close(ops.bmm_nt(a.to(cuda), w.to(cuda)), ref.bmm_nt(a, w), **tol(dtype, 64))
    # SAM2's mask product (<= 4 hypernetwork rows x 32 channels against every upscaled pixel): the lane-per-column short-K kernel
    # (N >= 4096), ragged N, 1..4 rows, the shared-W form
    for B, M, N in ((3, 4, 4096 + 37), (2, 1, 8192), (2, 3, 5000)):
        a, w = rnd(B, M, 32, dtype=dtype, seed=5 + M), rnd(B, N, 32, dtype=dtype, seed=9 + M)
        close(ops.bmm_nt(a.to(cuda), w.to(cuda), out_dtype=torch.float32), ref.bmm_nt(a, w, torch.float32), **tol(dtype, 32))
    a, w = rnd(2, 4, 32, dtype=dtype, seed=21), rnd(4608, 32, dtype=dtype, seed=22)
    close(ops.bmm_nt(a.to(cuda), w.to(cuda)), ref.bmm_nt(a, w), **tol(dtype, 32))


ATT = [  # B, Hq, Hkv, Sq, Skv, D, causal
    (1, 2, 2, 64, 64, 72, False),     # Hiera window
    (2, 4, 4, 16, 64, 72, False),     # Hiera q-pooled window
    (1, 16, 16, 1025, 1025, 88, False),  # InternVideo2
    (2, 16, 16, 577, 577, 64, False),    # CLIP
    (1, 8, 8, 7, 4096, 16, False),    # two-way: tokens -> image
    (2, 8, 8, 4096, 9, 16, False),    # two-way: image -> tokens
    (1, 8, 8, 7, 7, 32, False),       # token self-attn
    (1, 1, 1, 1024, 1024, 256, False),   # memory self-attn
    (1, 1, 1, 1024, 2100, 256, False),   # memory cross-attn (ragged kv)
    (1, 8, 2, 333, 333, 128, True),   # Llama GQA prefill
    (1, 8, 2, 1, 700, 128, True),     # Llama decode step
    (1, 4, 4, 100, 100, 96, True),    # Phi-3 head dim
    (1, 4, 4, 33, 160, 64, True),     # causal with offset
    (1, 32, 8, 213, 1697, 128, True),  # sequence-parallel prefill chunk (world 8): few query tiles, long KV -> split-KV + merge, causal offset
    (1, 32, 8, 849, 1697, 128, True),  # the same at world 2
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cfg", ATT)
def test_attention(cuda, dtype, cfg):
    from videoglamm_amd import ops
    B, Hq, Hkv, Sq, Skv, D, causal = cfg
    q, k, v = rnd(B, Sq, Hq, D, dtype=dtype, seed=1), rnd(B, Skv, Hkv, D, dtype=dtype, seed=2), rnd(B, Skv, Hkv, D, dtype=dtype, seed=3)
    o = ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), D ** -0.5, causal)
    t = dict(rtol=1e-3, atol=2e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=2e-2)
    close(o, ref.attention(q, k, v, D ** -0.5, causal), **t)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,Sq,Skv,D,DV", [(1, 1024, 2100, 256, 64),      # ragged KV, one workgroup row per 128 queries -> split-KV + merge
                                           (8, 4096, 7 * 4096 + 64, 256, 64),     # C4's per-frame shape (8 objects, 7 memory frames + pointers): no split
                                           (2, 320, 700, 256, 64), (1, 4096, 4096 + 4, 256, 64)])
def test_attention_dv(cuda, dtype, B, Sq, Skv, D, DV):
    """vg_attention_dv (SAM2 memory cross-attention on the un-projected 64-d memory): keys of 256 dims, values / output of 64, against the fp32
    statement; and the identity the host relies on — softmax(q k^T) (M Wv^T + b) == (softmax(q k^T) M) Wv^T + b — against vg_attention on
    the projected values."""
    from videoglamm_amd import ops
    if B * Sq * Skv > 3e8 and dtype == torch.float32:
        pytest.skip("the fp32 statement of this shape is a bf16-only size")
    q, k, v = rnd(B, Sq, 1, D, dtype=dtype, seed=1), rnd(B, Skv, 1, D, dtype=dtype, seed=2), rnd(B, Skv, 1, DV, dtype=dtype, seed=3)
    v[:, Skv // 3] *= 6.0                      # a few large value rows and one key spike (forces the running-max rescale)
    k[:, Skv // 2] *= 4.0
    o = ops.attention_dv(q.to(cuda), k.to(cuda), v.to(cuda), D ** -0.5)
    assert o.shape == (B, Sq, 1, DV)
    t = dict(rtol=1e-3, atol=2e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=2e-2)
    if B * Sq * Skv <= 3e8:
        close(o, ref.attention(q, k, v, D ** -0.5), **t)
    else:                                        # the big shape: a slice of the queries of two batch entries against the statement
        for b in (0, B - 1):
            close(o[b:b + 1, 100:356], ref.attention(q[b:b + 1, 100:356], k[b:b + 1], v[b:b + 1], D ** -0.5), **t)
    if dtype == torch.float32 and B * Sq * Skv <= 3e8:
        wv, bv = rnd(D, DV, seed=4, scale=DV ** -0.5), rnd(D, seed=5)
        full = ops.attention(q.to(cuda), k.to(cuda), ops.linear(v.to(cuda), wv.to(cuda), bv.to(cuda)), D ** -0.5)
        close(ops.linear(o, wv.to(cuda), bv.to(cuda)), full.cpu(), rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("dtype", DT)
def test_attention_dv_every_tail(cuda, dtype):
    """the fp32 DV instantiation once mis-scored a key when Skv % 32 was in [28, 31] (worked around by the padded-V route in
    ops.attention_dv; the generic V fetch reads a clamped last row since).  Sweep EVERY residue of the sequence end inside a 64-key tile — the
    video branch reaches Skv = 4 P + n x 4096 with P = 1 ... 16 pointers, i.e. residues 4, 8, ..., 60, 0 — in both dtypes, with and without
    the KV split (Skv < 512: one split; >= 512: several), with a large value on the LAST key so that a dropped or duplicated tail key shows."""
    from videoglamm_amd import ops
    D, DV, Sq = 256, 64, 192
    q = rnd(1, Sq, 1, D, dtype=dtype, seed=1)
    for base in (64, 1024):
        for r in range(64):
            Skv = base + r
            k, v = rnd(1, Skv, 1, D, dtype=dtype, seed=2 + r), rnd(1, Skv, 1, DV, dtype=dtype, seed=3 + r)
            v[:, -1] += 8.0
            k[:, -1] = q[:, 7] * 0.5                    # the last key matters to at least one query
            o = ops.attention_dv(q.to(cuda), k.to(cuda), v.to(cuda), D ** -0.5)
            t = dict(rtol=1e-3, atol=3e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
            close(o, ref.attention(q, k, v, D ** -0.5), **t)


@pytest.mark.parametrize("dtype", DT)
def test_memory_encoder_spatial_kernels(cuda, dtype):
    """r04: the strip depthwise 7x7 kernel (bf16: 8 channels x 4 pixels per thread; widths that are not multiples of the strip, image borders) and
    the fused Conv2d(3, stride 2, pad 1) + LayerNorm2d + GELU stages of the mask downsampler (1 -> 4 -> 16 channels, odd sizes too)."""
    from videoglamm_amd import ops
    for (B, H, W, C) in ((2, 64, 64, 256), (1, 9, 13, 16), (3, 5, 4, 8)):
        x, w, b = rnd(B, H, W, C, dtype=dtype, seed=1), rnd(49, C, seed=2, scale=0.2), rnd(C, seed=3)
        close(ops.dwconv(x.to(cuda), w.to(cuda), b.to(cuda), 7), ref.dwconv(x, w, b, 7), **tol(dtype, 49))
    for (B, H, W, Cin, Cout) in ((2, 64, 48, 1, 4), (2, 32, 32, 4, 16), (1, 7, 9, 4, 16), (1, 1024, 1024, 1, 4)):
        x = rnd(B, H, W, Cin, dtype=dtype, seed=4)
        kp = (9 * Cin + 15) // 16 * 16
        w = torch.zeros(Cout, kp, dtype=dtype)
        w[:, : 9 * Cin] = rnd(Cout, 9 * Cin, dtype=dtype, seed=5, scale=(9 * Cin) ** -0.5)
        bias, lw, lb = rnd(Cout, seed=6), 1.0 + 0.2 * rnd(Cout, seed=7), 0.1 * rnd(Cout, seed=8)
        y = ops.conv3s2_ln_gelu(x.to(cuda), w.to(cuda), bias.to(cuda), lw.to(cuda), lb.to(cuda), 1e-6)
        assert y.shape == (B, (H + 1) // 2, (W + 1) // 2, Cout)
        close(y, ref.conv3s2_ln_gelu(x, w, bias, lw, lb, 1e-6), **(dict(rtol=1e-3, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)))
    assert ops.conv3s2_ln_gelu(rnd(1, 8, 8, 16, dtype=dtype).to(cuda), rnd(64, 144, dtype=dtype).to(cuda), None, None, None, 1e-6) is None


def test_rope_axial_heads(cuda):
    """the vectorised / strided axial RoPE (r04) against the scalar kernel's arithmetic: bit for bit on a contiguous tensor (vg_rope_axial routes bf16
    there), and on the q | k columns of a fused q|k|v row (two heads, stride 3 C) against per-head rotations; tokens past n_rope untouched."""
    from videoglamm_amd import ops
    B, N, C, grid = 2, 70, 64, 64
    cos, sin = torch.cos(rnd(grid, C // 2, seed=1)), torch.sin(rnd(grid, C // 2, seed=1))
    x = rnd(B, N, C, dtype=torch.bfloat16, seed=2)
    y = ops.rope_axial_(x.clone().to(cuda), cos.to(cuda), sin.to(cuda), 64, grid)
    close(y, ref.rope_axial_(x.clone(), cos, sin, 64, grid), rtol=1e-2, atol=1e-2)
    xf = x.float()
    a, b = xf[:, :64, 0::2], xf[:, :64, 1::2]
    exact = torch.stack([a * cos - b * sin, a * sin + b * cos], dim=-1).reshape(B, 64, C).to(torch.bfloat16)
    close(y[:, :64], exact, rtol=8e-3, atol=1e-3)           # (one bf16 ulp: the device contracts a c - b s into an fma)
    assert torch.equal(y[:, 64:].cpu(), x[:, 64:])
    qkv = rnd(B, N, 3 * C, dtype=torch.bfloat16, seed=3)
    z = ops.rope_axial_heads_(qkv.clone().to(cuda), 2, cos.to(cuda), sin.to(cuda), 64, grid).cpu()
    for h in range(2):
        part = ops.rope_axial_(qkv[..., h * C:(h + 1) * C].contiguous().to(cuda), cos.to(cuda), sin.to(cuda), 64, grid).cpu()
        assert torch.equal(z[..., h * C:(h + 1) * C], part)
    assert torch.equal(z[..., 2 * C:], qkv[..., 2 * C:])


def test_attention_fused_qkv_strides_and_spike(cuda):
    """q/k/v as strided slices of one fused projection + a key spike that forces the online-softmax rescale."""
    from videoglamm_amd import ops
    B, S, H, D = 2, 200, 4, 72
    qkv = rnd(B, S, 3, H, D, seed=11)
    qkv[0, 150, 1] *= 30.0   # late, huge key -> running max jumps at a later tile
    g = qkv.to(cuda)
    o = ops.attention(g[:, :, 0], g[:, :, 1], g[:, :, 2], D ** -0.5)
    close(o, ref.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], D ** -0.5), rtol=1e-3, atol=2e-5)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,C", [(5, 4), (1025, 1408), (64, 256), (3, 4096), (4096, 64),
                                    (40000, 288), (70001, 144), (33000, 512), (36000, 1152), (2500, 4100), (4101, 576), (300001, 576), (4100, 1408), (9232, 1024), (3361, 4096), (3361, 3072)])   # more rows than any persistent grid covers in one pass
def test_norms(cuda, dtype, rows, C):
    from videoglamm_amd import ops
    x = rnd(rows, C, dtype=dtype, seed=1) + 3.0
    w, b = rnd(C, seed=2), rnd(C, seed=3)
    close(ops.layernorm(x.to(cuda), w.to(cuda), b.to(cuda), 1e-6), ref.layernorm(x, w, b, 1e-6), **tol(dtype))
    close(ops.rmsnorm(x.to(cuda), w.to(cuda), 1e-5), ref.rmsnorm(x, w, 1e-5), **tol(dtype))
    close(ops.layernorm(x.to(cuda), w.to(cuda), b.to(cuda), 1e-6, out_dtype=torch.float32),
          ref.layernorm(x, w, b, 1e-6, out_dtype=torch.float32), **tol(dtype))
    if C % 8 == 0 and rows <= 40000:      # strided rows: the q / k slices of a fused q|k|v projection (InternVideo2's QK-RMSNorm)
        wide = rnd(rows, 3 * C, dtype=dtype, seed=4)
        g = wide.to(cuda)
        close(ops.rmsnorm(g[:, C:2 * C], w.to(cuda), 1e-6), ref.rmsnorm(wide[:, C:2 * C], w, 1e-6), **tol(dtype))


@pytest.mark.parametrize("dtype", DT)
def test_pointwise(cuda, dtype):
    from videoglamm_amd import ops
    a, b = rnd(6, 50, 32, dtype=dtype, seed=1), rnd(50, 32, dtype=dtype, seed=2)
    close(ops.axpby(a.to(cuda), b.to(cuda), 1.0, 0.1), ref.axpby(a, b, 1.0, 0.1), **tol(dtype))
    close(ops.add(a.to(cuda), a.to(cuda)), ref.add(a, a), **tol(dtype))
    close(ops.axpby(a.to(cuda), rnd(32, seed=3).to(cuda), 2.0, 1.0), ref.axpby(a, rnd(32, seed=3), 2.0, 1.0), **tol(dtype))
    odd, ob = rnd(7, 5, 3, dtype=dtype, seed=5), rnd(5, 3, dtype=dtype, seed=6)          # sizes off the 8-element vector path
    close(ops.axpby(odd.to(cuda), ob.to(cuda), 0.5, -1.5), ref.axpby(odd, ob, 0.5, -1.5), **tol(dtype))
    for act in (1, 2, 3, 4, 5):
        close(ops.activation(a.to(cuda), act), ref.activation(a, act), **tol(dtype))
    gu = rnd(9, 2 * 48, dtype=dtype, seed=4)
    close(ops.swiglu(gu.to(cuda)), ref.swiglu(gu), **tol(dtype))
    close(ops.cast(a.to(cuda), torch.float32), a.float(), rtol=0, atol=0)
    close(ops.cast(a.float().to(cuda), torch.bfloat16), a.float().to(torch.bfloat16), rtol=0, atol=0)
    cond = torch.tensor([1.0, -1.0, 0.0, 2.0, -3.0, 5.0])
    close(ops.where_rows(cond.to(cuda), a.to(cuda), None, -1024.0), ref.where_rows(cond, a, None, -1024.0), rtol=0, atol=0)
    close(ops.where_rows(cond.to(cuda), a.to(cuda), b[0].to(cuda)), ref.where_rows(cond, a, b[0]), rtol=0, atol=0)
    bank = torch.full((6, 5, 50, 32), 3.0, dtype=dtype, device=cuda)
    dst = bank[:, 2]                                       # one strided row block per cond entry (the object pointers' rows of the memory bank)
    ops.where_rows(cond.to(cuda), a.to(cuda), b[0].to(cuda), out=dst)
    close(dst, ref.where_rows(cond, a, b[0]), rtol=0, atol=0)
    assert float((bank[:, :2] - 3.0).abs().max()) == 0.0 and float((bank[:, 3:] - 3.0).abs().max()) == 0.0
    x = rnd(3, 40, 40, seed=5) * 4
    for binz in (0, 1):
        close(ops.mask_for_mem(x.to(cuda), binz, 20.0, -10.0, dtype), ref.mask_for_mem(x, binz, 20.0, -10.0, dtype), **tol(dtype))
    assert torch.equal(ops.threshold(x.to(cuda)).cpu(), ref.threshold(x))


@pytest.mark.parametrize("dtype", DT)
def test_rope_embed_argmax(cuda, dtype):
    from videoglamm_amd import ops
    S, H, D = 37, 6, 64
    big = rnd(S, H + 2, D, dtype=dtype, seed=1)
    ang = torch.arange(100)[:, None].float() * (1.0 / (10000 ** (torch.arange(0, D, 2).float() / D)))[None]
    cos, sin = ang.cos(), ang.sin()
    g = big.to(cuda)
    ops.rope_half_(g[:, :H], cos.to(cuda), sin.to(cuda), 11)
    r = big.clone()
    ref.rope_half_(r[:, :H], cos, sin, 11)
    close(g, r, **tol(dtype))
    B, N, C, n_grid = 2, 2 * 64 + 8, 32, 64
    x = rnd(B, N, C, dtype=dtype, seed=2)
    c2, s2 = rnd(n_grid, C // 2, seed=3).cos(), rnd(n_grid, C // 2, seed=3).sin()
    g = x.to(cuda)
    ops.rope_axial_(g, c2.to(cuda), s2.to(cuda), 2 * 64, n_grid)
    close(g, ref.rope_axial_(x.clone(), c2, s2, 2 * 64, n_grid), **tol(dtype))
    table = rnd(50, 24, dtype=dtype, seed=4)
    ids = torch.tensor([3, 49, 0, 3, 7])
    close(ops.embed(ids.to(cuda), table.to(cuda)), ref.embed(ids, table), rtol=0, atol=0)
    lg = rnd(3, 1000, dtype=dtype, seed=5)
    lg[1, 17] = lg[1, 900] = 50.0   # tie -> lowest index
    assert torch.equal(ops.argmax(lg.to(cuda)).cpu(), torch.tensor([int(lg[0].float().argmax()), 17, int(lg[2].float().argmax())]))


@pytest.mark.parametrize("dtype", DT)
def test_spatial(cuda, dtype):
    from videoglamm_amd import ops
    x = rnd(2, 12, 20, 8, dtype=dtype, seed=1)
    g = x.to(cuda)
    for (kh, st, pad) in ((7, 4, 3), (3, 2, 1), (2, 2, 0), (1, 1, 0)):
        kp = -(-kh * kh * 8 // 8) * 8 + 8
        c, Ho, Wo = ops.im2col(g, kh, kh, st, pad, kp)
        cr, Hr, Wr = ref.im2col(x, kh, kh, st, pad, kp)
        assert (Ho, Wo) == (Hr, Wr)
        close(c, cr, rtol=0, atol=0)
    w, b = rnd(49, 8, seed=2), rnd(8, seed=3)
    close(ops.dwconv(g, w.to(cuda), b.to(cuda), 7), ref.dwconv(x, w, b, 7), **tol(dtype, 49))
    gg = rnd(2 * 12 * 20, 4 * 8, dtype=dtype, seed=4)
    close(ops.pixel_shuffle2(gg.to(cuda), b.to(cuda), 2, 12, 20, 8), ref.pixel_shuffle2(gg, b, 2, 12, 20, 8), **tol(dtype))
    close(ops.pool2(g, True), ref.pool2(x, True), rtol=0, atol=0)
    close(ops.pool2(g, False), ref.pool2(x, False), **tol(dtype))
    fused = rnd(2, 12, 20, 24, dtype=dtype, seed=5)
    close(ops.pool2(fused.to(cuda)[..., 8:16], True), ref.pool2(fused[..., 8:16], True), rtol=0, atol=0)
    big = rnd(3, 16, 24, 3 * 144, dtype=dtype, seed=6)         # Hiera's q slice of a fused projection: 8 channels per thread in bf16
    close(ops.pool2(big.to(cuda)[..., :144], True), ref.pool2(big[..., :144], True), rtol=0, atol=0)
    close(ops.pool2(big.to(cuda)[..., :144], False), ref.pool2(big[..., :144], False), **tol(dtype))
    close(ops.pool2(big.to(cuda)[..., 4:148], True), ref.pool2(big[..., 4:148], True), rtol=0, atol=0)      # misaligned view: scalar kernel
    for ws in (4, 5, 7):
        wn = ops.window_partition(g, ws)
        close(wn, ref.window_partition(x, ws), rtol=0, atol=0)
        close(ops.window_unpartition(wn, ws, 2, 12, 20), x, rtol=0, atol=0)
    close(ops.upsample2_add(rnd(2, 24, 40, 8, dtype=dtype, seed=6).to(cuda), g), ref.upsample2_add(rnd(2, 24, 40, 8, dtype=dtype, seed=6), x), **tol(dtype))
    p = ops.permute5(g, (2, 20, 12, 2, 4), (12 * 20 * 8, 8, 20 * 8, 4, 1))
    close(p, ref.permute5(x, (2, 20, 12, 2, 4), (12 * 20 * 8, 8, 20 * 8, 4, 1)), rtol=0, atol=0)


def test_bilinear(cuda):
    from videoglamm_amd import ops
    x = rnd(3, 64, 64, seed=1)
    for (Ho, Wo) in ((256, 256), (100, 37), (32, 32), (64, 64), (224, 400)):
        close(ops.bilinear(x.to(cuda), Ho, Wo), ref.bilinear(x, Ho, Wo), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("N,Hi,Wi,Ho,Wo", [(3, 64, 64, 40, 56), (2, 256, 256, 1024, 1024), (5, 16, 16, 33, 31), (1, 64, 64, 480, 854), (4, 256, 256, 512, 512)])
def test_bilinear_mask(cuda, N, Hi, Wi, Ho, Wo):
    """the fused upsample + threshold == vg_threshold(vg_bilinear(x)) bit for bit (odd widths take the byte-store path)"""
    from videoglamm_amd import ops
    x = rnd(N, Hi, Wi, seed=7).to(cuda)
    x[0, :3] = 0.0                      # exact zeros: "> 0" is strict
    got = ops.bilinear_mask(x, Ho, Wo)
    want = ops.threshold(ops.bilinear(x, Ho, Wo))
    assert got.dtype == torch.uint8 and torch.equal(got, want)
    cpu = ref.threshold(ref.bilinear(x.cpu(), Ho, Wo))          # independent statement: only logits within rounding of 0 may differ
    assert (got.cpu() != cpu).float().mean() < 1e-4


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,W,ws,K,N", [(2, 16, 16, 8, 144, 432), (1, 20, 12, 7, 72, 40), (3, 64, 64, 14, 64, 192), (2, 8, 8, 4, 288, 288),
                                         (16, 64, 64, 16, 576, 1728)])      # the last: Hiera stage 3's windowed qkv -> the 256x256 phase-split kernel's gather (r04)
def test_gemm_window(cuda, dtype, B, H, W, ws, K, N):
    """window_partition folded into the A-row gather and window_unpartition + residual into the epilogue scatter,
    incl. shapes that need the reference's zero padding (20x12 / 7, 64x64 / 14)."""
    from videoglamm_amd import ops
    x = rnd(B, H, W, K, dtype=dtype, seed=1)
    w, bias = rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    y = ops.linear_window(x.to(cuda), w.to(cuda), bias.to(cuda), B, H, W, ws, scatter=False)
    r = ref.linear_window(x, w, bias, B, H, W, ws, scatter=False)
    close(y, r, **tol(dtype, K))
    # same numbers as the two-kernel path, bit for bit
    assert torch.equal(y, ops.linear(ops.window_partition(x.to(cuda), ws), w.to(cuda), bias.to(cuda)))
    w2, res = rnd(K, N, dtype=dtype, seed=4, scale=N ** -0.5), rnd(B, H, W, K, dtype=dtype, seed=5)
    z = ops.linear_window(y, w2.to(cuda), None, B, H, W, ws, scatter=True, residual=res.to(cuda))
    close(z, ref.linear_window(r.to(dtype), w2, None, B, H, W, ws, scatter=True, residual=res), **tol(dtype, N))


@pytest.mark.parametrize("M,N,K,act,res", [(65536 + 77, 432, 144, 0, False), (65536, 144, 144, 0, True), (66000, 288, 144, 0, False), (65600, 864, 144, 0, False),
                                           (65536 + 300, 864, 288, 0, False), (65536, 288, 288, 0, True), (65537, 1152, 288, 1, False), (70000, 256, 288, 0, False),
                                           (131072, 16, 144, 1, False), (65536 + 255, 48, 288, 0, True)])
def test_gemm_rr(cuda, M, N, K, act, res):
    """the row-register kernel (vg_gemm_rr.hip: K = 144 / 288 over >= 65536 rows — Hiera stages 1-2, FPN laterals): W resident (K = 144, N <= 512) and streamed,
    ragged last panel, column blocks that end inside a 64-column chunk (N = 432, 144, 48, 16), GELU, residual; against fp32 matmul of the same bf16 operands,
    and the strided-input form (lda > K)."""
    from videoglamm_amd import ops
    from videoglamm_amd import _lib
    assert _lib.load().vg_gemm_route(M, N, K, ops.BF16, 0, 0) == 7, "the shape must take the row-register route"
    g = torch.Generator(device=cuda).manual_seed(M + N)
    x = (torch.randn(M, K + 16, device=cuda, generator=g) * 0.5).to(torch.bfloat16)[:, :K]          # rows with a stride: lda = K + 16
    w = (torch.randn(N, K, device=cuda, generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=cuda, generator=g)
    r = (torch.randn(M, N, device=cuda, generator=g)).to(torch.bfloat16) if res else None
    y = ops.linear(x, w, b, act=act, residual=r)
    z = x.float() @ w.float().t() + b
    if act == 1:
        z = torch.nn.functional.gelu(z)
    if res:
        z = z + r.float()
    err = (y.float() - z).abs()
    assert err.max().item() < 2e-2 + 8e-3 * z.abs().max().item(), err.max().item()
    assert (err > 1e-2 + 4e-3 * z.abs()).float().mean().item() < 1e-5
    # contiguous input: same numbers
    assert torch.equal(y, ops.linear(x.contiguous(), w, b, act=act, residual=r))


@pytest.mark.parametrize("M,N,K,act", [(65536 + 41, 432, 144, 0), (65536, 864, 288, 0), (70000, 1152, 288, 1), (65600, 288, 144, 1)])
def test_gemm_ln(cuda, M, N, K, act):
    """LayerNorm -> projection in one launch (vg_gemm_ln on the row-register kernel) against the two launches: the normalised rows are rounded to bf16 as
    vg_layernorm's output is, so the results agree up to the summation order of the row statistics (a bf16 step on a handful of elements at most)."""
    from videoglamm_amd import ops
    g = torch.Generator(device=cuda).manual_seed(M + K)
    x = (torch.randn(M, K, device=cuda, generator=g) * 2.0 + 0.3).to(torch.bfloat16)
    lw, lb = torch.randn(K, device=cuda, generator=g) * 0.2 + 1.0, torch.randn(K, device=cuda, generator=g) * 0.1
    w = (torch.randn(N, K, device=cuda, generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=cuda, generator=g)
    y = ops.linear_ln(x, (lw, lb, 1e-6), w, b, act=act)
    r = ops.linear(ops.layernorm(x, lw, lb, 1e-6), w, b, act=act)
    d = (y.float() - r.float()).abs()
    assert (d > 0).float().mean().item() < 2e-3 and d.max().item() <= 0.07, ((d > 0).float().mean().item(), d.max().item())
    z = torch.nn.functional.layer_norm(x.float(), (K,), lw, lb, 1e-6) @ w.float().t() + b
    if act == 1:
        z = torch.nn.functional.gelu(z)
    assert (y.float() - z).abs().max().item() < 4e-2 + 1e-2 * z.abs().max().item()


@pytest.mark.parametrize("B,H,W,ws,K,N", [(2, 256, 256, 8, 144, 432), (4, 128, 128, 4, 288, 864), (5, 120, 128, 7, 288, 288)])
def test_gemm_ln_window(cuda, B, H, W, ws, K, N):
    """the same behind the window gather: padding rows (7-token windows on a 120 x 128 grid) are zero BEHIND the norm — their outputs are the bias"""
    from videoglamm_amd import ops
    g = torch.Generator(device=cuda).manual_seed(B + ws)
    x = (torch.randn(B, H, W, K, device=cuda, generator=g) * 1.5).to(torch.bfloat16)
    lw, lb = torch.randn(K, device=cuda, generator=g) * 0.2 + 1.0, torch.randn(K, device=cuda, generator=g) * 0.1
    w = (torch.randn(N, K, device=cuda, generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=cuda, generator=g)
    y = ops.linear_ln(x, (lw, lb, 1e-6), w, b, window=(B, H, W, ws))
    r = ops.linear_window(ops.layernorm(x, lw, lb, 1e-6), w, b, B, H, W, ws, scatter=False)
    assert y.shape == r.shape
    d = (y.float() - r.float()).abs()
    assert (d > 0).float().mean().item() < 2e-3 and d.max().item() <= 0.07, ((d > 0).float().mean().item(), d.max().item())


@pytest.mark.parametrize("B,H,W,ws,K,N", [(2, 256, 256, 8, 144, 432), (4, 128, 128, 4, 288, 864), (3, 160, 144, 8, 144, 144), (5, 120, 128, 7, 288, 288)])
def test_gemm_rr_window(cuda, B, H, W, ws, K, N):
    """window gather (A rows) and window scatter + residual (C / R rows) on the row-register kernel, power-of-two windows and a 7-token window that pads the
    grid (padding rows read zeros and are dropped on the way back): equal to the two-kernel path bit for bit, close to the fp32 statement."""
    from videoglamm_amd import ops
    g = torch.Generator(device=cuda).manual_seed(B * H + ws)
    x = (torch.randn(B, H, W, K, device=cuda, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=cuda, generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=cuda, generator=g)
    y = ops.linear_window(x, w, b, B, H, W, ws, scatter=False)
    xp = ops.window_partition(x, ws)
    assert torch.equal(y, ops.linear(xp, w, b))
    z = xp.float() @ w.float().t() + b
    assert (y.float() - z).abs().max().item() < 2e-2 + 8e-3 * z.abs().max().item()
    w2 = (torch.randn(K, N, device=cuda, generator=g) * N ** -0.5).to(torch.bfloat16)
    if N in (144, 288):
        res = torch.randn(B, H, W, K, device=cuda, generator=g).to(torch.bfloat16)
        o = ops.linear_window(y, w2, None, B, H, W, ws, scatter=True, residual=res)
        zz = ops.window_unpartition((y.float() @ w2.float().t()), ws, B, H, W) + res.float() if hasattr(ops, "window_unpartition") else None
        if zz is not None:
            assert (o.float() - zz).abs().max().item() < 3e-2 + 8e-3 * zz.abs().max().item()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Bw,wtok,H,D", [(64, 16, 4, 72), (8, 64, 2, 72), (6, 16, 2, 32), (16, 49, 4, 72), (24, 32, 1, 64),
                                         (5, 256, 8, 72), (3, 256, 2, 64), (2, 256, 3, 80), (33, 256, 8, 72)])
def test_attention_windows(cuda, dtype, Bw, wtok, H, D):
    """many small windows packed into 128-token tiles under the block-diagonal mask == per-window attention; q/k/v are
    strided views of one fused [Bw, wtok, 3, H, D] projection as in Hiera (49 tokens / 6 windows: unpackable -> fallback).
    256-token windows (Hiera stage 3) take vg_window_attention in bf16: one workgroup per (window, head)."""
    from videoglamm_amd import ops
    qkv = rnd(Bw, wtok, 3, H, D, dtype=dtype, seed=7)
    if wtok == 256:      # a key spike late in the window and a large-magnitude window: the single-pass softmax must hold
        qkv[0, 200, 1] *= 20.0
        qkv[-1] *= 4.0
    g = qkv.to(cuda)
    o = ops.attention_windows(g[:, :, 0], g[:, :, 1], g[:, :, 2], D ** -0.5)
    t = dict(rtol=1e-3, atol=2e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=2e-2)
    close(o, ref.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], D ** -0.5), **t)


@pytest.mark.parametrize("Bw,wq,wk,H", [(37, 4, 16, 8), (64, 16, 64, 4), (5, 16, 16, 3), (9, 64, 64, 2), (1024, 16, 16, 4), (300, 64, 64, 16)])
def test_window_attention_small(cuda, Bw, wq, wk, H):
    """one wave per (window, head) on 16x16x32 MFMA tiles (vg_window_attention): the 16- and 64-token windows of Hiera stages 1 / 2 / 4
    and the q-pooled first block of a stage (4 queries x 16 keys, 16 x 64); item counts that do not fill the last workgroup;
    q is its own tensor (pooled), k / v strided views of the fused projection."""
    from videoglamm_amd import ops
    D = 72
    qkv = rnd(Bw, wk, 3, H, D, dtype=torch.bfloat16, seed=9)
    q = rnd(Bw, wq, H, D, dtype=torch.bfloat16, seed=10) if wq != wk else qkv[:, :, 0]
    qkv[0, wk - 3, 1] *= 15.0                      # a spike among the keys
    g = qkv.to(cuda)
    gq = q.to(cuda) if wq != wk else g[:, :, 0]
    o = ops.window_attention(gq, g[:, :, 1], g[:, :, 2], D ** -0.5)
    assert o is not None and o.shape == (Bw, wq, H, D)
    close(o, ref.attention(q, qkv[:, :, 1], qkv[:, :, 2], D ** -0.5), rtol=3e-2, atol=2e-2)
    assert ops.window_attention(gq.float(), g[:, :, 1].float(), g[:, :, 2].float(), D ** -0.5) is None      # fp32: the generic kernel's job


def test_errors_are_loud(cuda):
    from videoglamm_amd import _lib, ops
    with pytest.raises(_lib.VGKernelError):
        ops.linear(torch.zeros(4, 12, device=cuda)[:, :6], torch.zeros(4, 12, device=cuda)[:, :6])  # K=6 not a multiple of 4
    with pytest.raises(_lib.VGKernelError):
        ops.linear(torch.zeros(4, 8), torch.zeros(4, 8))  # CPU tensors: no fallback


@pytest.mark.parametrize("N,K,glu,norm,res", [(4096, 4096, False, False, True), (6144, 4096, False, True, False), (14336, 4096, True, True, False),
                                              (4096, 14336, False, False, True), (3072, 8192, False, False, False), (9216, 3072, False, True, False),
                                              (1001, 4096, False, False, False)])
def test_decode_gemv_w8(cuda, N, K, glu, norm, res):
    """fp8 (OCP e4m3) weights with per-row scales: exact against the fp32 product with the DEQUANTISED weights (the only
    difference to the bf16 kernel is the weight format), and within the quantisation error of the original weights."""
    from videoglamm_amd import ops
    rows = 2 * N if glu else N
    w = rnd(rows, K, seed=1, scale=K ** -0.5)
    x = rnd(1, K, dtype=torch.bfloat16, seed=2)
    nw = (1.0 + 0.1 * rnd(K, seed=3)) if norm else None
    r = rnd(1, N, dtype=torch.bfloat16, seed=4) if res else None
    q, sc = ops.quantize_fp8_rows(w.to(cuda))
    deq = q.view(torch.float8_e4m3fn).float().cpu() * sc.cpu()[:, None]
    assert (deq - w).abs().max() <= 0.0625 * w.abs().amax(dim=1).max() + 1e-6          # e4m3: 3 mantissa bits
    y = ops.decode_gemv_w8(x.to(cuda), q, sc, norm_w=None if nw is None else nw.to(cuda), eps=1e-5, residual=None if r is None else r.to(cuda), glu=glu)
    want = ref.decode_gemv(x, deq.to(torch.bfloat16).float(), norm_w=nw, eps=1e-5, residual=r, glu=glu) if False else None
    xf = x.float()
    if nw is not None:
        xf = ((xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).to(torch.bfloat16).float() * nw).to(torch.bfloat16).float()
    acc = xf @ deq.t()
    if glu:
        g, u = acc[:, :N].to(torch.bfloat16).float(), acc[:, N:].to(torch.bfloat16).float()
        acc = torch.nn.functional.silu(g).to(torch.bfloat16).float() * u
    if r is not None:
        acc = acc + r.float()
    close(y, acc.to(torch.bfloat16), rtol=2e-2, atol=2e-2)
    yf = ops.decode_gemv_w8(x.to(cuda), q, sc, norm_w=None if nw is None else nw.to(cuda), eps=1e-5, glu=glu, out_dtype=torch.float32)
    if not glu and r is None:
        close(yf, acc, rtol=2e-3, atol=2e-3)
    from videoglamm_amd import _lib
    with pytest.raises(_lib.VGKernelError):
        ops.decode_gemv_w8(x.to(cuda)[:, :1024].contiguous(), q[:, :1024].contiguous(), sc)


@pytest.mark.parametrize("M,N,K,glu,res", [(200, 512, 256, False, False), (1697, 6144, 4096, False, False), (333, 4096, 4096, False, True),
                                           (257, 1024, 4096, True, False), (640, 4096, 14336, False, True), (130, 264, 144, False, False)])
def test_gemm_f8(cuda, M, N, K, glu, res):
    """fp8 x fp8 GEMM with per-row scales on both operands: the device quantiser must give torch's e4m3 codes, and the
    product must match the fp32 product of the DEQUANTISED operands (the MFMA sums exact products in fp32)."""
    from videoglamm_amd import ops
    x = rnd(M, K, dtype=torch.bfloat16, seed=1)
    w = rnd((2 * N if glu else N), K, seed=2, scale=K ** -0.5)
    r = rnd(M, N, dtype=torch.bfloat16, seed=3) if res else None
    q, qs = ops.quantize_fp8(x.to(cuda))
    sc_ref = (x.float().abs().amax(dim=1).clamp_min(1e-12) / 448.0)
    torch.testing.assert_close(qs.cpu(), sc_ref, rtol=1e-6, atol=0)
    q_ref = (x.float() / qs.cpu()[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(q.cpu(), q_ref)
    w8, ws = ops.quantize_fp8_rows(w.to(cuda))
    y = ops.linear_f8(q, qs, w8, ws, residual=None if r is None else r.to(cuda), glu=glu)
    xd = q.cpu().view(torch.float8_e4m3fn).float() * qs.cpu()[:, None]
    wd = w8.cpu().view(torch.float8_e4m3fn).float() * ws.cpu()[:, None]
    acc = xd @ wd.t()
    if glu:
        g, u = acc[:, :N].to(torch.bfloat16).float(), acc[:, N:].to(torch.bfloat16).float()
        acc = torch.nn.functional.silu(g).to(torch.bfloat16).float() * u
    if r is not None:
        acc = acc + r.float()
    close(y, acc.to(torch.bfloat16), rtol=2e-2, atol=2e-2)
    if not glu and r is None:
        close(ops.linear_f8(q, qs, w8, ws, out_dtype=torch.float32), acc, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(213, 4096, 14336), (213, 4096, 4096), (2050, 1408, 6144), (100, 264, 4160), (425, 1024, 8192)])
def test_gemm_splitk(cuda, dtype, M, N, K, monkeypatch):
    """few tiles, long K: ops.linear cuts K into slices (vg_gemm_splitk) — same result as the single-pass GEMM up to fp32
    summation order, with bias / activation / LayerScale / residual applied by the reducing pass."""
    from videoglamm_amd import ops
    monkeypatch.setenv("VG_GEMM_SPLITK", "1")
    monkeypatch.delenv("VG_GEMM_SPLITK_TILES", raising=False)
    assert ops._splitk(M, N, K, 2 if dtype == torch.bfloat16 else 4) >= 2      # every case here is routed to split-K
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    bias, gamma, res = rnd(N, seed=3), 1.0 + 0.1 * rnd(N, seed=4), rnd(M, N, dtype=dtype, seed=5)
    y = ops.linear(x.to(cuda), w.to(cuda), bias.to(cuda), act=ops.ACT_GELU, gamma=gamma.to(cuda), residual=res.to(cuda))
    close(y, ref.linear(x, w, bias, act=ref.ACT_GELU, gamma=gamma, residual=res), **tol(dtype, K))
    close(ops.linear(x.to(cuda), w.to(cuda)), ref.linear(x, w), **tol(dtype, K))


BIG = [  # (M, N, K, glu): bf16 shapes the launcher routes to the 256x256-tile kernels (vg_gemm_route == 3) — the LLM prefill GEMMs of C1 / C2
    (3361, 4096, 14336, False),      # C2 down projection (+ residual)
    (1697, 14336, 4096, True),       # C1 gate|up with the SwiGLU epilogue
    (4096, 4096, 4096, False),
    (3333, 4104, 4160, False),       # ragged M and N, K = 65 steps
    (3361, 4096, 4096, False),       # C2 o projection
]


def _check_big_gemm(cuda, M, N, K, glu, routed=True):
    from videoglamm_amd import _lib, ops
    assert (_lib.load().vg_gemm_route(M, N, K, 1, 1 if glu else 0, 0) == 3) == routed, "this shape must take the 256x256-tile kernel"
    dtype = torch.bfloat16
    x = rnd(M, K, dtype=dtype, seed=1)
    w = rnd((2 if glu else 1) * N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    bias = rnd((2 if glu else 1) * N, seed=3)
    if glu:
        y = ops.linear(x.to(cuda), w.to(cuda), bias.to(cuda), glu=True)
        close(y, ref.linear(x, w, bias, glu=True), rtol=2e-2, atol=2e-2)
        assert torch.equal(y, ops.swiglu(ops.linear(x.to(cuda), w.to(cuda), bias.to(cuda))))     # same roundings as GEMM + vg_swiglu
        return
    gamma, res = 1.0 + 0.1 * rnd(N, seed=4), rnd(M, N, dtype=dtype, seed=5)
    y = ops.linear(x.to(cuda), w.to(cuda), bias.to(cuda), ops.ACT_GELU, gamma.to(cuda), res.to(cuda))
    close(y, ref.linear(x, w, bias, ref.ACT_GELU, gamma, res), rtol=2e-2, atol=2e-2)
    y = ops.linear(x.to(cuda), w.to(cuda), residual=res.to(cuda))
    close(y, ref.linear(x, w, residual=res), rtol=2e-2, atol=2e-2)
    y32 = ops.linear(x.to(cuda), w.to(cuda), out_dtype=torch.float32)                          # fp32 output: no output rounding in the way
    close(y32, ref.linear(x, w, out_dtype=torch.float32), rtol=2e-3, atol=2e-3)


NARROW = [  # (M, N, K): bf16 shapes launch_gemm routes to the 256x192-tile kernel (vg_gemm_route == 6)
    (65536, 576, 2304),      # Hiera stage 3 fc2 (three exact column tiles; 768 tiles = 3 rounds)
    (16384, 576, 576),       # ... its proj at a quarter of the rows (K = 9 steps: prologue / tail kinds back to back)
    (3361, 6144, 4096),      # Llama q|k|v at the C2 prompt length: ragged M (13 x 256 + 33)
    (9232, 1024, 4096),      # CLIP fc2: ragged N (5 x 192 + 64) and ragged M
    (16384, 1160, 1152),     # N = 6 x 192 + 8: a last column tile with one 8-column group
]


@pytest.mark.parametrize("M,N,K", NARROW)
def test_gemm_p8n(cuda, M, N, K):
    """the 256x192-tile phase-split kernel against the fp32 statement: every epilogue family (bias + activation, LayerScale + residual, residual
    alone, plain, fp32 output) on exact, ragged-M and ragged-N grids, long and minimal K."""
    from videoglamm_amd import _lib, ops
    assert _lib.load().vg_gemm_route(M, N, K, 1, 0, 0) == 6, "this shape must take the 256x192-tile kernel"
    dtype = torch.bfloat16
    x = rnd(M, K, dtype=dtype, seed=1)
    w = rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    bias, gamma, res = rnd(N, seed=3), 1.0 + 0.1 * rnd(N, seed=4), rnd(M, N, dtype=dtype, seed=5)
    xd, wd = x.to(cuda), w.to(cuda)
    close(ops.linear(xd, wd, bias.to(cuda), ops.ACT_GELU), ref.linear(x, w, bias, ref.ACT_GELU), rtol=2e-2, atol=2e-2)
    close(ops.linear(xd, wd, bias.to(cuda), ops.ACT_NONE, gamma.to(cuda), res.to(cuda)), ref.linear(x, w, bias, ref.ACT_NONE, gamma, res), rtol=2e-2, atol=2e-2)
    close(ops.linear(xd, wd, residual=res.to(cuda)), ref.linear(x, w, residual=res), rtol=2e-2, atol=2e-2)
    close(ops.linear(xd, wd), ref.linear(x, w), rtol=2e-2, atol=2e-2)
    close(ops.linear(xd, wd, bias.to(cuda), ops.ACT_SILU, None, res.to(cuda)), ref.linear(x, w, bias, ref.ACT_SILU, None, res), rtol=2e-2, atol=2e-2)   # run-time activation path
    close(ops.linear(xd, wd, bias.to(cuda), out_dtype=torch.float32), ref.linear(x, w, bias, out_dtype=torch.float32), rtol=2e-3, atol=2e-3)
    # a transposed / shifted tile would survive random data only by luck: one-hot rows pick single W columns
    eye = torch.zeros(M, K, dtype=dtype)
    idx = torch.arange(M) % K
    eye[torch.arange(M), idx] = 1.0
    y = ops.linear(eye.to(cuda), wd).float().cpu()
    assert torch.equal(y, w.float().t()[idx])


def _check_p8n(cuda, M, N, K):
    from videoglamm_amd import ops
    dtype = torch.bfloat16
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    bias, res = rnd(N, seed=3), rnd(M, N, dtype=dtype, seed=5)
    close(ops.linear(x.to(cuda), w.to(cuda), bias.to(cuda), ops.ACT_GELU), ref.linear(x, w, bias, ref.ACT_GELU), rtol=2e-2, atol=2e-2)
    close(ops.linear(x.to(cuda), w.to(cuda), residual=res.to(cuda)), ref.linear(x, w, residual=res), rtol=2e-2, atol=2e-2)


def test_gemm_p8n_minimal_k_forced(cuda):
    """K = 2 / 3 / 4 / 5 steps (the pipeline's PRELAST / LAST tails directly behind the prologue) and grids of a few tiles, which the shape rule never
    sends to the 256x192 kernel: VG_GEMM_P8=3 forces it (read once per process: a child process runs it)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, torch; sys.path[:0] = [%r, %r]\n"
            "import test_kernels_gpu as t\n"
            "from videoglamm_amd import _lib\n"
            "lib = _lib.load(); assert lib.vg_init(0) > 0\n"
            "dev = torch.device('cuda:0')\n"
            "for s in ((4111, 1160, 128), (4111, 1160, 192), (8192, 1152, 256), (777, 384, 320), (300, 192, 4096)):\n"
            "    assert lib.vg_gemm_route(*s, 1, 0, 0) == 6\n"
            "    t._check_p8n(dev, *s)\n"
            "print('forced ok')\n") % (os.path.dirname(here), here)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VG_GEMM_P8="3"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "forced ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_gemm_p8n_window_forms(cuda):
    """Hiera stage 3's window-folded proj (N = 576: the 256x192 route) — rows scattered to image order + residual — and a gathered A on the same
    kernel, against the unfolded statement."""
    from videoglamm_amd import ops
    B, H, W, ws, C, N = 16, 64, 64, 16, 576, 576
    dtype = torch.bfloat16
    x = rnd(B, H, W, C, dtype=dtype, seed=1)
    w, bias = rnd(N, C, dtype=dtype, seed=2, scale=C ** -0.5), rnd(N, seed=3)
    res = rnd(B, H, W, N, dtype=dtype, seed=4)
    y = ops.linear_window(x.to(cuda), w.to(cuda), bias.to(cuda), B, H, W, ws, scatter=False)
    close(y, ref.linear_window(x, w, bias, B, H, W, ws, scatter=False), rtol=2e-2, atol=2e-2)
    xw = ref.window_partition(x, ws)
    z = ops.linear_window(xw.to(cuda), w.to(cuda), bias.to(cuda), B, H, W, ws, scatter=True, residual=res.to(cuda))
    close(z, ref.linear_window(xw, w, bias, B, H, W, ws, scatter=True, residual=res), rtol=2e-2, atol=2e-2)


def test_gemm_tail_rows_split(cuda):
    """M = 13 x 256 + 33 rows on the 256x256-tile route: when dropping the last row of tiles saves a round of 256 workgroups, ops.linear sends
    the 33 tail rows through their own launch (GLU epilogue, residual and plain) — same values as the one-launch result up to bf16 rounding."""
    from videoglamm_amd import _lib, ops
    M, N, K = 3361, 14336, 4096
    assert _lib.load().vg_gemm_route(M, N, K, 1, 1, 0) == 3
    x = rnd(M, K, dtype=torch.bfloat16, seed=1)
    w = rnd(2 * N, K, dtype=torch.bfloat16, seed=2, scale=K ** -0.5)
    y = ops.linear(x.to(cuda), w.to(cuda), glu=True)
    close(y[-40:], ref.linear(x[-40:], w, glu=True), rtol=2e-2, atol=2e-2)          # the tail launch ...
    close(y[:300], ref.linear(x[:300], w, glu=True), rtol=2e-2, atol=2e-2)          # ... and the head
    head = ops.linear(x[:3328].to(cuda), w.to(cuda), glu=True)                       # whole tiles only: the head rows are bit-identical
    assert torch.equal(y[:3328], head)


@pytest.mark.parametrize("M,N,K,glu", BIG)
def test_gemm_p8(cuda, M, N, K, glu):
    """the default (eight-wave) 256x256-tile kernel on the bench's heaviest bf16 GEMMs, against the fp32 statement."""
    _check_big_gemm(cuda, M, N, K, glu)


def test_gemm_p8_persistent_queue(cuda):
    """the persistent 256x256 kernel's tile queue: more tiles than CUs over a BATCH of problems (tiles of all batch entries form one
    queue; every workgroup walks several tiles and prefetches the next tile's first K step under the last), K = 1152 (18 K steps: the
    short-K shapes r02 routes here), against the per-entry GEMMs that take other kernels."""
    from videoglamm_amd import _lib, ops
    B, M, N, K = 24, 1024, 1024, 1152
    a = (torch.randn(B, M, K, device=cuda, dtype=torch.float32) * 0.5).to(torch.bfloat16)
    w = (torch.randn(B, N, K, device=cuda, dtype=torch.float32) * 0.5).to(torch.bfloat16)
    y = ops.bmm_nt(a, w)
    for b in (0, 7, B - 1):
        assert _lib.load().vg_gemm_route(M, N, K, 1, 0, 0) != 3          # one entry alone does not fill the chip: another kernel
        close(y[b], ops.linear(a[b], w[b]).float().cpu(), rtol=2e-2, atol=0.3)
        close(y[b], a[b].float().cpu() @ w[b].float().cpu().t(), rtol=2e-2, atol=0.3)


def test_gemm_route_knobs(cuda):
    """VG_GEMM_P8=0 (the 256x256 route switched off: the same shapes on the 128x128 kernels) and VG_ATTN_XCD=0 (attention workgroups in plain
    dispatch order) are the A/B knobs the launchers still read; they are read once per process, so a child process runs them."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, torch; sys.path[:0] = [%r, %r]\n"
            "import test_kernels_gpu as t\n"
            "from videoglamm_amd import _lib\n"
            "lib = _lib.load(); assert lib.vg_init(0) > 0\n"
            "dev = torch.device('cuda:0')\n"
            "assert all(lib.vg_gemm_route(M, N, K, 1, 1 if g else 0, 0) != 3 for M, N, K, g in t.BIG)\n"
            "for c in t.BIG[:3]: t._check_big_gemm(dev, *c, routed=False)\n"
            "for c in t.ATT: t.test_attention(dev, torch.bfloat16, c)\n"
            "print('knobs ok')\n") % (os.path.dirname(here), here)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VG_GEMM_P8="0", VG_ATTN_XCD="0"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "knobs ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


WIN = [  # (Hq, Hkv, Sq, Skv, D, window): causal + sliding window (the query's own position and the window - 1 before it)
    (4, 4, 700, 700, 96, 100),       # Phi-3 head dim, window shorter than a query tile
    (4, 4, 700, 700, 96, 257),       # window spanning several key tiles, ragged edge
    (8, 2, 333, 333, 128, 64),       # GQA
    (4, 4, 213, 1697, 96, 300),      # chunked prefill rows (Sq < Skv): split-KV + merge with whole splits outside the window
    (4, 4, 1, 700, 96, 130),         # one row against a cache
    (32, 32, 3361, 3361, 96, 2048),  # the released model at NUM_FRAMES = 16: Phi-3-mini heads, S = 3361, 2048 visible keys
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cfg", WIN)
def test_attention_sliding_window(cuda, dtype, cfg):
    from videoglamm_amd import ops
    Hq, Hkv, Sq, Skv, D, window = cfg
    if Sq > 3000 and dtype == torch.float32:
        pytest.skip("full-size case runs in bf16 only")
    q, k, v = rnd(1, Sq, Hq, D, dtype=dtype, seed=1), rnd(1, Skv, Hkv, D, dtype=dtype, seed=2), rnd(1, Skv, Hkv, D, dtype=dtype, seed=3)
    o = ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), D ** -0.5, True, window=window)
    t = dict(rtol=1e-3, atol=2e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=2e-2)
    want = ref.attention(q, k, v, D ** -0.5, True, window=window)
    close(o, want, **t)
    assert (want.float() - ref.attention(q, k, v, D ** -0.5, True).float()).abs().max() > 1e-2     # the window matters in this case


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("H,Hkv,D,max_len,window", [(4, 4, 96, 1024, 100), (4, 4, 96, 1024, 64), (4, 4, 96, 1024, 65), (32, 8, 128, 2048, 700),
                                                    (32, 32, 96, 4096, 2048)])
def test_decode_attention_sliding_window(cuda, dtype, H, Hkv, D, max_len, window):
    """the fused decode attention with a window: whole splits before the window never run, the first visible split is masked
    per key, and the merge / arrival counter only count the visible splits."""
    from videoglamm_amd import ops
    kc, vc = rnd(max_len, Hkv, D, dtype=dtype, seed=2), rnd(max_len, Hkv, D, dtype=dtype, seed=3)
    ang = torch.arange(max_len)[:, None].float() * (1.0 / (10000 ** (torch.arange(0, D, 2).float() / D)))[None]
    cos, sin = ang.cos(), ang.sin()
    ws = ops.decode_attention_workspace(H, Hkv, D, max_len, cuda)
    t = dict(rtol=1e-3, atol=2e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=2e-2)
    for pos in (0, 5, window - 1, window, window + 1, window + 63, window + 64, max_len // 2 + 7, max_len - 1):
        if pos >= max_len:
            continue
        qkv = rnd(1, (H + 2 * Hkv) * D, dtype=dtype, seed=10 + pos)
        pos_dev = torch.tensor([pos], dtype=torch.int32)
        g_kc, g_vc = kc.to(cuda), vc.to(cuda)
        o = ops.decode_attention(qkv.to(cuda), g_kc, g_vc, cos.to(cuda), sin.to(cuda), H, Hkv, D, pos_dev.to(cuda), D ** -0.5, ws, window=window)
        r_qkv, r_kc, r_vc = qkv.clone(), kc.clone(), vc.clone()
        ref.rope_kv_append_(r_qkv, r_kc, r_vc, cos, sin, H, Hkv, D, 0, pos_dev)
        close(o, ref.attention_decode(r_qkv[:, : H * D].view(1, 1, H, D), r_kc, r_vc, pos_dev, D ** -0.5, window).view(1, H * D), **t)
        # and the unfused path (vg_attention_splitkv with the device-side length)
        u_qkv = qkv.to(cuda)
        ops.rope_kv_append_(u_qkv, kc.to(cuda), vc.to(cuda), cos.to(cuda), sin.to(cuda), H, Hkv, D, 0, pos_dev.to(cuda))
        o2 = ops.attention_decode(u_qkv[:, : H * D].view(1, 1, H, D), g_kc, g_vc, pos_dev.to(cuda), D ** -0.5, window=window)
        close(o2.view(1, H * D), o, **t)
        # the window under 128 keys per workgroup (first visible block, partially visible blocks at both granularities)
        o3 = ops.decode_attention(qkv.to(cuda), kc.to(cuda), vc.to(cuda), cos.to(cuda), sin.to(cuda), H, Hkv, D, pos_dev.to(cuda), D ** -0.5, ws,
                                  window=window, keys_per_wg=128)
        close(o3, o, **(dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=1.6e-2, atol=2e-3)))
    assert int(ws[-Hkv:].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("N,P,nt,TP", [(3, 256, 7, 8), (2, 4096, 7, 8), (2, 4096, 9, 16), (1, 96, 1, 8), (2, 1024, 16, 16)])
def test_twoway_image_update(cuda, N, P, nt, TP):
    """vg_twoway_image_update (SAM2 mask decoder, image -> token pass fused): scores GEMM + per-head softmax over the tokens + output GEMM +
    residual + LayerNorm + dense PE against the fp32 statement on the same bf16 inputs.  The probabilities are rounded to bf16 between the
    two GEMMs (as the attention kernels round P), LayerNorm outputs are O(1): bf16-sized tolerance."""
    from videoglamm_amd import ops
    dt = torch.bfloat16
    NC = 8 * TP
    x, pe = rnd(N, P, 256, dtype=dt, seed=1), rnd(P, 256, dtype=dt, seed=2)
    xpe = (x.float() + pe.float()).to(dt)
    u2 = rnd(N, NC, 256, dtype=dt, seed=3, scale=0.05)
    c2 = rnd(N, NC, seed=4)
    w2t = rnd(N, 256, NC, dtype=dt, seed=5, scale=0.5)
    bo, lw, lb = rnd(256, seed=6), 1.0 + 0.1 * rnd(256, seed=7), rnd(256, seed=8)
    xo, xpo = ops.twoway_image_update(xpe.to(cuda), x.to(cuda), u2.to(cuda), c2.to(cuda), w2t.to(cuda), bo.to(cuda), lw.to(cuda), lb.to(cuda),
                                      1e-5, pe.to(cuda), nt, TP)
    ro, rpo = ref.twoway_image_update(xpe, x, u2, c2, w2t, bo, lw, lb, 1e-5, pe, nt, TP)
    close(xo, ro, rtol=3e-2, atol=3e-2)
    close(xpo, rpo, rtol=3e-2, atol=4e-2)
    # the masked token columns (t >= nt) carry no weight: garbage in their u2 / w2t entries must not change anything
    u2b, w2b = u2.clone().view(N, 8, TP, 256), w2t.clone().view(N, 256, 8, TP)
    u2b[:, :, nt:] = 1e3
    w2b[..., nt:] = 1e3
    xo2, _ = ops.twoway_image_update(xpe.to(cuda), x.to(cuda), u2b.view(N, NC, 256).to(cuda), c2.to(cuda), w2b.view(N, 256, NC).contiguous().to(cuda),
                                     bo.to(cuda), lw.to(cuda), lb.to(cuda), 1e-5, pe.to(cuda), nt, TP)
    assert torch.equal(xo2, xo)


@pytest.mark.parametrize("N,Bi,es", [(2, 2, 16), (4, 2, 16), (2, 1, 64), (3, 3, 8)])
def test_mask_upscale(cuda, N, Bi, es):
    """vg_mask_upscale (mask decoder: ConvT + s1 -> LayerNorm2d -> GELU -> ConvT + s0 -> GELU -> hypernetwork product, one kernel) against the fp32
    statement of the unfused chain on the same bf16 inputs; instance n uses the high-resolution features of image n % Bi."""
    from videoglamm_amd import ops
    dt = torch.bfloat16
    P = es * es
    x = rnd(N, P, 256, dtype=dt, seed=1)
    w0, b0 = rnd(256, 256, dtype=dt, seed=2, scale=0.06), rnd(64, seed=3, scale=0.1)
    w1, b1 = rnd(128, 64, dtype=dt, seed=4, scale=0.12), rnd(32, seed=5, scale=0.1)
    s1, s0 = rnd(Bi, 4 * P, 64, dtype=dt, seed=6), rnd(Bi, 16 * P, 32, dtype=dt, seed=7)
    lw, lb = 1.0 + 0.1 * rnd(64, seed=8), 0.1 * rnd(64, seed=9)
    hyper = rnd(N, 4, 32, dtype=dt, seed=10, scale=0.3)
    y = ops.mask_upscale(x.to(cuda), w0.to(cuda), b0.to(cuda), s1.to(cuda), lw.to(cuda), lb.to(cuda), 1e-6, w1.to(cuda), b1.to(cuda), s0.to(cuda),
                         hyper.to(cuda), es)
    r = ref.mask_upscale(x, w0, b0, s1, lw, lb, 1e-6, w1, b1, s0, hyper, es)
    close(y, r, rtol=3e-2, atol=3e-2)
