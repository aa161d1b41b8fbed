"""vg_mlp3_grouped (-m gpu): G three-layer MLP heads in one launch against the plain statement (tests/_cpu_ops.py: fp32 products of the bf16 operands, bf16
between the layers) and against the one-vg_gemm-per-layer route it replaces in SAM2's mask decoder."""
import pytest
import torch

import _cpu_ops as ref

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def rnd(*shape, seed=0, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


@pytest.mark.parametrize("G,R,K,Hd,No,out_dtype,sig", [
    (4, 1, 256, 256, 32, torch.bfloat16, 0),      # output_hypernetworks_mlps, one object
    (4, 8, 256, 256, 32, torch.bfloat16, 0),      # eight objects
    (4, 70, 256, 256, 32, torch.bfloat16, 0),     # framewise branch: (object, frame) pairs — three row chunks, the last ragged
    (1, 3, 256, 256, 4, torch.float32, 1),        # iou_prediction_head (sigmoid)
    (1, 33, 256, 256, 1, torch.float32, 0),       # pred_obj_score_head
    (1, 5, 256, 256, 256, torch.bfloat16, 0),     # obj_ptr_proj
    (3, 2, 64, 48, 17, torch.float32, 5),         # narrow shapes, an output width that is no multiple of anything, sigmoid on heads 0 and 2
])
def test_mlp3_grouped_vs_statement(cuda, G, R, K, Hd, No, out_dtype, sig):
    from videoglamm_amd import ops
    nt = G + 2
    x = rnd(R, nt, K, seed=1)                                     # the heads read token rows 1 .. G of a wider token tensor (a strided view)
    w0, w1, w2 = rnd(G, Hd, K, seed=2, scale=K ** -0.5), rnd(G, Hd, Hd, seed=3, scale=Hd ** -0.5), rnd(G, No, Hd, seed=4, scale=Hd ** -0.5)
    b0, b1, b2 = (rnd(G, n, seed=5 + i, scale=0.3, dtype=torch.float32) for i, n in enumerate((Hd, Hd, No)))
    want = ref.mlp3_grouped(x[:, 1:], G, ref.mlp3_pack(w0), b0, ref.mlp3_pack(w1), b1, ref.mlp3_pack(w2), b2, torch.zeros(R, G, No, dtype=out_dtype), sig)
    out = torch.full((R, G, No + 3), 7.0, dtype=out_dtype, device=cuda)      # a wider destination: the columns past No stay untouched
    c = lambda t: t.to(cuda)      # noqa: E731
    ops.mlp3_grouped(c(x)[:, 1:], G, ops.mlp3_pack(c(w0)), c(b0), ops.mlp3_pack(c(w1)), c(b1), ops.mlp3_pack(c(w2)), c(b2), out, sig)
    got = out.cpu()
    assert (got[:, :, No:] == 7.0).all()
    tol = dict(rtol=2e-2, atol=2e-2) if out_dtype == torch.bfloat16 else dict(rtol=4e-3, atol=4e-3)      # (bf16 roundings between the layers may differ by one step)
    torch.testing.assert_close(got[:, :, :No].float(), want.float(), **tol)


def test_mlp3_grouped_vs_per_layer_route(cuda):
    """the route it replaces: ops.linear per layer (bf16 between the layers) — same operands, same roundings up to the fp32 summation order"""
    from videoglamm_amd import ops
    G, R, K, Hd, No = 4, 6, 256, 256, 32
    x = rnd(R, G, K, seed=11).to(cuda)
    w0, w1, w2 = rnd(G, Hd, K, seed=12, scale=K ** -0.5).to(cuda), rnd(G, Hd, Hd, seed=13, scale=Hd ** -0.5).to(cuda), rnd(G, No, Hd, seed=14, scale=Hd ** -0.5).to(cuda)
    b0, b1, b2 = (rnd(G, n, seed=15 + i, scale=0.3, dtype=torch.float32).to(cuda) for i, n in enumerate((Hd, Hd, No)))
    got = ops.mlp3_grouped(x, G, ops.mlp3_pack(w0), b0, ops.mlp3_pack(w1), b1, ops.mlp3_pack(w2), b2, torch.empty(R, G, No, dtype=torch.bfloat16, device=cuda))
    for g in range(G):
        h = ops.linear(x[:, g, :].contiguous(), w0[g], b0[g], act=ops.ACT_RELU)
        h = ops.linear(h, w1[g], b1[g], act=ops.ACT_RELU)
        y = ops.linear(h, w2[g], b2[g])
        torch.testing.assert_close(got[:, g, :].float(), y.float(), rtol=2e-2, atol=2e-2)
        assert float((got[:, g, :].float() - y.float()).abs().mean()) < 2e-3


def test_mlp3_grouped_rejects_bad_shapes(cuda):
    from videoglamm_amd import ops
    from videoglamm_amd._lib import VGKernelError
    x = torch.zeros(2, 1, 24, dtype=torch.bfloat16, device=cuda)      # K = 24: not a multiple of 16
    w0, w1, w2 = (torch.zeros(1, a, b, dtype=torch.bfloat16, device=cuda) for a, b in ((32, 24), (32, 32), (4, 32)))
    b0, b1, b2 = (torch.zeros(1, n, device=cuda) for n in (32, 32, 4))
    with pytest.raises(VGKernelError):
        ops._lib.check(ops._lib.load().vg_mlp3_grouped(ops._p(x), 24, 24, ops._p(w0), ops._p(b0), ops._p(w1), ops._p(b1), ops._p(w2), ops._p(b2), ops._p(x), 4, 4, 1, 1, 2, 24, 32, 4, 0, None), 'vg_mlp3_grouped')
