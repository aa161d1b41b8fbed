"""vg_attention_dma.hip (-m gpu): the LDS-DMA-staged flash kernel that takes the long bf16 sequences (LLM prefill d = 128 causal, Hiera's global
blocks d = 72, the towers d = 64 / 88) against the fp32 statement on the same bf16-rounded operands — ragged lengths on both axes, GQA, Skv > Sq
with the causal diagonal shifted, strided (fused q|k|v) operands, head dims below the padded tile width."""
import pytest
import torch

import _cpu_ops as ref

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


@pytest.mark.parametrize("B,H,Hkv,Sq,Skv,D,causal", [
    (1, 8, 2, 1300, 1300, 128, True),      # LLM prefill shape family: GQA, causal, ragged (5.08 query tiles of 256, 20.3 key tiles of 64)
    (1, 4, 4, 600, 777, 128, True),        # more keys than queries: the diagonal starts at key 177
    (1, 4, 4, 513, 513, 128, False),
    (2, 4, 4, 1024, 1024, 72, False),      # Hiera's global blocks (head dim 72 in a 96-wide tile)
    (1, 2, 2, 4096, 4096, 72, False),
    (3, 4, 4, 1025, 1025, 64, False),      # CLIP
    (2, 4, 4, 1025, 1025, 88, False),      # InternVideo2
    (1, 2, 2, 700, 70, 96, False),         # few keys: two key tiles, the second one ragged
])
def test_attention_dma_vs_fp32_statement(cuda, B, H, Hkv, Sq, Skv, D, causal):
    from videoglamm_amd import ops
    q, k, v = rnd(B, Sq, H, D, seed=1), rnd(B, Skv, Hkv, D, seed=2), rnd(B, Skv, Hkv, D, seed=3)
    want = ref.attention(q, k, v, D ** -0.5, causal).float()
    got = ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), D ** -0.5, causal).float().cpu()
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got, want, rtol=2e-2, atol=2e-2)
    # sharper than the bf16 output rounding: the mean absolute error is that of rounding alone (a mis-masked tile or a wrong fragment moves it by orders)
    assert float((got - want).abs().mean()) < 2e-3


def test_attention_dma_strided_fused_qkv(cuda):
    """q, k, v as views of ONE fused projection [B, S, 3, H, D] (what the towers and Hiera hand over): token / head strides that are not the dense ones"""
    from videoglamm_amd import ops
    B, S, H, D = 2, 1025, 4, 88
    qkv = rnd(B, S, 3, H, D, seed=5)
    g = qkv.to(cuda)
    got = ops.attention(g[:, :, 0], g[:, :, 1], g[:, :, 2], D ** -0.5).float().cpu()
    want = ref.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], D ** -0.5).float()
    torch.testing.assert_close(got, want, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("B,Sq,Skv", [
    (1, 4096, 4096 + 4),         # the video branch's first propagated frame: one memory + one pointer (4 tokens): split-KV, last split ragged
    (1, 4096, 7 * 4096 + 64),    # steady state, one object: 16 query tiles x many splits
    (8, 4096, 7 * 4096 + 64),    # eight objects (sliced check)
    (2, 300, 1000),              # ragged query tile (300 = 256 + 44), several splits
    (1, 256, 130),               # no split, three key tiles, the last with two keys
    (3, 511, 64),                # one key tile
])
def test_attention_dma_dv_vs_fp32_statement(cuda, B, Sq, Skv):
    """attn_dma_d256v64_kernel (vg_attention_dv's bf16 route for Sq >= 256): keys of 256 dims, 64-wide values, split-KV partials merged by
    attn_combine_kernel; a key spike and a few large value rows force the deferred running-max update."""
    from videoglamm_amd import ops
    D, DV = 256, 64
    q, k, v = rnd(B, Sq, 1, D, seed=1), rnd(B, Skv, 1, D, seed=2), rnd(B, Skv, 1, DV, seed=3)
    v[:, Skv // 3] *= 6.0
    k[:, Skv // 2] *= 4.0
    k[:, -1] = q[:, 7] * 0.5
    v[:, -1] += 8.0                  # a dropped or duplicated tail key shows
    got = ops.attention_dv(q.to(cuda), k.to(cuda), v.to(cuda), D ** -0.5).float().cpu()
    assert got.shape == (B, Sq, 1, DV) and torch.isfinite(got).all()
    sl = slice(0, Sq) if B * Sq * Skv <= 3e8 else slice(100, 612)
    for b in sorted({0, B - 1}):
        want = ref.attention(q[b:b + 1, sl], k[b:b + 1], v[b:b + 1], D ** -0.5).float()
        torch.testing.assert_close(got[b:b + 1, sl], want, rtol=3e-2, atol=3e-2)
        assert float((got[b:b + 1, sl] - want).abs().mean()) < 2e-3


def test_attention_dma_dv_every_tail(cuda):
    """every residue of Skv inside a 64-key tile, with one split and with several"""
    from videoglamm_amd import ops
    D, DV, Sq = 256, 64, 256
    q = rnd(1, Sq, 1, D, seed=1)
    for base in (64, 1024):
        for r in range(0, 64, 1):
            Skv = base + r
            k, v = rnd(1, Skv, 1, D, seed=2 + r), rnd(1, Skv, 1, DV, seed=3 + r)
            v[:, -1] += 8.0
            k[:, -1] = q[:, 7] * 0.5
            got = ops.attention_dv(q.to(cuda), k.to(cuda), v.to(cuda), D ** -0.5).float().cpu()
            torch.testing.assert_close(got, ref.attention(q, k, v, D ** -0.5).float(), rtol=3e-2, atol=3e-2)


def test_attention_dma_full_size_properties(cuda):
    """Size-independent properties at the clip's FULL sizes (no fp32 statement fits there):
    (a) the memory cross-attention (4096 queries x 28 736 keys, d = 256, 64-wide values, eight objects) is linear in V: attn(q, k, v1 + v2) = attn(q, k, v1) +
        attn(q, k, v2) up to the bf16 output rounding, and a batch entry's rows do not depend on which other objects run beside it (bit-equal alone / in the batch);
    (b) the causal LLM prefill shape (3361 rows, 32 query heads on 8 KV heads, d = 128): a head's output is bit-equal whether half of the heads run alone or inside the
        32-head launch (the longest-first launch order permutes the workgroups, never the arithmetic), and row i does not change when later rows are cut off."""
    from videoglamm_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    B, Sq, Skv, D, DV = 8, 4096, 7 * 4096 + 64, 256, 64
    q = (torch.randn(B, Sq, 1, D, generator=g) * 0.5).to(torch.bfloat16).to(cuda)
    k = (torch.randn(B, Skv, 1, D, generator=g) * 0.5).to(torch.bfloat16).to(cuda)
    v1 = torch.randn(B, Skv, 1, DV, generator=g).to(torch.bfloat16).to(cuda)
    v2 = torch.randn(B, Skv, 1, DV, generator=g).to(torch.bfloat16).to(cuda)
    o1, o2 = ops.attention_dv(q, k, v1, D ** -0.5).float(), ops.attention_dv(q, k, v2, D ** -0.5).float()
    o12 = ops.attention_dv(q, k, (v1.float() + v2.float()).to(torch.bfloat16), D ** -0.5).float()
    assert torch.isfinite(o12).all()
    # outputs are means of ~N(0, 1) values over many keys: |o| ~ 1e-2 .. 1e-1; the three bf16 roundings bound the defect
    assert float((o12 - (o1 + o2)).abs().max()) < 3e-2 and float((o12 - (o1 + o2)).abs().mean()) < 2e-3
    alone = ops.attention_dv(q[5:6], k[5:6], v1[5:6], D ** -0.5).float()
    assert float((alone - o1[5:6]).abs().max()) < 2e-2       # (one object takes another KV split than eight: partial sums associate differently)
    same = ops.attention_dv(q[4:6], k[4:6], v1[4:6], D ** -0.5).float()
    assert float((same[1:] - o1[5:6]).abs().max()) < 2e-2

    S, H, Hkv, D = 3361, 32, 8, 128
    q = torch.randn(1, S, H, D, generator=g).to(torch.bfloat16).to(cuda)
    k = torch.randn(1, S, Hkv, D, generator=g).to(torch.bfloat16).to(cuda)
    v = torch.randn(1, S, Hkv, D, generator=g).to(torch.bfloat16).to(cuda)
    full = ops.attention(q, k, v, D ** -0.5, True)
    # KV heads 2 .. 5 with their query heads 8 .. 23, as strided views of the same tensors (16 heads still fill the chip without a KV split: a split launch merges
    # partial softmaxes, another association of the same sums — 8 heads alone differ from the 32-head launch in the last bf16 bit of 17 % of the outputs)
    grp = ops.attention(q[:, :, 8:24], k[:, :, 2:6], v[:, :, 2:6], D ** -0.5, True)
    assert torch.equal(grp, full[:, :, 8:24])
    few = ops.attention(q[:, :, 8:16], k[:, :, 2:4], v[:, :, 2:4], D ** -0.5, True)      # 8 heads: split-KV + merge
    assert float((few.float() - full[:, :, 8:16].float()).abs().max()) < 4e-3
    cut = ops.attention(q[:, :1800].contiguous(), k[:, :1800].contiguous(), v[:, :1800].contiguous(), D ** -0.5, True)
    assert torch.equal(cut[:, :1536], full[:, :1536])       # rows of whole 256-row query tiles: the same key tiles in the same order
    assert float((cut.float() - full[:, :1800].float()).abs().max()) < 2e-2
