"""r06 decode-step kernels (-m gpu): vg_decode_qkv_rope (RoPE + KV append in the q|k|v GEMV's epilogue), vg_decode_attention2 (wave-private flash
pass, split merge through write-through partials), vg_decode_advance (the loop's bookkeeping on the device) — each against the kernels /
statements it replaces, and the decoder / generate() loop built on them against the r05 path (VG_DECODE_ROPE=0, VG_DECODE_AHEAD=0)."""
import pytest
import torch

import _cpu_ops as ref

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def tables(max_len, D, theta=500000.0):
    ang = torch.arange(max_len)[:, None].float() * (1.0 / (theta ** (torch.arange(0, D, 2).float() / D)))[None]
    return ang.cos().contiguous(), ang.sin().contiguous()


@pytest.mark.parametrize("dtype,K", [(torch.bfloat16, 4096), (torch.bfloat16, 2048), (torch.float32, 4096)])
@pytest.mark.parametrize("H,Hkv", [(32, 8), (8, 8), (16, 2)])
def test_decode_qkv_rope_equals_gemv_then_rope(cuda, dtype, K, H, Hkv):
    """norm -> q|k|v -> RoPE -> append in one launch == vg_decode_gemv followed by vg_rope_kv_append, bit for bit (bf16; fp32 to 1 ulp: the
    compiler contracts x1*c - x2*s differently in two kernels), at several positions; the rest of the caches untouched."""
    from videoglamm_amd import ops
    D, max_len = 128, 512
    x = rnd(1, K, dtype=dtype, seed=1)
    w = rnd((H + 2 * Hkv) * D, K, dtype=dtype, seed=2, scale=K ** -0.5).to(cuda)
    nw = (1.0 + 0.1 * rnd(K, seed=3)).float().to(cuda)
    cos, sin = tables(max_len, D)
    gcos, gsin = cos.to(cuda), sin.to(cuda)
    kc0, vc0 = rnd(max_len, Hkv, D, dtype=dtype, seed=4), rnd(max_len, Hkv, D, dtype=dtype, seed=5)
    rope_cs = torch.zeros(D, dtype=torch.float32, device=cuda)
    for pos in (0, 1, 77, max_len - 1):
        pos_dev = torch.tensor([pos], dtype=torch.int32, device=cuda)
        ops.decode_advance_(pos_dev, 0, rope=(gcos, gsin, rope_cs))
        assert torch.equal(rope_cs.cpu(), torch.cat([cos[pos], sin[pos]]))
        a_kc, a_vc, b_kc, b_vc = kc0.to(cuda), vc0.to(cuda), kc0.to(cuda), vc0.to(cuda)
        q = ops.decode_qkv_rope(x.to(cuda), w, nw, 1e-5, a_kc, a_vc, rope_cs, pos_dev, H, Hkv, D)
        qkv = ops.decode_gemv(x.to(cuda), w, norm_w=nw, eps=1e-5)
        ops.rope_kv_append_(qkv, b_kc, b_vc, gcos, gsin, H, Hkv, D, 0, pos_dev)
        if dtype == torch.bfloat16:
            assert torch.equal(q, qkv[:, : H * D]) and torch.equal(a_kc, b_kc) and torch.equal(a_vc, b_vc)
        else:
            torch.testing.assert_close(q, qkv[:, : H * D], rtol=1e-6, atol=1e-6)
            torch.testing.assert_close(a_kc, b_kc, rtol=1e-6, atol=1e-6)
            assert torch.equal(a_vc, b_vc)
        keep = torch.ones(max_len, dtype=torch.bool)
        keep[pos] = False
        assert torch.equal(a_kc.cpu()[keep], kc0[keep]) and torch.equal(a_vc.cpu()[keep], vc0[keep])


@pytest.mark.parametrize("H,Hkv", [(32, 8), (8, 8), (16, 8), (16, 2)])
@pytest.mark.parametrize("kpw", [128, 256])
def test_decode_attention2(cuda, H, Hkv, kpw):
    """the wave-private flash pass vs the fp32 statement on the same bf16 values, at block boundaries of both granularities, with and without a
    sliding window, replayed on ONE workspace (the arrival counters reset themselves); garbage-free masking: rows past the position hold large
    finite values and must not leak; and against the kernel it replaces."""
    from videoglamm_amd import ops
    D, max_len = 128, 2048
    dtype = torch.bfloat16
    kc, vc = rnd(max_len, Hkv, D, dtype=dtype, seed=2), rnd(max_len, Hkv, D, dtype=dtype, seed=3)
    cos, sin = tables(max_len, D)
    ws = ops.decode_attention_workspace(H, Hkv, D, max_len, cuda)
    ws_old = ops.decode_attention_workspace(H, Hkv, D, max_len, cuda)
    t = dict(rtol=3e-2, atol=2e-2)
    for window in (0, 300):
        for pos in (0, 1, 31, 32, 33, 127, 128, 129, 255, 256, 257, 1000, max_len - 1):
            q = rnd(1, H * D, dtype=dtype, seed=10 + pos)
            pos_dev = torch.tensor([pos], dtype=torch.int32)
            g_kc, g_vc = kc.clone(), vc.clone()
            g_kc[pos + 1:] = 3.0e4                      # what a previous, longer clip left behind: finite, huge — masked, never read into the result
            g_vc[pos + 1:] = -3.0e4
            o = ops.decode_attention2(q.to(cuda), g_kc.to(cuda), g_vc.to(cuda), H, Hkv, D, pos_dev.to(cuda), D ** -0.5, ws, window=window, keys_per_wg=kpw)
            want = ref.attention_decode(q.view(1, 1, H, D), kc, vc, pos_dev, D ** -0.5, window).view(1, H * D)
            assert torch.isfinite(o).all()
            torch.testing.assert_close(o.float().cpu(), want.float(), **t)
    assert int(ws[-Hkv:].view(torch.int32).abs().sum()) == 0
    # the same step through the r05 kernel (RoPE + append inside): q pre-rotation and the append are exactly its arithmetic, so feed it the raw row
    for pos in (5, 700, 1999):
        qkv = rnd(1, (H + 2 * Hkv) * D, dtype=dtype, seed=50 + pos).to(cuda)
        pos_dev = torch.tensor([pos], dtype=torch.int32, device=cuda)
        a_kc, a_vc, b_kc, b_vc = kc.to(cuda), vc.to(cuda), kc.to(cuda), vc.to(cuda)
        o_old = ops.decode_attention(qkv, a_kc, a_vc, cos.to(cuda), sin.to(cuda), H, Hkv, D, pos_dev, D ** -0.5, ws_old)
        r = qkv.clone()
        ops.rope_kv_append_(r, b_kc, b_vc, cos.to(cuda), sin.to(cuda), H, Hkv, D, 0, pos_dev)
        o_new = ops.decode_attention2(r[:, : H * D].contiguous(), b_kc, b_vc, H, Hkv, D, pos_dev, D ** -0.5, ws, keys_per_wg=kpw)
        torch.testing.assert_close(o_new.float(), o_old.float(), rtol=1.6e-2, atol=4e-3)


def test_decode_attention2_graph_replay_and_stress(cuda):
    """64 launches of the attention in one captured graph, replayed 20 times on one workspace, at the C2 position: every replay returns the same
    bits (the ticket / merge is order-independent: each partial is merged by index, not by arrival), and the counters end at zero."""
    from videoglamm_amd import ops
    H, Hkv, D, max_len, pos = 32, 8, 128, 4096, 3391
    kc, vc = rnd(max_len, Hkv, D, dtype=torch.bfloat16, seed=2).to(cuda), rnd(max_len, Hkv, D, dtype=torch.bfloat16, seed=3).to(cuda)
    q = rnd(1, H * D, dtype=torch.bfloat16, seed=4).to(cuda)
    pos_dev = torch.tensor([pos], dtype=torch.int32, device=cuda)
    ws = ops.decode_attention_workspace(H, Hkv, D, max_len, cuda)
    first = ops.decode_attention2(q, kc, vc, H, Hkv, D, pos_dev, D ** -0.5, ws).clone()
    want = ref.attention_decode(q.cpu().view(1, 1, H, D), kc.cpu(), vc.cpu(), pos_dev.cpu(), D ** -0.5).view(1, H * D)
    torch.testing.assert_close(first.float().cpu(), want.float(), rtol=3e-2, atol=2e-2)
    outs = []
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        for _ in range(64):
            outs.append(ops.decode_attention2(q, kc, vc, H, Hkv, D, pos_dev, D ** -0.5, ws))
    for _ in range(20):
        g.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(o, first) for o in outs)
    assert int(ws[-Hkv:].view(torch.int32).abs().sum()) == 0


def test_decode_advance(cuda):
    from videoglamm_amd import ops
    D, max_len = 128, 64
    cos, sin = tables(max_len, D)
    g = dict(tok=torch.tensor([11], dtype=torch.int64), pos=torch.tensor([7], dtype=torch.int32), step=torch.zeros(1, dtype=torch.int32),
             forced=torch.tensor([-1, 500, -1, 7], dtype=torch.int64), hist=torch.zeros(6, dtype=torch.int64), raw=torch.zeros(6, dtype=torch.int64),
             cs=torch.zeros(D))
    c = {k: v.clone() for k, v in g.items()}
    d = {k: v.to(cuda) for k, v in g.items()}
    for i, inc in enumerate((0, 1, 1, 1, 1)):
        for s, o in ((c, ref), (d, ops)):
            s["tok"].fill_(100 + i)
            o.decode_advance_(s["pos"], inc, s["tok"], s["step"], s["forced"], s["hist"], s["raw"], rope=(cos.to(s["cs"].device), sin.to(s["cs"].device), s["cs"]))
        for k in c:
            assert torch.equal(c[k], d[k].cpu()), (i, k, c[k], d[k])
    assert c["hist"].tolist() == [100, 500, 102, 7, 104, 0] and c["raw"].tolist() == [100, 101, 102, 103, 104, 0] and int(c["pos"]) == 11


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_decode_step_begin_and_end_equal_the_separate_launches(cuda, dtype):
    """vg_decode_step_begin == vg_embed + the cos / sin row; vg_argmax_partial + vg_decode_step_end == vg_argmax + vg_store_row + vg_decode_advance(inc = 1):
    same token (ties -> lowest index), same stored row, same history / forcing, accumulator left zero, replayed three times on one accumulator."""
    from videoglamm_amd import ops
    D, V, max_len, hd = 4096, 128257, 64, 64
    cos, sin = tables(max_len, 2 * hd)
    gcos, gsin = cos.to(cuda), sin.to(cuda)
    table = torch.randn(V, D, device=cuda, dtype=dtype, generator=torch.Generator(device=cuda).manual_seed(1))      # (the emitted ids index it: full vocabulary)
    st = dict(tok=torch.tensor([17], dtype=torch.int64), pos=torch.tensor([9], dtype=torch.int32), step=torch.zeros(1, dtype=torch.int32),
              forced=torch.tensor([-1, 250, -1], dtype=torch.int64), hist=torch.zeros(4, dtype=torch.int64), raw=torch.zeros(4, dtype=torch.int64))
    a = {k: v.to(cuda) for k, v in st.items()}
    b = {k: v.to(cuda) for k, v in st.items()}
    hid_a, hid_b = torch.zeros(max_len, D, dtype=dtype, device=cuda), torch.zeros(max_len, D, dtype=dtype, device=cuda)
    cs_a, cs_b = torch.zeros(2 * hd, device=cuda), torch.zeros(2 * hd, device=cuda)
    acc = torch.zeros(1, dtype=torch.int64, device=cuda)
    for i in range(3):
        logits = rnd(1, V, seed=20 + i).to(cuda)
        logits[0, 5000 + i] = logits[0, 70000 + i] = 9.0          # a tie: the lower index wins
        row = rnd(1, D, dtype=dtype, seed=30 + i).to(cuda)
        # fused
        xa = ops.decode_step_begin(a["tok"], table, a["pos"], rope=(gcos, gsin, cs_a))
        ops.argmax_partial(logits.view(-1), acc)
        ops.decode_step_end(acc, a["tok"], a["pos"], a["step"], row, hid_a, a["forced"], a["hist"], a["raw"])
        # separate launches
        xb = ops.embed(b["tok"], table)
        ops.decode_advance_(b["pos"], 0, rope=(gcos, gsin, cs_b))
        ops.store_row_(row, hid_b, b["pos"])
        ops.argmax(logits, out=b["tok"])
        ops.decode_advance_(b["pos"], 1, b["tok"], b["step"], b["forced"], b["hist"], b["raw"])
        assert torch.equal(xa, xb) and torch.equal(cs_a, cs_b) and torch.equal(hid_a, hid_b) and int(acc[0]) == 0
        for k in a:
            assert torch.equal(a[k], b[k]), (i, k, a[k], b[k])
    assert a["hist"].tolist() == [5000, 250, 5002, 0] and a["raw"].tolist() == [5000, 5001, 5002, 0] and int(a["pos"]) == 12


def _llama2(S, layers=2):
    from oracle import seeded
    from videoglamm_amd import synth
    c = dict(synth.LLAMA3_8B, num_layers=layers, vocab=8192)
    man = {k: v for k, v in synth.vlm_manifest(dict(synth.videoglamm_llama3_8b(), llm=c)).items()
           if k.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))}
    sd = seeded.seeded_state_dict(man, 5)
    sd = {k: (v.to(torch.bfloat16) if v.dim() >= 2 else v) for k, v in sd.items()}
    x = (torch.randn(S, c["hidden"], generator=torch.Generator().manual_seed(3)) * 0.5).to(torch.bfloat16)
    return c, sd, x


def test_decoder_rope_path_vs_r05_path(cuda, monkeypatch):
    """2 layers at Llama-3-8B width, 300-row prefill, 12 teacher-forced tokens through decode_step() (graph replay): the r06 launches
    (VG_DECODE_ROPE=1) against the r05 ones — the KV rows they append are IDENTICAL (same projection, same RoPE arithmetic), the hidden rows agree
    to the attention's fp32 summation order under bf16 activations, the greedy tokens are the same."""
    from videoglamm_amd.params import Params
    from videoglamm_amd.vlm import LlamaDecoder
    S, G = 300, 12
    c, sd, x = _llama2(S)
    P = Params(sd, cuda, torch.bfloat16)
    toks = torch.randint(0, c["vocab"], (G,), generator=torch.Generator().manual_seed(11))
    got = {}
    for rope in ("1", "0"):
        monkeypatch.setenv("VG_DECODE_ROPE", rope)
        dec = LlamaDecoder(P, c, 1024, use_graph=True)
        assert dec.rope_path == (rope == "1")
        dec.forward(x[:S].to(cuda))
        rows, picks = [], []
        for i in range(G):
            dec.tok_dev.fill_(int(toks[i]))
            dec.decode_step()
            rows.append(dec.hid_all[dec.pos - 1].clone())
            picks.append(int(dec.tok_dev[0]))
        assert int(dec.pos_dev[0]) == S + G == dec.pos and int(dec.step_dev[0]) == G
        got[rope] = (torch.stack(rows).float(), picks, dec.kc[0][S:S + G].clone(), dec.vc[1][S:S + G].clone())
        del dec
    a, b = got["1"], got["0"]
    assert torch.isfinite(a[0]).all()
    assert torch.equal(a[2], b[2])                                   # layer 0's appended keys: bit-identical
    rel = (a[0] - b[0]).norm(dim=1) / b[0].norm(dim=1)
    assert rel.max() < 2e-2, float(rel.max())
    assert sum(p == q for p, q in zip(a[1], b[1])) >= G - 1


def test_generate_runs_ahead_equals_synchronous(cuda, monkeypatch):
    """vlm.generate's device-side loop (step k + 1 enqueued before token k is read) against the synchronous hand-over: same ids with plain
    greedy decoding, with a forced-token table (incl. step 0 and the last step), and with an EOS that stops the loop early (the step that was
    in flight is wasted, not observed); trace['argmax'] holds the model's own choices before forcing in both."""
    from videoglamm_amd import synth, vlm
    from videoglamm_amd.params import Params
    S = 40
    c, sd, x = _llama2(S)
    sd = dict(sd)
    sd["model.text_hidden_fcs.0.0.weight"] = rnd(4096, 4096, dtype=torch.bfloat16, seed=1, scale=0.02)
    sd["model.text_hidden_fcs.0.0.bias"] = rnd(4096, dtype=torch.bfloat16, seed=2, scale=0.02)
    sd["model.text_hidden_fcs.0.2.weight"] = rnd(256, 4096, dtype=torch.bfloat16, seed=3, scale=0.02)
    sd["model.text_hidden_fcs.0.2.bias"] = rnd(256, dtype=torch.bfloat16, seed=4, scale=0.02)
    P = Params(sd, cuda, torch.bfloat16)
    cfg = dict(llm=c, seg_token_idx=8191)
    ids = torch.randint(0, 8000, (S,), generator=torch.Generator().manual_seed(5))

    def run(ahead, hook, eos, n=10):
        monkeypatch.setenv("VG_DECODE_AHEAD", ahead)
        tr = {}
        out, emb = vlm.generate(P, cfg, None, None, None, ids, n, eos_token_id=eos, visual=torch.empty(0, 4096, dtype=torch.bfloat16, device=cuda),
                                token_hook=hook, trace=tr)
        return out.tolist(), emb.float().cpu(), tr["argmax"]

    base = run("0", None, None)
    assert run("1", None, None)[0] == base[0] and len(base[0]) == S + 10
    table = {0: 8191, 3: 17, 9: 8191}
    f0, f1 = run("0", synth.forced_tokens_hook(table), None), run("1", synth.forced_tokens_hook(table), None)
    assert f0[0] == f1[0] and f0[2] == f1[2] and f1[0][S] == 8191 and f1[0][S + 3] == 17 and f1[0][S + 9] == 8191
    assert f1[1].shape == (2, 256) and torch.equal(f0[1], f1[1])
    eos = base[0][S + 4]                                              # the 5th greedy token as EOS: generation stops there
    e0, e1 = run("0", None, eos), run("1", None, eos)
    assert e0[0] == e1[0] and len(e1[0]) <= S + 5 and e1[0][-1] == eos
    # an opaque Python hook keeps the synchronous loop (and still works)
    h = run("1", lambda step, tok: 17 if step == 3 else None, None)
    assert h[0][S + 3] == 17 and h[0][:S + 3] == base[0][:S + 3]
