"""the video branch's token-side launches one by one, graph-replayed: which of them sit above the ~2.3 us a minimal launch costs in a replayed graph"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videoglamm_amd import ops
from mlp3_bench import graph_time

N, nt, TP = 1, 9, 16
bf = lambda *s: torch.randn(*s, device="cuda").bfloat16()      # noqa: E731
q, pe = bf(N, nt, 256), bf(N, nt, 256)
lw, lb = torch.randn(256, device="cuda"), torch.randn(256, device="cuda")
print(f"axpby [9, 256]: {graph_time(lambda: ops.add(q, pe)):.1f} us")
print(f"layernorm [9, 256]: {graph_time(lambda: ops.layernorm(q, lw, lb, 1e-5)):.1f} us")
x128 = bf(N, nt, 128)
print(f"heads_blockdiag: {graph_time(lambda: ops.heads_blockdiag(x128, TP)):.1f} us")
kbd = ops.heads_blockdiag(x128, TP)
w = bf(256, 128)
print(f"linear [{kbd.shape[0]}, 128] x [256, 128] (small64): {graph_time(lambda: ops.linear(kbd, w)):.1f} us")
full = bf(N * 8 * TP, 128)
print(f"heads_blockdiag_gather: {graph_time(lambda: ops.heads_blockdiag_gather(full, N, nt, TP)):.1f} us")
qa = bf(N, nt, 8, 32)
print(f"attention 9 x 9, 8 heads x 32: {graph_time(lambda: ops.attention(qa, qa, qa, 32 ** -0.5)):.1f} us")
U, xk = bf(N, 8 * TP, 1, 256), bf(N, 4096, 1, 256)
print(f"token -> image attention (128 queries, 4096 keys, d = 256): {graph_time(lambda: ops.attention(U, xk, xk, 1.0)):.1f} us")
w1 = bf(2048, 256)
print(f"linear [9, 256] x [2048, 256] + relu (skinny): {graph_time(lambda: ops.linear(q.view(-1, 256), w1, act=ops.ACT_RELU)):.1f} us")
