"""vg_mlp3_grouped on the video branch's per-frame shapes against the per-layer launches it replaces (graph-replayed, so launch gaps count as they do in the product)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videoglamm_amd import ops


def graph_time(fn, n=50):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


if __name__ == "__main__":
  for G, R, No in ((4, 1, 32), (4, 8, 32), (1, 1, 4), (1, 8, 256), (4, 128, 32)):
      K = Hd = 256
      x = torch.randn(R, G, K, device="cuda").bfloat16()
      w0, w1, w2 = (torch.randn(G, a, b, device="cuda").bfloat16() * 0.06 for a, b in ((Hd, K), (Hd, Hd), (No, Hd)))
      b0, b1, b2 = (torch.randn(G, n, device="cuda") * 0.1 for n in (Hd, Hd, No))
      out = torch.empty(R, G, No, device="cuda", dtype=torch.bfloat16)
      p0, p1, p2 = ops.mlp3_pack(w0), ops.mlp3_pack(w1), ops.mlp3_pack(w2)
      t1 = graph_time(lambda: ops.mlp3_grouped(x, G, p0, b0, p1, b1, p2, b2, out))
      xs = [x[:, g, :].contiguous() for g in range(G)]

      def per_layer():
          for g in range(G):
              h = ops.linear(xs[g], w0[g], b0[g], act=ops.ACT_RELU)
              h = ops.linear(h, w1[g], b1[g], act=ops.ACT_RELU)
              ops.linear(h, w2[g], b2[g], out=out[:, g, :])
      t2 = graph_time(per_layer)
      print(f"G={G} R={R} No={No}: grouped {t1:.1f} us, {3 * G} per-layer launches {t2:.1f} us")
