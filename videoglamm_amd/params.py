"""Checkpoint ingest: reference state-dict names -> device tensors in the layouts the HIP kernels want.

The reference consumes one merged HF directory via from_pretrained (R/chat.py:277-319); SAM2 tensors use
the ".gamma -> .weight" rename of R/model/segment_anything_2/sam2/build_sam.py:92-112.  Everything here is
one-time load/repack work (pad K to a multiple of 8, conv -> GEMM layouts, fused qkv / gate|up, fp32
copies of bias / norm vectors); none of it runs inside the timed hot path.
"""
import torch


class Params:
    """Lazy, cached packer over a {name: tensor} source (CPU or already-on-device tensors)."""

    def __init__(self, sd, device, dtype):
        self.sd = {k.replace(".gamma", ".weight") if "fuser.layers" in k else k: v for k, v in sd.items()}
        self.device = torch.device(device)
        self.dtype = dtype
        self.kalign = 8 if dtype == torch.bfloat16 else 4
        self._c = {}

    def has(self, name):
        return name in self.sd

    def _get(self, key, fn):
        if key not in self._c:
            self._c[key] = fn()
        return self._c[key]

    def _padk(self, w):
        K = w.shape[1]
        Kp = -(-K // self.kalign) * self.kalign
        if Kp != K:
            w = torch.nn.functional.pad(w, (0, Kp - K))
        return w.to(device=self.device, dtype=self.dtype).contiguous()

    # ---- dense
    def w(self, name):
        """nn.Linear / 1x1 conv weight -> [N, Kpad] in the model dtype."""
        return self._get(("w", name), lambda: self._padk(self.sd[name + ".weight"].reshape(self.sd[name + ".weight"].shape[0], -1)))

    def f32(self, name):
        """bias / norm weight / LayerScale vector -> fp32 contiguous (None when absent)."""
        if name not in self.sd:
            return None
        return self._get(("f32", name), lambda: self.sd[name].reshape(-1).to(device=self.device, dtype=torch.float32).contiguous())

    def b(self, name):
        return self.f32(name + ".bias")

    def t(self, name, dtype=None):
        """any tensor as-is on device (model dtype unless given)."""
        dt = dtype or self.dtype
        return self._get(("t", name, dt), lambda: self.sd[name].to(device=self.device, dtype=dt).contiguous())

    def fused(self, names, stored=None):
        """row-concatenation of several [N_i, K] weights (qkv, gate|up) -> ([sum N_i, Kpad], fp32 bias or None).
        stored: name of a checkpoint tensor that already IS that concatenation (Phi-3's qkv_proj / gate_up_proj)."""
        if stored is not None and stored + ".weight" in self.sd:
            return self.w(stored), self.b(stored)

        def make():
            w = torch.cat([self.sd[n + ".weight"] for n in names], dim=0)
            bs = [self.sd.get(n + ".bias") for n in names]
            bias = None
            if any(x is not None for x in bs):
                ref = next(x for x in bs if x is not None)
                bias = torch.cat([x if x is not None else torch.zeros(self.sd[n + ".weight"].shape[0], dtype=ref.dtype, device=ref.device)
                                  for x, n in zip(bs, names)])
                bias = bias.to(device=self.device, dtype=torch.float32).contiguous()
            return self._padk(w), bias
        return self._get(("fused",) + tuple(names), make)

    def fp8(self, names, stored=None):
        """fused()/w() of the named weights quantised for the decode GEMVs: (uint8 [N,K] OCP-e4m3 codes, fp32 [N] row scales).
        Made once from the packed model-dtype weight (so the bf16 prefill and the fp8 decode see the same rounding of the
        checkpoint first); kept beside it — the prefill GEMMs stay bf16."""
        from . import ops

        names = [names] if isinstance(names, str) else list(names)

        def make():
            w = self.w(names[0]) if len(names) == 1 and stored is None else self.fused(names, stored=stored)[0]
            return ops.quantize_fp8_rows(w)
        return self._get(("fp8",) + tuple(names), make)

    # ---- convolutions as GEMMs
    def conv_w(self, name):
        """Conv2d [Cout,Cin,kh,kw] (or Conv3d with kt=1) -> [Cout, (ky*kw+kx)*Cin + c] padded (vg_im2col column order)."""
        def make():
            w = self.sd[name + ".weight"]
            if w.dim() == 5:
                w = w[:, :, 0]
            return self._padk(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1))
        return self._get(("conv", name), make)

    def convT_w(self, name):
        """ConvTranspose2d k2 s2 [Cin,Cout,2,2] -> [(dy*2+dx)*Cout + co, Cin] (vg_pixel_shuffle2 tap order)."""
        def make():
            w = self.sd[name + ".weight"]
            return self._padk(w.permute(2, 3, 1, 0).reshape(4 * w.shape[1], w.shape[0]))
        return self._get(("convT", name), make)

    def dw_w(self, name):
        """depthwise Conv2d [C,1,k,k] -> fp32 [k*k, C]."""
        def make():
            w = self.sd[name + ".weight"]
            return w[:, 0].permute(1, 2, 0).reshape(-1, w.shape[0]).to(device=self.device, dtype=torch.float32).contiguous()
        return self._get(("dw", name), make)

    def const(self, key, fn, dtype=None):
        """derived constant (positional tables, interpolated pos-embeds) computed once on the host."""
        dt = dtype or self.dtype
        return self._get(("const", key, dt), lambda: fn().to(device=self.device, dtype=dt).contiguous())
