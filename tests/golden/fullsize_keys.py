"""Fixed sampling strides of the full-size fixtures (tests/golden/make_golden_fullsize.py writes with them, the tests read with them).
Each entry: strides over the LAST len(strides) dimensions of the tensor."""
SUB = {
    "fpn0": (4, 8, 8),        # [1, 32, 256, 256] -> channels ::4, pixels ::8
    "fpn1": (8, 4, 4),        # [1, 64, 128, 128]
    "fpn2": (16, 2, 2),       # [1, 256, 64, 64]
    "low": (2, 2),            # [..., 256, 256] low-res mask logits
    "logits": (4, 4),         # [..., 480, 640] mask logits at the video's size
    "maskmem": (4, 2, 2),     # [N, 64, 64, 64] memory features
    "tokens16": (16, 1),      # [B, L, C]: every 16th token
}


def sub(kind, t):
    st = SUB[kind]
    idx = (Ellipsis,) + tuple(slice(None, None, s) for s in st)
    return t[idx].contiguous()
