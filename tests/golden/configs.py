"""Micro configurations shared by the golden generator, the oracle tests and the GPU parity tests."""

SAM2_MICRO = dict(
    image_size=256,
    trunk=dict(embed_dim=16, num_heads=1, stages=[1, 2, 3, 1], global_att_blocks=[4, 5], window_spec=[8, 4, 8, 4],
               window_pos_embed_bkg_spatial_size=[7, 7]),
    neck_channels=[128, 64, 32, 16],
)

# the reference hard-codes SAM2 feature sizes for 1024^2 inputs in the framewise path
# (R/model/VideoGLaMM.py:228), so the end-to-end fixture keeps image_size 1024 with the micro trunk
SAM2_E2E = dict(image_size=1024, trunk=SAM2_MICRO["trunk"], neck_channels=SAM2_MICRO["neck_channels"])

IV2_TINY = dict(img_size=56, patch_size=14, embed_dim=64, depth=3, num_heads=4, mlp_ratio=4.0)
CLIP_TINY = dict(img_size=56, patch_size=14, hidden=64, mlp=128, num_layers=3, num_heads=4)
LLAMA_TINY = dict(vocab=320, hidden=64, ffn=176, num_layers=2, num_heads=4, num_kv_heads=2, rms_eps=1e-5, rope_theta=10000.0)

# end-to-end: the widths arch.py hard-codes (IV2 D=1408/L=256, CLIP 1024) with minimal depth
E2E = dict(
    te=4, t_sam=4, max_new_tokens=6, seg_token_idx=300,
    iv2=dict(img_size=224, patch_size=14, embed_dim=1408, depth=2, num_heads=16, mlp_ratio=48 / 11),
    clip=dict(img_size=336, patch_size=14, hidden=1024, mlp=512, num_layers=2, num_heads=16),
    llm=dict(vocab=320, hidden=64, ffn=176, num_layers=2, num_heads=4, num_kv_heads=2, rms_eps=1e-5, rope_theta=10000.0),
)
# Phi-3-mini shape family (the LLM of the released VideoGLaMM checkpoint): fused qkv_proj / gate_up_proj, MHA
PHI3_TINY = dict(vocab=320, hidden=64, ffn=176, num_layers=2, num_heads=4, num_kv_heads=4, rms_eps=1e-5, rope_theta=10000.0,
                 sliding_window=2047)
# the same decoder with a window shorter than the fixtures' 45-token sequence (4.41 semantics: position i sees [i - 11, i], 12 keys)
PHI3_TINY_WIN = dict(PHI3_TINY, sliding_window=11)
