"""videoglamm_amd.synth manifests vs manifests dumped from the reference's own nn.Modules
(tests/golden/*_manifest.json): every tensor the hot path reads must exist with the same shape."""
import _golden as G
from videoglamm_amd import synth


def test_sam2_manifest_matches_reference():
    ref = G.manifest("sam2_micro_manifest.json")
    got = synth.sam2_manifest(G.sam2_cfg())
    assert set(got) == set(ref), (sorted(set(ref) - set(got))[:5], sorted(set(got) - set(ref))[:5])
    assert all(list(got[k]) == list(ref[k]) for k in ref)


def test_vlm_manifest_subset_of_reference():
    E = G.configs.E2E
    cfg = dict(iv2=dict(E["iv2"], mlp_hidden=int(E["iv2"]["embed_dim"] * E["iv2"]["mlp_ratio"])), clip=E["clip"], llm=E["llm"],
               sam2=dict(image_size=1024, trunk=G.configs.SAM2_E2E["trunk"]), projector_depth=2)
    ref = G.manifest("e2e_manifest.json")
    got = synth.manifest(cfg)
    norm = {k.replace("vision_tower.vision_model.", "vision_tower."): v for k, v in got.items()}
    missing = [k for k in norm if k not in ref]
    assert not missing, missing[:8]
    bad = [k for k in norm if list(norm[k]) != list(ref[k])]
    assert not bad, [(k, norm[k], ref[k]) for k in bad[:5]]
