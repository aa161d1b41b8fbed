"""does a keep_graph=True capture of the SAM2 propagation replay identically the second time?  (r05: the unstreamed leg of test_dist_hip differed)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _golden as G  # noqa: E402
from oracle import seeded  # noqa: E402
from videoglamm_amd import sam2 as S2  # noqa: E402
from videoglamm_amd.params import Params  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
sd = G.weights("sam2_micro_manifest.json", 1, seeded.sam2_overrides())
for dt in (torch.float32, torch.bfloat16):
    m = S2.SAM2(Params(sd, dev, dt), "", G.sam2_cfg())
    T, N, hw = 6, 3, (40, 56)
    images = G.rnd((T, 3, m.S, m.S), 11).to(dev)
    text = G.rnd((N, 256), 12, 0.5).to(dev).to(dt)
    feats = m.hiera_frames(images)
    eager = m.video_branch(images, text, hw, frame_feats=feats)
    outs = [m.video_branch_graphed(images, text, hw, feats) for _ in range(3)]
    print(dt, "replays equal eager:", [bool(torch.equal(o, eager)) for o in outs], "max diff", [float((o - eager).abs().max()) for o in outs])
    feats2 = {t: [f.clone() for f in feats[t]] for t in feats}
    o4 = m.video_branch_graphed(images, text, hw, feats2)
    print("   fresh feature tensors:", bool(torch.equal(o4, eager)), m.video_graph_nodes())
