"""One tracked frame of the SAM2-L propagation, kernel by kernel.
  run:      rocprofv3 --kernel-trace -d DIR -o kt -- python tools/lab/video_frame_seq.py run [N]
  analyse:  python tools/lab/video_frame_seq.py show <results.db>
`run` drives an eager bf16 video_branch with the device idled between frames (sleep), so that every frame is its own busy segment in the trace;
`show` prints the launch sequence of the LAST frame (names shortened, duration in us) and the totals per kernel."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(n_obj):
    import torch
    from videoglamm_amd import sam2 as S2, synth
    from videoglamm_amd.params import Params
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    cfg = synth.SAM2_L
    sd = synth.device_state_dict(synth.sam2_manifest(cfg), dev, torch.bfloat16)
    m = S2.SAM2(Params(sd, dev, torch.bfloat16), "", cfg)
    T = 10
    images = torch.randn(T, 3, 1024, 1024, device=dev)
    text = (torch.randn(n_obj, 256, device=dev) * 0.5).to(torch.bfloat16)
    feats = m.hiera_frames(images)
    m.video_branch(images, text, (1024, 1024), frame_feats=feats, as_masks=True)      # warm: weight packing, constants
    torch.cuda.synchronize()
    # the instrumented pass: idle the device at every frame boundary (memory encoder's last launch = end of a frame)
    orig = S2.SAM2.encode_new_memory

    def paused(self, *a, **k):
        r = orig(self, *a, **k)
        torch.cuda.synchronize()
        time.sleep(0.02)
        return r
    S2.SAM2.encode_new_memory = paused
    m.video_branch(images, text, (1024, 1024), frame_feats=feats, as_masks=True)
    torch.cuda.synchronize()


def show(db_path):
    import re
    import sqlite3
    from collections import defaultdict
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, start, end from kernels order by start"))
    segs, cur, cur_end = [], [], None
    for r in rows:
        if cur and r[1] - cur_end > 5e6:
            segs.append(cur)
            cur = []
        cur.append(r)
        cur_end = r[2]
    segs.append(cur)
    seg = segs[-2] if len(segs[-1]) < 20 else segs[-1]        # (the last segment may be the final upsample alone)

    def short(n):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)
        return n[:100]
    t0, t1 = seg[0][1], seg[-1][2]
    busy = sum(r[2] - r[1] for r in seg)
    print(f"{len(segs)} segments; frame segment: {len(seg)} launches, span {(t1 - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us")
    for r in seg:
        print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f}  {short(r[0])}")
    tot = defaultdict(lambda: [0, 0.0])
    for r in seg:
        k = tot[short(r[0])]
        k[0] += 1
        k[1] += (r[2] - r[1]) / 1e3
    print("\nper kernel:")
    for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"{c:5d} {us:9.1f} us  {n}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    else:
        show(sys.argv[2])
