"""rocprofv3 kernel-trace database -> profiles/<tag>_kt_<mode>.json: per kernel class (the classes of tools/pmc_json.py) the average
launch duration, launches and kernel time PER PASS of the hot path, for ONE stream configuration:
  mode "overlapped" = bench.py's timed configuration (Hiera on its side stream next to the towers / prefill),
  mode "serial"     = VG_HIERA_START=serial VG_TOWERS_OVERLAP=0, the configuration of bench.py's instrumented pass (the one its
                      live per-launch HIP events — `roofline.frac` — are taken in).
Every pass in the traced command runs in that one configuration (bench.py --no-roofline: warm-up + timed passes only), so the
per-pass columns divide by the true number of passes.  Optional third database: a --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass of the same command -> mfma_busy_frac per class = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE) / 8 x 1024 SIMDs): the fraction
of the matrix pipes' cycles, at the clock the kernel actually ran at, that an MFMA occupied.  Calibration (r03, MI355X): rocprofv3 reports
GRBM_GUI_ACTIVE summed over the 8 XCDs (16-18 k "cycles" per microsecond of kernel time = 8 x 2.0-2.2 GHz) and the SQ counter summed over
every SIMD of the chip (the 256x256-tile GEMM: 286.0 M busy cycles per launch against 32 cycles x 0.2692 TFLOP / 32768 flop = 263 M for the
algorithmic MFMAs + 6.6 % of row padding at M = 3361).  A PMC pass serialises the dispatches, so the counter fraction is a property of the
kernel running alone (the "serial" configuration) whichever stream configuration the command asks for.
usage: python tools/kt_json.py <kt.db> <out.json> <passes> <mode> "<command>" [<mfma_pmc.db>]"""
import hashlib
import json
import os
import re
import sqlite3
import sys

from pmc_json import CLASSES

SIMDS_PER_XCD_SUM = 1024 / 8.0     # GRBM_GUI_ACTIVE arrives summed over the 8 XCDs
INIT_KERNELS = r"distribution_elementwise_grid_stride_kernel|AUnaryFunctor<float, float, float, at::native::binary_internal::MulFunctor"


def main():
    dbp, out, passes, mode, command = sys.argv[1:6]
    passes = float(passes)
    db = sqlite3.connect(dbp)
    rows = list(db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels group by name"))
    pm = {}
    if len(sys.argv) > 6:
        pdb = sqlite3.connect(sys.argv[6])
        q = ("select k.name, p.counter_name, count(*), sum(p.counter_value) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
             "group by k.name, p.counter_name")
        for name, ctr, n, tot in pdb.execute(q):
            pm.setdefault(name, {})[ctr] = (n, tot)
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "videoglamm_amd", "csrc", "libvgkernels.so")
    lib_sha = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16] if os.path.exists(so) else None
    # (model-build kernels — the synthetic weights' RNG and scaling — run once per process, not per pass: kept out of the per-pass total)
    rows = [r for r in rows if not re.search(INIT_KERNELS, r[0])]
    res = {"round": 6, "lib_sha16": lib_sha, "mode": mode, "command": command, "passes": passes,
           "total_kernel_ms_per_pass": round(sum(r[2] for r in rows) / 1e3 / passes, 2), "kernels": {}}
    for key, pat in CLASSES:
        sel = [r for r in rows if re.search(pat, r[0])]
        if not sel:
            continue
        n = sum(r[1] for r in sel)
        us = sum(r[2] for r in sel)
        ent = {"avg_us": round(us / n, 2), "launches_per_pass": round(n / passes, 1), "ms_per_pass": round(us / 1e3 / passes, 2),
               "kernel_names": sorted(r[0][:120] for r in sel)}
        busy = sum(pm.get(r[0], {}).get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0.0))[1] for r in sel)
        act = sum(pm.get(r[0], {}).get("GRBM_GUI_ACTIVE", (0, 0.0))[1] for r in sel)
        if act > 0:
            ent.update(mfma_busy_cycles_sum=busy, grbm_gui_active_sum=act, mfma_busy_frac=round(busy / (act * SIMDS_PER_XCD_SUM), 4))
        res["kernels"][key] = ent
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    for k, v in res["kernels"].items():
        print(f"{k:18s} avg {v['avg_us']:9.1f} us  {v['launches_per_pass']:7.1f} launches/pass  {v['ms_per_pass']:7.2f} ms/pass  mfma_busy {v.get('mfma_busy_frac')}")


if __name__ == "__main__":
    main()
