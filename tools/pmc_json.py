"""rocprofv3 PMC databases (one pass with FETCH_SIZE, one with WRITE_SIZE) -> profiles/<tag>_pmc_<kernel>.json, the per-launch
HBM traffic bench.py quotes as `traffic`.  usage: python tools/pmc_json.py <fetch.db> <write.db> <outdir> <tag> "<workload>" "<command>"
traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes): on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at
64 B (/opt/skills/guides/MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as reported."""
import json
import os
import re
import sqlite3
import sys

CLASSES = [
    ("gemm_glds", r"gemm_tile_glds_kernel<unsigned short"),
    ("gemm_k64b", r"gemm_tile_k64b_kernel<unsigned short|gemm_tile_ring64_kernel<unsigned short"),
    ("gemm_w128", r"gemm_tile_p8_kernel<unsigned short"),      # the 256x256 route (label kept from the kernel it replaced in r04)
    ("gemm_p8n", r"gemm_tile_p8n_kernel<unsigned short"),      # r05: the 256x192 route
    ("gemm_rr", r"gemm_rr_kernel<"),                             # r05: the row-register short-K route
    ("gemm_s128", r"gemm_tile_s128_kernel<unsigned short"),
    ("mlp_rows", r"mlp_rows_kernel"),
    ("decode_gemv_glu", r"decode_gemv_fast_kernel<unsigned short, unsigned short, true"),
    ("attn_d64", r"attn_kernel<unsigned short, 64,|attn_dma_kernel<64,"),          # (r06: the long sequences run on the LDS-DMA-staged kernel)
    ("attn_d96", r"attn_kernel<unsigned short, 96,|attn_dma_kernel<96,"),
    ("attn_d128", r"attn_kernel<unsigned short, 128,|attn_dma_kernel<128,"),
    ("attn_d256", r"attn_kernel<unsigned short, 256,"),
    ("attn_window", r"win256_attn_kernel|tiny_win_attn_kernel"),
    ("norm_short", r"norm_short_kernel"),
]


def per_kernel(dbpath, counter):
    db = sqlite3.connect(dbpath)
    q = ("select k.name, count(*), avg(p.counter_value) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
         "where p.counter_name = ? group by k.name")
    return {r[0]: (r[1], r[2]) for r in db.execute(q, (counter,))}


def main():
    fdb, wdb, outdir, tag, workload, command = sys.argv[1:7]
    fetch, write = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    for key, pat in CLASSES:
        names = [n for n in fetch if re.search(pat, n)]
        if not names:
            continue
        nf = sum(fetch[n][0] for n in names)
        f = sum(fetch[n][0] * fetch[n][1] for n in names) / nf
        nw = sum(write[n][0] for n in names if n in write)
        w = sum(write[n][0] * write[n][1] for n in names if n in write) / max(nw, 1)
        out = {"round": 6, "workload": workload, "command": command, "fetch_correction": 2.0, "kernel": key, "kernel_names": sorted(n[:120] for n in names),
               "dispatches": nf, "FETCH_SIZE_KB_avg_per_dispatch": round(f, 1), "WRITE_SIZE_KB_avg_per_dispatch": round(w, 1),
               "traffic_bytes_per_launch": round((2.0 * f + w) * 1000.0)}
        with open(os.path.join(outdir, f"{tag}_pmc_{key}.json"), "w") as fh:
            json.dump(out, fh, indent=1)
        print(key, out["dispatches"], out["traffic_bytes_per_launch"])


if __name__ == "__main__":
    main()
