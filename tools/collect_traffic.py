"""Summarise the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py (tools/gpu_job.sh <tag> traffic) into profiles/<tag>_traffic.json (tag = second argument, default r04).

    bash tools/gpu_job.sh <tag> traffic      # on the GPU box: writes gpurun_out/<tag>/{pmc_FETCH_SIZE,pmc_WRITE_SIZE,stats}
    python tools/collect_traffic.py gpurun_out/traffic5
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
root = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "traffic5")
KERNELS = ("chain_ws_kernel", "chain_kernel", "chain_gather_kernel", "chain_sigma_kernel", "mlp3_kernel", "knn3_kernel", "knn_quad_kernel", "knn_nb_kernel", "knn_set_kernel", "march_kernel", "proj_rows_kernel", "merge_kernel", "final_color_kernel",
           "composite_kernel", "linear_s3w_kernel", "linear_f32_kernel<2, 2, 1, 0, 4", "gather_rows", "ksum_kernel")
out = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("%s/pmc_%s/*/*counter_collection.csv" % (root, name))[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            per[r["Kernel_Name"].split("(")[0].strip()].append(float(r["Counter_Value"]))
    out[name] = per
kern = {}
for k, fs in out["FETCH_SIZE"].items():
    if not any(s in k for s in KERNELS):
        continue
    ws = out["WRITE_SIZE"].get(k, [])
    big = [i for i, v in enumerate(fs) if v > 0.5 * max(fs)]            # the large launches of the name (per-neighbour layers, full frames)
    f2 = [fs[i] for i in big]
    w2 = [ws[i] for i in big] if len(ws) == len(fs) else ws
    raw = sum(f2) / len(f2) * 1024                                       # the counters are in KiB
    wr = sum(w2) / len(w2) * 1024 if w2 else 0.0
    kern[k] = dict(launches=len(f2), fetch_bytes_raw=raw, fetch_bytes_corrected=2 * raw, write_bytes=wr, hbm_bytes=2 * raw + wr)
    print("%-55s n=%2d fetch x2 %9.1f MB  write %9.1f MB" % (k[:55], len(f2), 2 * raw / 1e6, wr / 1e6))
note = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-leg` on MI355X "
        "(tools/gpu_job.sh <tag> traffic); per-launch averages over the LARGE launches of each kernel name; FETCH_SIZE doubled per MI355X_MICROARCH.md "
        "(gfx950 tallies 128-B requests as 64 B on wide coalesced reads; calibrated in round 1 on ksum_kernel: 12.4 GB raw vs 24.3 GB of rows actually read). "
        "chain_ws_kernel<0> = the fused per-neighbour chain (one launch per frame); its weight image (848 KiB) is re-read by every workgroup tile from L2, "
        "which these memory-side counters do not see.")
json.dump(dict(note=note, kernels=kern), open(os.path.join(ROOT, "profiles", (sys.argv[2] if len(sys.argv) > 2 else "r05") + "_traffic.json"), "w"), indent=1)
