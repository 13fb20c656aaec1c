"""Summarise the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/probe_train.py (tools/gpu_job.sh <tag> train_traffic) into
profiles/<tag>_train_traffic.json (tag = second argument, default r04): HBM bytes per launch of the training step's large kernels."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
root = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "train_traffic")
KERNELS = ("h2wgrad_dma256_kernel<false>", "h2wgrad_dma256_kernel<true>", "h2wgrad_dma_kernel", "h2wgrad_kernel<8, 9", "h2wgrad_kernel<8, 2", "h2lin_ws_kernel", "h2lin_kernel<16>", "chain_kernel<4, 3>", "chain_ws_kernel<8>", "train_ksum_bwd_kernel", "segment_sum_rows_csr_kernel", "train_extras_dgrad_kernel",
           "chain_gather_kernel", "merge_bwd_kernel", "mlp3_kernel<18")
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
    big = [i for i, v in enumerate(fs) if v > 0.5 * max(fs)]            # the large launches of the name (the 307 k-row layers)
    f2 = [fs[i] for i in big]
    w2 = [ws[i] for i in big] if len(ws) == len(fs) else ws
    raw = sum(f2) / len(f2) * 1024                                       # the counters are in KiB
    wr = sum(w2) / len(w2) * 1024 if w2 else 0.0
    kern[k] = dict(launches=len(f2), fetch_bytes_raw=raw, fetch_bytes_corrected=2 * raw, write_bytes=wr, hbm_bytes=2 * raw + wr)
    print("%-60s n=%2d fetch x2 %9.1f MB  write %9.1f MB" % (k[:60], len(f2), 2 * raw / 1e6, wr / 1e6))
note = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `tools/probe_train.py --steps 4` (C3: 3136 rays, 307 120 row slots) on MI355X "
        "(tools/gpu_job.sh train_traffic); per-launch averages over the LARGE launches of each kernel name; FETCH_SIZE doubled per MI355X_MICROARCH.md "
        "(gfx950 tallies 128-B requests as 64 B on wide coalesced reads). The step's tensors (300 MB each) are larger than the 256 MiB Infinity Cache, "
        "but a kernel that re-reads what the previous kernel has just written can be served from it: the counters are memory-side.")
json.dump(dict(note=note, kernels=kern), open(os.path.join(ROOT, "profiles", (sys.argv[2] if len(sys.argv) > 2 else "r04") + "_train_traffic.json"), "w"), indent=1)
