"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into profiles/r01_traffic.json.
Run on the GPU box after the two PMC passes (see profiles/README.md)."""
import collections, csv, glob, json, sys

out = {}
root = sys.argv[1]
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("%s/pmc_%s/*/*counter_collection.csv" % (root, name))[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    out[name] = {k: v for k, v in per.items()}
res = {}
for k in out["FETCH_SIZE"]:
    if not any(s in k for s in ("linear_f32_kernel<2, 2, 1", "knn2_kernel", "march_kernel", "gather_rows", "ksum", "proj_rows")):
        continue
    fs, ws = out["FETCH_SIZE"][k], out["WRITE_SIZE"].get(k, [])
    if "linear_f32_kernel<2, 2, 1" in k:
        # the 4 per-neighbour launches of a frame are the big ones (M = 23.8 M rows); 3 colour-feature launches are ~7x smaller
        thr = 0.5 * max(fs)
        big = [i for i, v in enumerate(fs) if v > thr]
        fs = [fs[i] for i in big]
        ws = [ws[i] for i in big] if len(ws) >= len(out["FETCH_SIZE"][k]) else ws
    res[k] = dict(launches=len(fs), fetch_kib_avg=sum(fs) / len(fs), write_kib_avg=(sum(ws) / len(ws)) if ws else None)
json.dump(res, open("%s/traffic_summary.json" % root, "w"), indent=1)
print(json.dumps(res, indent=1))
