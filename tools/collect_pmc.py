"""Summarise SQ / GRBM counter passes of the bench frame (tools/gpu_job.sh: pmc_g1.csv, pmc_g2.csv, pmc_g3.csv = rocprofv3 --pmc
counter_collection CSVs of separate passes) into profiles/<tag>_chain_pmc.json: per kernel the largest launch of every counter, MFMA-busy fraction
= SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs.
python tools/collect_pmc.py gpurun_out/<tag> [rNN]"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
root = sys.argv[1]; tag = sys.argv[2] if len(sys.argv) > 2 else "r04"
KERN = ("march_kernel", "knn3_kernel", "knn_nb_kernel", "knn_quad_kernel", "chain_gather_kernel", "chain_ws_kernel", "chain_sigma_kernel", "mlp3_kernel", "merge_wp_kernel", "mixfinal_wp_kernel", "composite_kernel", "cf_ws_kernel")
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pmc_g*.csv"))):
    if f.endswith("_trace.csv"): continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].strip()
        if any(s in k for s in KERN):
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in per.items():
    c = {n: max(v) for n, v in cs.items()}
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    e = dict(counters=c, launches=max(len(v) for v in cs.values()), kernel_cycles=cyc)
    if cyc > 0:
        e["mfma_busy_fraction"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc), 4)
    if c.get("SQ_INSTS_MFMA"): e["valu_per_mfma"] = round(c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"], 3)
    if c.get("SQ_WAVE_CYCLES"):
        e["wait_any_frac"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
        e["wait_inst_any_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
    out["chain_ws_kernel" if "chain_ws_kernel" in k else k] = e
    print("%-50s cycles %.0f mfma_busy %s valu/mfma %s" % (k[:50], cyc, e.get("mfma_busy_fraction"), e.get("valu_per_mfma")))
note = ("rocprofv3 --kernel-trace --pmc, separate passes of `bench.py --no-cpu-baseline --no-train-leg --no-f32-anchor --steps 2 --warmup 1` (tools/gpu_job.sh); "
        "per kernel the LARGEST launch of each counter (the frame launches). SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (= 32 x SQ_INSTS_MFMA for "
        "v_mfma_f32_32x32x16_f16); GRBM_GUI_ACTIVE is summed over the 8 XCDs; mfma_busy_fraction = busy cycles per SIMD / kernel cycles.")
import hashlib
sha = {}
for f in ("chain_ws.hip", "chain_defs.h", "query.hip", "mlp.hip"):
    sha[f] = hashlib.sha256(open(os.path.join(ROOT, "hybridneuralrendering_amd", "csrc", f), "rb").read()).hexdigest()
json.dump(dict(note=note, source_sha256=sha, kernels=out, **{k: v for k, v in out.items() if k == "chain_ws_kernel"}), open(os.path.join(ROOT, "profiles", tag + "_chain_pmc.json"), "w"), indent=1)
