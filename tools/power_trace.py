"""Power / shader-clock trace of the GPU while bench.py renders frames back to back (substantiate or retire the
"package power cap" reading of the chain kernel's clock).

    python tools/power_trace.py [--steps 300] [--out gpurun_out/power_trace.csv]

The parent never touches the GPU: it starts bench.py as a child process and samples, every ~50 ms, the hwmon files of the amdgpu
device (power1_average / power1_input in microwatts, freq1_input in Hz) and -- once a second -- `rocm-smi --showpower --showclocks`
as a cross-check.  Output: CSV (t_s, power_w, sclk_mhz, source) + a summary line on stdout.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hwmon_paths():
    out = []
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        p = next((os.path.join(hw, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw, n))), None)
        f = os.path.join(hw, "freq1_input") if os.path.exists(os.path.join(hw, "freq1_input")) else None
        if p or f:
            out.append((p, f))
    return out


def read_num(path):
    try:
        return float(open(path).read().strip())
    except Exception:
        return None


def smi_sample():
    """(power W, sclk MHz) from rocm-smi's JSON, None where absent."""
    try:
        txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(txt)
        card = d[sorted(d.keys())[0]]
        pw = next((float(v) for k, v in card.items() if "ower" in k and "(W)" in k), None)
        ck = next((v for k, v in card.items() if k.lower().startswith("sclk")), None)
        mhz = float(str(ck).strip("()").lower().replace("mhz", "")) if ck is not None else None
        return pw, mhz
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "power_trace.csv"))
    a = ap.parse_args()
    hw = hwmon_paths()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    log = open(a.out + ".bench.log", "w")
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", "3", "--no-cpu-baseline", "--no-train-leg"],
                             stdout=subprocess.PIPE, stderr=log)
    rows, t0, last_smi = [], time.time(), 0.0
    while child.poll() is None:
        t = time.time() - t0
        for ci, (p, f) in enumerate(hw):                       # every card the box exposes: the one under load is picked in the summary
            pw = read_num(p) if p else None
            fr = read_num(f) if f else None
            rows.append((t, None if pw is None else pw / 1e6, None if fr is None else fr / 1e6, "hwmon%d" % ci))
        if t - last_smi > 1.0:
            last_smi = t
            pw, mhz = smi_sample()
            rows.append((time.time() - t0, pw, mhz, "rocm-smi"))
        time.sleep(0.05)
    out = child.stdout.read().decode("utf-8", "replace")
    with open(a.out, "w") as f:
        f.write("t_s,power_w,sclk_mhz,source\n")
        for r in rows:
            f.write("%.3f,%s,%s,%s\n" % (r[0], "" if r[1] is None else "%.1f" % r[1], "" if r[2] is None else "%.0f" % r[2], r[3]))
    line = next((l for l in out.splitlines() if l.startswith("{")), None)
    ms = json.loads(line)["ms_per_step"] if line else None
    # the rendering phase = the last steps * ms_per_step seconds before the child exits (minus ~1 s of teardown)
    t_end = rows[-1][0] if rows else 0.0
    span = (a.steps * ms / 1e3) if ms else 5.0
    window = [r for r in rows if r[3].startswith("hwmon") and t_end - 1.0 - span <= r[0] <= t_end - 1.0]
    cards = sorted(set(r[3] for r in window))
    mean_p = {c: (sum(r[1] for r in window if r[3] == c and r[1] is not None) / max(1, sum(1 for r in window if r[3] == c and r[1] is not None))) for c in cards}
    card = max(mean_p, key=mean_p.get) if mean_p else None         # the GPU this process's child was given = the one drawing power
    busy = [r for r in window if r[3] == card]
    pws = [r[1] for r in busy if r[1] is not None]
    cks = [r[2] for r in busy if r[2] is not None]
    print(json.dumps(dict(ms_per_step=ms, card=card, samples=len(busy), power_w_mean=sum(pws) / len(pws) if pws else None, power_w_max=max(pws) if pws else None,
                          sclk_mhz_mean=sum(cks) / len(cks) if cks else None, sclk_mhz_min=min(cks) if cks else None, sclk_mhz_max=max(cks) if cks else None,
                          hwmon=[list(x) for x in hw], smi=[r for r in rows if r[3] == "rocm-smi"][-3:])))
    return child.returncode


if __name__ == "__main__":
    sys.exit(main())
