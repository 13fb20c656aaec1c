"""Instruction mix between consecutive MFMAs of chain_ws_kernel<0> (from the -save-temps assembly).
python tools/isa_slots.py [asm file] [kernel symbol substring]
Prints, per MFMA slot of the tile loop, the number of VALU / LDS / VMEM / SALU / wait instructions that follow it, and a histogram."""
import re, sys, collections

args = [a for a in sys.argv[1:] if not a.startswith("-") and not a.isdigit()]
path = args[0] if len(args) > 0 else "hybridneuralrendering_amd/csrc/build/chain_ws-hip-amdgcn-amd-amdhsa-gfx950.s"
sym = args[1] if len(args) > 1 else "chain_ws_kernelILi0"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and sym in l and l.rstrip().split(":")[0].endswith("E") and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]

def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu"
    return "other"

slots = []            # per MFMA: Counter of what follows until the next MFMA
cur = None
labels = []
for l in body:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if t.startswith(".LBB"): labels.append((len(slots), t))
        continue
    if t.endswith(":"):
        continue
    op = t.split()[0]
    k = kind(op)
    if k == "mfma":
        cur = collections.Counter(); cur["ops"] = []
        slots.append(cur)
        continue
    if cur is not None:
        cur[k] += 1
        cur["ops"].append(t.split(";")[0].strip())

print("MFMAs in kernel body: %d" % len(slots))
tot = collections.Counter()
for s in slots:
    for k in ("valu", "lds", "vmem", "salu", "wait", "barrier", "nop"):
        tot[k] += s[k]
print("totals after MFMAs:", dict(tot))
hist = collections.Counter(s["valu"] for s in slots)
print("VALU-per-slot histogram:", sorted(hist.items()))
if "-v" in sys.argv:
    for i, s in enumerate(slots):
        print("%4d: valu %2d lds %d vmem %d salu %d wait %d nop %d %s" % (i, s["valu"], s["lds"], s["vmem"], s["salu"], s["wait"], s["nop"], "BAR" if s["barrier"] else ""))
if "-vv" in sys.argv:
    a, b = int(sys.argv[sys.argv.index("-vv") + 1]), int(sys.argv[sys.argv.index("-vv") + 2])
    for i in range(a, b):
        print("---- slot %d" % i)
        for o in slots[i]["ops"]: print("    " + o)
