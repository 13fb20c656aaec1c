#!/bin/bash
# Every fp32 division of the device code must go through hnr_div (csrc/hnr_common.h): compiles each source with hnr_div's `/` fallback removed and
# lists the v_div_fmas_f32 (the hardware division sequence that reads a lane mask from VCC) that remain, per kernel.  Exit 1 if any.
cd "$(dirname "$0")/../hybridneuralrendering_amd/csrc"
bad=0
for f in *.hip; do
  /opt/rocm/bin/hipcc -DHNR_DIV_NO_FALLBACK --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I../../include -S --cuda-device-only -o /tmp/chkdiv.s $f 2>/dev/null
  python3 - "$f" <<'PY' || bad=1
import sys, re, collections
c = collections.Counter(); cur = None
for l in open('/tmp/chkdiv.s'):
    m = re.match(r'^(_Z\w+):', l)
    if m: cur = m.group(1)
    if 'v_div_fmas_f32' in l or 'v_div_fmas_f64' in l: c[cur] += 1
for k, v in c.items(): print("%s: %d division(s) in %s" % (sys.argv[1], v, k))
sys.exit(1 if c else 0)
PY
done
[ $bad = 0 ] && echo "no hardware division sequence outside hnr_div's fallback"
exit $bad
