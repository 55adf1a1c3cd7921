#!/bin/bash
# Samples board power and clocks with rocm-smi while the headline workload runs (evidence for DESIGN 5h: the K loops sit on the power cap).
# usage (GPU box): bash scripts/power_during_bench.sh [extra bench.py flags]  ->  gpurun_out/power_trace.log + summary on stdout
mkdir -p gpurun_out
( for i in $(seq 1 400); do /opt/rocm/bin/rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ';'; echo; sleep 0.25; done ) > gpurun_out/power_trace.log 2>&1 &
SMI=$!
python bench.py --steps 60 --warmup 3 --no-secondary --no-cpu-baseline --no-decode-tail --no-roofline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('images/s', d['value'], d['dtype'])"
kill $SMI 2>/dev/null
python - <<'PY'
import re
pw, sclk = [], []
cap = None
for ln in open("gpurun_out/power_trace.log"):
    m = re.search(r"Average Graphics Package Power \(W\): ([\d.]+)|Current Socket Graphics Package Power \(W\): ([\d.]+)", ln)
    if m: pw.append(float(m.group(1) or m.group(2)))
    m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", ln)
    if m: sclk.append(int(m.group(1)))
    m = re.search(r"Max Graphics Package Power \(W\): ([\d.]+)", ln)
    if m: cap = float(m.group(1))
busy = [p for p in pw if p > 0.6 * max(pw)] if pw else []
print(f"samples {len(pw)}; power cap {cap} W; power while busy: mean {sum(busy)/max(1,len(busy)):.0f} W, max {max(pw) if pw else 0:.0f} W; sclk while sampled: min {min(sclk) if sclk else 0} / median {sorted(sclk)[len(sclk)//2] if sclk else 0} / max {max(sclk) if sclk else 0} MHz")
PY
head -3 gpurun_out/power_trace.log | cut -c1-400
