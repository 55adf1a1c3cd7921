#!/bin/bash
# After `gpurun -- 'bash scripts/collect_profiles.sh TAG [workload] [--fp8]'`: copy every summary kind of TAG from gpurun_out/ (what gpurun merges back) into profiles/
# (what is committed and judged).  The PMC summaries carry the kernel-source stamp bench.py matches: without them `roofline.traffic` and the SQ fields are null.
#   bash scripts/copy_profiles.sh r05_e
set -u
TAG=${1:?tag}
cd "$(dirname "$0")/.."
n=0
for w in "" _fp8 _sigma1024 _flux1024 _flux1024_fp8; do
  for k in bench.json bench_under_rocprof.json kernel_stats.csv pmc_traffic.json pmc_sq.json; do
    f=gpurun_out/${TAG}${w}_$k
    [ -f "$f" ] && cp "$f" profiles/ && n=$((n + 1))
  done
done
echo "copied $n files of $TAG into profiles/"
