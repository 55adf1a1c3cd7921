#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root): rocprofv3 kernel trace of the default bench, then the two PMC
# passes (FETCH_SIZE and WRITE_SIZE cannot share a pass; counter runs carry no trace domains), then the stamped traffic summary.
#   gpurun -- 'bash scripts/collect_profiles.sh r02_c'
# Outputs under gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r02_x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 bench.py --no-cpu-baseline --no-secondary --no-decode-tail > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_prof.log
cp $(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_$c -- python3 bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-secondary --no-decode-tail > /dev/null 2> gpurun_out/${TAG}_pmc_$c.log
done
python3 scripts/pmc_traffic.py gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE gpurun_out/${TAG}_pmc_traffic.json alpha256_b16 f16 | tail -20
rm -rf gpurun_out/${TAG}_prof gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE
mkdir -p profiles_tmp && cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json   # so that the bench run below finds the stamped profile
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -3 gpurun_out/${TAG}_bench.err
rmdir profiles_tmp 2>/dev/null
