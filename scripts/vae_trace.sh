#!/bin/bash
# kernel trace of scripts/time_vae.py on the GPU box: bash scripts/vae_trace.sh <tag>  ->  gpurun_out/<tag>_vae_kernel_stats.csv, gpurun_out/<tag>_vae_kernel_trace.csv
set -u
TAG=${1:-vae}
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 scripts/time_vae.py 16 8 > gpurun_out/${TAG}_vae_time.log 2>&1
cp $(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_vae_kernel_stats.csv
cp $(find gpurun_out/${TAG}_prof -name "*kernel_trace.csv" | head -1) gpurun_out/${TAG}_vae_kernel_trace.csv
rm -rf gpurun_out/${TAG}_prof
# HBM traffic by kernel (two counter passes, no trace domains beside them): gpurun_out/<tag>_vae_pmc_traffic.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_vpmc_$c -- python3 scripts/time_vae.py 16 1 > /dev/null 2> gpurun_out/${TAG}_vpmc_$c.log
done
python3 scripts/pmc_vae.py gpurun_out/${TAG}_vpmc_FETCH_SIZE gpurun_out/${TAG}_vpmc_WRITE_SIZE gpurun_out/${TAG}_vae_pmc_traffic.json 4 | tail -14
rm -rf gpurun_out/${TAG}_vpmc_FETCH_SIZE gpurun_out/${TAG}_vpmc_WRITE_SIZE
