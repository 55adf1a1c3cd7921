#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root): rocprofv3 kernel trace of the bench, then the two PMC
# passes (FETCH_SIZE and WRITE_SIZE cannot share a pass; counter runs carry no trace domains), then the stamped traffic summary.
#   gpurun -- 'bash scripts/collect_profiles.sh r03_a'                      # headline: PixArt-alpha 256, batch 16
#   gpurun -- 'bash scripts/collect_profiles.sh r03_a sigma1024'            # BASELINE config 3
#   gpurun -- 'bash scripts/collect_profiles.sh r03_a flux1024 [--fp8]'     # BASELINE config 5 (bf16, or the fp8 mode it names)
# Outputs under gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards (bash scripts/copy_profiles.sh <tag>: the box's own profiles/ is not merged back).  The program sits directly after `--` (never env / bash -c).
set -u
TAG=${1:-r03_x}
WL=${2:-alpha256}
FP8=${3:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
case "$WL" in
  alpha256) PFX=${TAG}; STAMP=alpha256_b16; DT=f16; WARGS="" ;;
  sigma1024) PFX=${TAG}_sigma1024; STAMP=sigma1024_b4; DT=f16; WARGS="--workload sigma1024 --steps 3 --warmup 1" ;;
  flux1024) PFX=${TAG}_flux1024; STAMP=flux1024_b2; DT=bf16; WARGS="--workload flux1024 --steps 2 --warmup 1" ;;
  *) echo "unknown workload $WL"; exit 2 ;;
esac
if [ -n "$FP8" ]; then PFX=${PFX}_fp8; DT=fp8; WARGS="$WARGS --fp8"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${PFX}_prof -- python3 bench.py $WARGS --streams 1 --no-cpu-baseline --no-secondary --no-decode-tail > gpurun_out/${PFX}_bench_under_rocprof.json 2> gpurun_out/${PFX}_prof.log
cp $(find gpurun_out/${PFX}_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${PFX}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${PFX}_pmc_$c -- python3 bench.py $WARGS --streams 1 --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-secondary --no-decode-tail > /dev/null 2> gpurun_out/${PFX}_pmc_$c.log
done
python3 scripts/pmc_traffic.py gpurun_out/${PFX}_pmc_FETCH_SIZE gpurun_out/${PFX}_pmc_WRITE_SIZE gpurun_out/${PFX}_pmc_traffic.json $STAMP $DT | tail -20
# SQ pass (matrix-pipe busy cycles, wave states) and GRBM pass (busy cycles -> effective clock) over the same product library: scripts/pmc_sq.py
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/${PFX}_pmc_SQ -- python3 bench.py $WARGS --streams 1 --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-secondary --no-decode-tail > /dev/null 2> gpurun_out/${PFX}_pmc_SQ.log
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${PFX}_pmc_GRBM -- python3 bench.py $WARGS --streams 1 --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-secondary --no-decode-tail > /dev/null 2> gpurun_out/${PFX}_pmc_GRBM.log
python3 scripts/pmc_sq.py gpurun_out/${PFX}_pmc_SQ gpurun_out/${PFX}_pmc_GRBM gpurun_out/${PFX}_kernel_stats.csv gpurun_out/${PFX}_pmc_sq.json $STAMP $DT | tail -24
cp gpurun_out/${PFX}_pmc_sq.json profiles/${PFX}_pmc_sq.json
rm -rf gpurun_out/${PFX}_prof gpurun_out/${PFX}_pmc_FETCH_SIZE gpurun_out/${PFX}_pmc_WRITE_SIZE gpurun_out/${PFX}_pmc_SQ gpurun_out/${PFX}_pmc_GRBM
cp gpurun_out/${PFX}_pmc_traffic.json profiles/${PFX}_pmc_traffic.json   # so that the bench run below finds the stamped profile
if [ "$WL" = alpha256 ] && [ -n "$FP8" ]; then
  python3 bench.py --fp8 --streams 3 --steps 12 --warmup 2 --no-cpu-baseline --no-secondary --no-decode-tail > gpurun_out/${PFX}_bench.json 2> gpurun_out/${PFX}_bench.err
elif [ "$WL" = alpha256 ]; then
  python3 bench.py > gpurun_out/${PFX}_bench.json 2> gpurun_out/${PFX}_bench.err
else
  python3 bench.py $WARGS --no-cpu-baseline --no-secondary --no-decode-tail > gpurun_out/${PFX}_bench.json 2> gpurun_out/${PFX}_bench.err
fi
tail -3 gpurun_out/${PFX}_bench.err
