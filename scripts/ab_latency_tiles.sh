set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "query_projection" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -k "latency_tiles or fused" 2>&1 | tail -5
for b in 1 2 4; do for l in 1 0; do
  echo "== batch $b latency_tiles $l"
  ECAD_LATENCY_TILES=$l python bench.py --batch $b --streams 1 --steps 10 --warmup 3 --no-secondary --no-decode-tail 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
