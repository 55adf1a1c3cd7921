# A/B of the few-rows switch: scripts/ab_latency_tiles.sh  (prints images/s and ms per batch)
cd "$(dirname "$0")/.."
run() { python bench.py --batch $1 --streams 1 --steps 10 --warmup 3 --no-secondary --no-decode-tail 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],2))"; }
for b in 6 8 12 16; do
  for f in 50 100 200 400; do echo -n "batch $b fill $f: "; ECAD_LATENCY_FILL=$f run $b; done
  echo -n "batch $b off: "; ECAD_LATENCY_TILES=0 run $b
done
