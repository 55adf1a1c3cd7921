cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(time python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r04_c_tests.log 2>&1
for i in 1 2; do
  python bench.py --no-secondary --no-cpu-baseline --no-decode-tail > gpurun_out/r04_c_bench_base$i.json 2>/dev/null
  ECAD_HIP_LIB=build_tmp/libecad_stag.so python bench.py --no-secondary --no-cpu-baseline --no-decode-tail > gpurun_out/r04_c_bench_stag$i.json 2>/dev/null
done
tail -6 gpurun_out/r04_c_tests.log
python - <<'PY'
import json
for n in ("base1","stag1","base2","stag2"):
    d=json.load(open(f"gpurun_out/r04_c_bench_{n}.json"))
    t=d["time_breakdown_ms_per_batch"]
    print(n, round(d["value"],1), round(d["config"]["one_batch_at_a_time_images_per_s"],1), "ff1", t["gemm_ff1_gelu"], "kv", t["gemm_attn1_kv_wide"], "out1", t["gemm_attn1_out"], "ff2", t["gemm_ff2"], "sum", t["_sum"])
PY
