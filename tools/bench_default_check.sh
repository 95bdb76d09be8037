T0=$(date +%s); python bench.py > gpurun_out/r6z_bench_default.json 2> gpurun_out/r6z_bench_default.err; echo rc=$? secs=$(( $(date +%s) - T0 )); wc -l gpurun_out/r6z_bench_default.json
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r6z_bench_default.json").read().strip().splitlines()[-1])
print({k:j[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","dtype","scaling","vs_baseline")})
print(j["roofline"]["frac"], j["roofline"]["traffic"], j["roofline"]["traffic_stale"], j["cpu_baseline"]["value"], j["self_check"])
PY
tail -3 gpurun_out/r6z_bench_default.err
