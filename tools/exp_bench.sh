cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t1
timeout 1200 python bench.py --no-cpu-baseline $BARGS > gpurun_out/t1/bench_f16.json 2> gpurun_out/t1/bench_f16.err; tail -5 gpurun_out/t1/bench_f16.err
python - <<PY
import json
l=json.loads(open("gpurun_out/t1/bench_f16.json").read().strip().splitlines()[-1])
print(l["value"], l["roofline"]["frac"], l["roofline"]["avg_us"])
print(json.dumps(l["roofline"]["modes"], indent=1)[:900])
print(json.dumps(l["extras"]["free_running"], indent=1)[:1200])
print({k:(v.get("hbm_frac"), v.get("f16_mode")) for k,v in l["extras"]["kernels"].items() if isinstance(v,dict) and "cfg" in k})
PY
