cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4bf
SECONDS=0; timeout 1500 python bench.py > gpurun_out/r4bf/line.json 2> gpurun_out/r4bf/err.txt; echo rc=$? wall=${SECONDS}s
grep "bench \|Elapsed (wall" gpurun_out/r4bf/err.txt | tail -14
python - <<'PY'
import json
l=json.load(open('gpurun_out/r4bf/line.json')); e=l.get('extras') or {}
print(l['value'], l['roofline']['avg_us'], l['roofline']['frac'], l['roofline']['traffic'], sorted(e.keys())[:12], e.get('error'))
print(l['config']['workload'][-120:])
PY
