python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/t.log 2>&1; tail -5 gpurun_out/t.log
python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/b32.json 2> gpurun_out/b32.err
python bench.py --dtype f64 --no-cpu-baseline > gpurun_out/b64.json 2> gpurun_out/b64.err
ILQR_BACKWARD_NO_PIN=1 python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/b32r.json 2> gpurun_out/b32r.err
python - <<'PY'
import json
for f in ["gpurun_out/b32.json","gpurun_out/b64.json","gpurun_out/b32r.json"]:
    try:
        d=json.load(open(f)); print(f, d["dtype"], "%.3g it/s"%d["value"], "%.3f ms"%d["ms_per_step"], "bw %.0f GB/s %.3f"%(d["roofline"]["achieved"], d["roofline"]["frac"]), {k:round(v,1) for k,v in d["phases_us_per_step"].items()})
    except Exception as e: print(f, "ERR", e); print(open(f.replace(".json",".err")).read()[-1500:])
PY
