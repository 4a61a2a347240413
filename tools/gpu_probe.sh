# diagnostic: cycles per step and held clock of the two sequential kernels (clock probe build flag via env)
for dt in f32 f64; do
  ILQR_CLOCK_PROBE=1 python bench.py --dtype $dt --no-cpu-baseline > gpurun_out/p_$dt.json 2> gpurun_out/p_$dt.err
  ILQR_CLOCK_PROBE=1 ILQR_BACKWARD_REG_RING=1 python bench.py --dtype $dt --no-cpu-baseline > gpurun_out/pr_$dt.json 2> gpurun_out/pr_$dt.err
done
python - <<'PY'
import json
for f in ["p_f32","pr_f32","p_f64","pr_f64"]:
    try:
        d=json.load(open(f"gpurun_out/{f}.json")); print(f, "%.3f ms"%d["ms_per_step"], {k:round(v,1) for k,v in d["phases_us_per_step"].items()}, d.get("clock_probe"))
    except Exception as e: print(f, "ERR", e); print(open(f"gpurun_out/{f}.err").read()[-1500:])
PY
