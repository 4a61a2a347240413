# A/B of the cache-policy variants built by tools/build_variants.sh (s = tile store nt, l = tile load nt, c = candidate store nt)
for v in s0l0c0 s1l0c0 s0l1c0 s1l1c0; do
  for dt in f32 f64; do
    ILQR_LIB=$PWD/tools/variants/libilqr_$v.so python bench.py --dtype $dt --no-cpu-baseline > gpurun_out/ab_${v}_$dt.json 2> gpurun_out/ab_${v}_$dt.err
  done
done
python - <<'PY'
import json,glob
for v in "s0l0c0 s1l0c0 s0l1c0 s1l1c0".split():
    for dt in ("f32","f64"):
        try:
            d=json.load(open(f"gpurun_out/ab_{v}_{dt}.json")); p=d["phases_us_per_step"]
            print(v, dt, "%.3f ms/step"%d["ms_per_step"], "lin %.1f bwd %.1f (b2b %.1f) fwd %.1f"%(p["linearize"],p["backward"],d["roofline"]["avg_launch_us_back_to_back"],p["forward"]))
        except Exception as e: print(v, dt, "ERR", e)
PY
