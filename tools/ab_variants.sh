# A/B of library variants built by tools/build_variants.sh: tools/ab_variants.sh name1 name2 ...  (f32 and f64 bench each)
for v in "$@"; do
  for dt in f32 f64; do
    ILQR_LIB=$PWD/tools/variants/libilqr_$v.so python bench.py --dtype $dt --no-cpu-baseline > gpurun_out/ab_${v}_$dt.json 2> gpurun_out/ab_${v}_$dt.err
  done
done
python - "$@" <<'PY'
import json, sys
for v in sys.argv[1:]:
    for dt in ("f32", "f64"):
        try:
            d = json.load(open(f"gpurun_out/ab_{v}_{dt}.json")); p = d["phases_us_per_step"]
            print(v, dt, "%.3f ms/step" % d["ms_per_step"], "lin %.1f bwd %.1f (b2b %.1f) fwd %.1f sel %.1f" % (p["linearize"], p["backward"], d["roofline"]["avg_launch_us_back_to_back"], p["forward"], p["select"]))
        except Exception as e:
            print(v, dt, "ERR", e)
PY
