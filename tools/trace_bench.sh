# rocprofv3 kernel trace + stats of the default bench (f32), summary printed
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tb
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
cd $R
cat $OUT/bench.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phases_us_per_step'])"
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/tb/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
