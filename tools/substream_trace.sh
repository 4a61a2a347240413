R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sstrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/tools/substream_trace.py > $OUT/t.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/sstrace/t/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
print(rows[0].keys())
for r in rows[-40:]:
    print(r.get("Queue_Id"), r.get("Stream_Id", ""), r["Kernel_Name"].split("(")[0][-45:], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
