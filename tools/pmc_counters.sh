# SQ-level counters of a target script's kernels: tools/pmc_counters.sh <target.py> <out-subdir> [kernel-name filter ...]
# (separate --pmc passes, kernel-trace only, as the pool requires; the program itself follows `--`)
R=$GRAFT_REPO_ROOT
T=$1; OUT=$R/gpurun_out/$2; shift 2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/$T > $OUT/g$i.log 2>&1 || { echo "group $i failed"; tail -3 $OUT/g$i.log; }
done
cd $R
python3 - "$OUT" "$@" <<'PY'
import csv, glob, collections, sys
out, filt = sys.argv[1], sys.argv[2:] or ["forward", "backward", "linearize", "select"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for k, d in sorted(agg.items()):
    if not any(s in k for s in filt): continue
    lines.append(k)
    for c, v in sorted(d.items()):
        lines.append(f"   {c:30s} n={len(v):3d} mean={sum(v)/len(v):14.1f} last={v[-1]:14.1f}")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
