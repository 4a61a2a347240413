# rocprofv3 evidence for profiles/: kernel trace + stats of the default bench, then PMC passes
# (separate runs, --kernel-trace only, as the pool requires) for HBM traffic of the backward sweep.
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 5 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --no-cpu-baseline --steps 5 > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace32 -- python3 $R/bench.py --no-cpu-baseline --dtype f32 > $OUT/trace32.log 2>&1
cd $R
find gpurun_out/prof -name "*.csv" | head -30
cat $OUT/bench_default.json
