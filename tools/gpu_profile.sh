# rocprofv3 evidence for profiles/: kernel trace + stats of the default bench, then PMC passes
# (separate runs, --kernel-trace only, as the pool requires) for HBM traffic of the backward sweep.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 $R/bench.py --dtype f64 --no-cpu-baseline > $OUT/bench_f64.json 2> $OUT/bench_f64.err
for dt in f32 f64; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$dt -- python3 $R/bench.py --no-cpu-baseline --no-solve-extra --dtype $dt > $OUT/trace_$dt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$dt -- python3 $R/bench.py --no-cpu-baseline --no-solve-extra --steps 5 --dtype $dt > $OUT/pmc_fetch_$dt.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$dt -- python3 $R/bench.py --no-cpu-baseline --no-solve-extra --steps 5 --dtype $dt > $OUT/pmc_write_$dt.log 2>&1
done
cd $R
cat $OUT/bench_default.json; cat $OUT/bench_f64.json
