# rocprofv3 evidence for profiles/rNN (copied there by tools/collect_profiles.py rNN): bench lines, kernel trace + stats,
# FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace only, as the pool requires; the program itself follows
# `--`) for the default bench in both precisions, for the c5 shard and for the c4 MPC shard, SQ counters of the c3 and c5
# kernels, and the small measurements DESIGN.md quotes (achievable HBM rate, lone-wave issue costs, fp32 error).
# Every GPU step is joined with `|| exit 1`: after a step fails or times out no further one starts.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
step() { echo "== $1"; }
step bench
timeout -k 10 500 python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
timeout -k 10 300 python3 $R/bench.py --dtype f64 --no-cpu-baseline > $OUT/bench_f64.json 2> $OUT/bench_f64.err || exit 1
for dt in f32 f64; do
step "c3 $dt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$dt -- python3 $R/bench.py --no-cpu-baseline --no-solve-extra --dtype $dt > $OUT/trace_$dt.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$dt -- python3 $R/bench.py --no-cpu-baseline --no-solve-extra --steps 5 --dtype $dt > $OUT/pmc_fetch_$dt.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$dt -- python3 $R/bench.py --no-cpu-baseline --no-solve-extra --steps 5 --dtype $dt > $OUT/pmc_write_$dt.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $OUT/pmc_valu_$dt -- python3 $R/bench.py --no-cpu-baseline --no-solve-extra --steps 5 --dtype $dt > $OUT/pmc_valu_$dt.log 2>&1 || exit 1
done
# (bench.py runs the fused iteration in its timed region and the materialised one -- linearize_kernel, backward_tile16_kernel,
# select_kernel -- beside it in the same process, so the passes above hold the kernels of both paths)
for w in c5 mpc; do
step "$w f32"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 $R/tools/pmc_target_$w.py > $OUT/trace_$w.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$w -- python3 $R/tools/pmc_target_$w.py > $OUT/pmc_fetch_$w.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$w -- python3 $R/tools/pmc_target_$w.py > $OUT/pmc_write_$w.log 2>&1 || exit 1
done
cd $R
step "SQ counters"
timeout -k 10 600 bash tools/pmc_counters.sh tools/pmc_target.py prof/sq_c3 forward backward linearize select > $OUT/sq_c3.log 2>&1 || exit 1
timeout -k 10 600 bash tools/pmc_counters.sh tools/pmc_target_c5.py prof/sq_c5 forward backward linearize > $OUT/sq_c5.log 2>&1 || exit 1
step "small measurements"
timeout -k 10 120 python3 tools/hbm_peak.py > $OUT/hbm_peak.json 2> /dev/null || exit 1
timeout -k 10 120 ./tools/micro/issue_rate > $OUT/issue_rate.log || exit 1
timeout -k 10 120 ./tools/micro/range_probe > $OUT/range_probe.log || exit 1
# (built beforehand: for f in issue_rate range_probe store_hazard store_hazard2; do hipcc --offload-arch=gfx950 -O2 -o tools/micro/$f tools/micro/$f.hip; done)
timeout -k 10 120 ./tools/micro/store_hazard > $OUT/store_hazard.log || exit 1
timeout -k 10 120 ./tools/micro/store_hazard2 >> $OUT/store_hazard.log || exit 1
timeout -k 10 300 python3 tools/f32_error.py > $OUT/f32_error.log 2>&1 || exit 1
timeout -k 10 600 python3 tools/f32_status_parity.py --out $OUT/status_parity.json > $OUT/status_parity.log 2>&1 || exit 1
timeout -k 10 300 python3 tools/c5_anomaly.py > $OUT/c5_anomaly.log 2>&1 || exit 1
timeout -k 10 300 python3 tools/fused_ab.py --f64 > $OUT/fused_ab.log 2>&1 || exit 1
timeout -k 10 300 python3 tools/c5_sweep.py > $OUT/c5_sweep.log 2>&1 || exit 1
timeout -k 10 300 python3 tools/sweep_scaling.py > $OUT/sweep_scaling.log 2>&1 || exit 1
timeout -k 10 300 python3 tools/other_systems.py > $OUT/other_systems.log 2>&1 || exit 1
cat $OUT/bench_default.json; cat $OUT/bench_f64.json
