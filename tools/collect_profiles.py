"""Copies the judged summaries of a tools/gpu_profile.sh run from gpurun_out/prof (scratch) into
profiles/rNN (tracked): bench JSON lines, rocprofv3 --kernel-trace --stats summaries, and the PMC
traffic of the hot kernels (FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md)."""
import csv, glob, json, os, shutil, sys


def newest(pattern):
    """gpurun merges every run's files into gpurun_out/: take the most recent match."""
    m = sorted(glob.glob(pattern), key=os.path.getmtime)
    return m[-1:] 

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = "gpurun_out/prof", os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)
for name, out in (("bench_default.json", "bench_default_f32.json"), ("bench_f64.json", "bench_f64.json")):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, out))
for dt in ("f32", "f64"):
    for f in newest(f"{src}/trace_{dt}/*/*_kernel_stats.csv"):
        shutil.copy(f, os.path.join(dst, f"rocprofv3_kernel_stats_{dt}.csv"))
    fetch = newest(f"{src}/pmc_fetch_{dt}/*/*_counter_collection.csv")
    write = newest(f"{src}/pmc_write_{dt}/*/*_counter_collection.csv")
    if not (fetch and write):
        continue
    shutil.copy(fetch[0], os.path.join(dst, f"rocprofv3_pmc_FETCH_SIZE_{dt}.csv"))
    shutil.copy(write[0], os.path.join(dst, f"rocprofv3_pmc_WRITE_SIZE_{dt}.csv"))

    def mean_counter(path, kern):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kern in r["Kernel_Name"]]
        return (sum(v) / len(v), len(v)) if v else (0.0, 0)

    out = {"note": "rocprofv3 --pmc passes (separate runs, --kernel-trace only) of `python bench.py --no-cpu-baseline "
                   f"--steps 5 --dtype {dt}` (B=4096, N=200). FETCH_SIZE/WRITE_SIZE are in KiB; per MI355X_MICROARCH.md "
                   "(HBM section) FETCH_SIZE on gfx950 reports exactly half of a wide coalesced streaming read, so "
                   "read bytes = 2 * FETCH_SIZE * 1024.",
           "dtype": dt, "batch": 4096, "horizon": 200, "kernels": {}}
    for kern in ("backward_fused16_kernel", "backward_tile16_kernel", "linearize_kernel", "forward_ring_kernel"):
        f, nf = mean_counter(fetch[0], kern)
        w, nw = mean_counter(write[0], kern)
        out["kernels"][kern] = {"FETCH_SIZE_KiB_mean": f, "WRITE_SIZE_KiB_mean": w, "launches": nf,
                                "read_bytes_corrected": 2 * f * 1024, "write_bytes": w * 1024,
                                "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024}
    json.dump(out, open(os.path.join(dst, f"pmc_traffic_{dt}.json"), "w"), indent=1)
    print(dt, {k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out["kernels"].items()}, "MB per launch")
    valu = newest(f"{src}/pmc_valu_{dt}/*/*_counter_collection.csv")
    if valu:
        shutil.copy(valu[0], os.path.join(dst, f"rocprofv3_pmc_SQ_INSTS_VALU_{dt}.csv"))
        vo = {"note": "rocprofv3 --pmc SQ_INSTS_VALU pass (its own run, --kernel-trace only) of `python bench.py --no-cpu-baseline "
                      f"--no-solve-extra --steps 5 --dtype {dt}` (B=4096, N=200): vector-ALU instructions issued per launch, summed "
                      "over all waves of the launch (a wave64 instruction counts once).",
              "dtype": dt, "batch": 4096, "horizon": 200, "kernels": {}}
        for kern in ("backward_fused16_kernel", "backward_tile16_kernel", "linearize_kernel", "forward_ring_kernel"):
            v, nv = mean_counter(valu[0], kern)
            if nv:
                vo["kernels"][kern] = {"SQ_INSTS_VALU_per_launch": v, "launches": nv}
        json.dump(vo, open(os.path.join(dst, f"pmc_valu_{dt}.json"), "w"), indent=1)
        print(dt, "VALU instructions per launch", {k: round(v["SQ_INSTS_VALU_per_launch"] / 1e6, 2) for k, v in vo["kernels"].items()}, "M")
# the c5 shard (n=16, m=8, N=500, B=128) and the c4 MPC shard (1024 instances): stats + traffic of every kernel seen
for w, what in (("c5", "tools/pmc_target_c5.py (c5 shard: n=16 m=8 N=500 B=128 f32, 4 iterations)"),
                ("mpc", "tools/pmc_target_mpc.py (c4 shard: 1024 MPC instances, N=200, 2 cold + 6 warm steps, f32)")):
    for f in newest(f"{src}/trace_{w}/*/*_kernel_stats.csv"):
        shutil.copy(f, os.path.join(dst, f"rocprofv3_kernel_stats_{w}.csv"))
    fetch = newest(f"{src}/pmc_fetch_{w}/*/*_counter_collection.csv")
    write = newest(f"{src}/pmc_write_{w}/*/*_counter_collection.csv")
    if not (fetch and write):
        continue
    shutil.copy(fetch[0], os.path.join(dst, f"rocprofv3_pmc_FETCH_SIZE_{w}.csv"))
    shutil.copy(write[0], os.path.join(dst, f"rocprofv3_pmc_WRITE_SIZE_{w}.csv"))
    per = {}
    for path, key in ((fetch[0], "fetch"), (write[0], "write")):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ilqr::", "")
            per.setdefault(k, {"fetch": [], "write": []})[key].append(float(r["Counter_Value"]))
    out = {"note": f"rocprofv3 --pmc passes (separate runs, --kernel-trace only) of {what}. FETCH_SIZE/WRITE_SIZE are in "
                   "KiB; read bytes = 2 * FETCH_SIZE * 1024 (gfx950 correction of MI355X_MICROARCH.md, HBM section).",
           "kernels": {}}
    for k, d in sorted(per.items()):
        f = sum(d["fetch"]) / max(1, len(d["fetch"]))
        wv = sum(d["write"]) / max(1, len(d["write"]))
        out["kernels"][k] = {"FETCH_SIZE_KiB_mean": f, "WRITE_SIZE_KiB_mean": wv, "launches": len(d["fetch"]),
                             "hbm_bytes_per_launch": 2 * f * 1024 + wv * 1024}
    json.dump(out, open(os.path.join(dst, f"pmc_traffic_{w}.json"), "w"), indent=1)
    print(w, {k: round(v["hbm_bytes_per_launch"] / 1e6, 2) for k, v in out["kernels"].items()}, "MB per launch")
for name in ("sq_c3/summary.txt", "sq_c5/summary.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, "sq_counters_" + name.split("/")[0][3:] + ".txt"))
for name in ("hbm_peak.json", "issue_rate.log", "range_probe.log", "store_hazard.log", "f32_error.log", "c5_sweep.log", "sweep_scaling.log",
             "status_parity.json", "c5_anomaly.log", "fused_ab.log", "other_systems.log"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
for dt in ("f32", "f64", "c5", "mpc"):
    p = os.path.join(dst, f"rocprofv3_kernel_stats_{dt}.csv")
    if os.path.exists(p):
        print(dt, [(r["Name"][:48], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1)) for r in csv.DictReader(open(p))
                   if float(r["Percentage"]) > 0.5])
