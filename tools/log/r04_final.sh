# round 4, closing run: suite, smoke, the driver's bench command, every workload's line, rocprofv3 summaries of every workload
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r04_gpu_tests_tail.txt
cat gpurun_out/r04_gpu_tests_tail.txt
python3 __graft_entry__.py --smoke 2>&1 | tail -2
python3 bench.py --keep-pmc gpurun_out/r04_pmc_raw > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
tail -c 200 gpurun_out/r04_bench_default.err
for W in C1 C3 C4; do python3 bench.py --workload $W > gpurun_out/r04_bench_$W.json 2> gpurun_out/r04_bench_$W.err; done
python3 bench.py --workload C3 --asym > gpurun_out/r04_bench_C3_asym.json 2> /dev/null
for v in baseline main wolfe; do python3 bench.py --workload C5 --c5-variant $v > gpurun_out/r04_bench_C5_$v.json 2> /dev/null; done
python3 bench.py --workload C5 --c5-variant memory --outer 8 --steps 3 --warmup 1 > gpurun_out/r04_bench_C5_memory.json 2> /dev/null
SPECINV_EXACT=0 python3 bench.py --no-extra > gpurun_out/r04_bench_C2_approx.json 2> /dev/null
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04_bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d['roofline']
        print(f.split('r04_bench_')[1][:-5], round(d['value'] / 1e6, 2), 'M', round(d['ms_per_step'], 3), 'ms', r['bound'], round(r['frac'] or 0, 3), round(r['launch_ms'], 4), d.get('check', {}).get('ok'), round(d.get('bench_seconds', 0)), 's')
    except Exception as e:
        print(f, 'unreadable', e)
PY
bash tools/profile_workloads.sh r04 > gpurun_out/r04_profile_workloads.log 2>&1
python3 tools/collect_workload_profiles.py r04 2>&1 | tail -6
# the coverage path: float64 2048 / 512 and float32 512 two-sided
rm -rf gpurun_out/r04_generic_kt gpurun_out/r04_generic_pmc_*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_generic_kt -- python3 tools/bench_generic_r04.py > gpurun_out/r04_generic_kt.log 2>&1
cp $(find gpurun_out/r04_generic_kt -name "*kernel_stats.csv" | head -1) gpurun_out/r04_generic_kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r04_generic_pmc_fetch -- python3 tools/bench_generic_one.py 2048 512 1024 16 f64 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r04_generic_pmc_write -- python3 tools/bench_generic_one.py 2048 512 1024 16 f64 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r04_generic_pmc_sq -- python3 tools/bench_generic_one.py 2048 512 1024 16 f64 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, json, collections
out = {"command": "rocprofv3 --pmc <group> -- python3 tools/bench_generic_one.py 2048 512 1024 16 f64 (one run per counter group)", "kernels": {}}
for grp in ("fetch", "write", "sq"):
    for f in glob.glob(f"gpurun_out/r04_generic_pmc_{grp}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            for k in ("k_iter_pair", "k_ola"):
                if k in r["Kernel_Name"]:
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in agg.items():
            out["kernels"].setdefault(k, {}).update({n: sum(v) / len(v) for n, v in c.items()})
for k, c in out["kernels"].items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        c["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    if c.get("GRBM_GUI_ACTIVE"):
        c["valu_issue_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        c["lds_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("SQ_WAVE_CYCLES"):
        c["wait_share_of_wave_life"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
json.dump(out, open("gpurun_out/r04_generic_pmc.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1)[:1500])
PY
