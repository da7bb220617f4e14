# launch gaps at a small (training-like) shape; output under gpurun_out/prof_small
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_small
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_small -- python3 tools/bench_iter.py --batch 16 --frames 256 --n-fft 1024 --launches 100 --rounds 2 > gpurun_out/prof_small.log 2>&1
grep chunk gpurun_out/prof_small.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_small/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_fused" in r["Kernel_Name"] and "istft" not in r["Kernel_Name"]]
rows = rows[-100:]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
g = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3 for i in range(len(rows) - 1)]
print(f"kernel {sum(d)/len(d):.1f} us  gap {sum(g)/len(g):.1f} us (min {min(g):.1f} max {max(g):.1f})")
PY
