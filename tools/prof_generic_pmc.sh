# (every pass under `timeout`: an unknown counter name makes rocprofv3 abort and then hang until killed)
# PMC passes over the generic iteration kernel (k_iter_pair) at the C2 shape; outputs under gpurun_out/gen_pmc_*
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="python3 tools/bench_iter.py --generic --launches 10 --rounds 1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gen_kt -- $CMD > gpurun_out/gen_kt.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/gen_pmc_sq -- $CMD > gpurun_out/gen_pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/gen_pmc_inst -- $CMD > gpurun_out/gen_pmc_inst.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for g in ("sq","inst"):
    for f in glob.glob(f"gpurun_out/gen_pmc_{g}/*/*counter_collection.csv"):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_iter_pair" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items(): print(g,k,sum(v)/len(v),len(v))
PY
tail -2 gpurun_out/gen_kt.log
