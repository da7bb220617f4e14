# PMC passes over the chunked frame kernel (k_hop) at 2048 / 333, B 32, T 1024; outputs under gpurun_out/hop_pmc_*
# (every pass under `timeout`; only counter names that are known to exist on this box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="python3 tools/bench_iter.py --n-fft 2048 --hop 333 --batch 32 --frames 1024 --launches 10 --rounds 1"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/hop_pmc_sq -- $CMD > gpurun_out/hop_pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/hop_pmc_inst -- $CMD > gpurun_out/hop_pmc_inst.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/hop_pmc_fetch -- $CMD > gpurun_out/hop_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/hop_pmc_write -- $CMD > gpurun_out/hop_pmc_write.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for g in ("sq","inst","fetch","write"):
    for f in glob.glob(f"gpurun_out/hop_pmc_{g}/*/*counter_collection.csv"):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_hop<16, 0, false>" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items(): print(g,k,f"{sum(v)/len(v):.5g}",len(v))
PY
