# rocprofv3 runs of the headline bench (on the GPU box via gpurun); outputs land in gpurun_out/r01_*.
# 1) kernel trace + stats of `python3 bench.py`; 2) PMC passes (separate runs, kernel dispatch only).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01_kt -- $CMD > gpurun_out/r01_kt.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01_pmc_fetch -- $CMD > gpurun_out/r01_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01_pmc_write -- $CMD > gpurun_out/r01_pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/r01_pmc_sq -- $CMD > gpurun_out/r01_pmc_sq.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r01_pmc_inst -- $CMD > gpurun_out/r01_pmc_inst.log 2>&1
tail -1 gpurun_out/r01_kt.log
