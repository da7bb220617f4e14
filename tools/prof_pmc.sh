# PMC passes for the fused iteration kernel (run on the GPU box via gpurun). Counters are collected in
# separate runs, without any trace domain besides kernel dispatch (see MI355X_MICROARCH.md, rocprofv3 PMC slots).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ARGS="${BENCH_ARGS:---launches 10 --rounds 1}"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/bench_iter.py $ARGS > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/bench_iter.py $ARGS > gpurun_out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_sq -- python3 tools/bench_iter.py $ARGS > gpurun_out/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_inst -- python3 tools/bench_iter.py $ARGS > gpurun_out/pmc_inst.log 2>&1
find gpurun_out -name "*counter_collection.csv" | head
