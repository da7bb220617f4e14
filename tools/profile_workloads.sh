# rocprofv3 runs of every bench workload (on the GPU box via gpurun); outputs land in gpurun_out/${TAG}_<W>_*.
# Per workload: 1) kernel trace + stats; 2) PMC passes (separate runs, kernel dispatch only - never combined with trace domains).
TAG=${1:-r06}
shift
WL=${@:-C2 C4 C3 C5 F64 S32 W400}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m spectrogram_inversion_amd.build --hash > gpurun_out/${TAG}_csrc_sha1.txt   # the sources the profiled library was built from
for W in $WL; do
  CMD="python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extra --no-pmc --no-h2d"
  if [ $W = C5 ]; then CMD="$CMD --outer 2"; fi        # (two optimizer.step calls: the per-dispatch counter files of 50 would not fit gpurun_out)
  P=gpurun_out/${TAG}_${W}
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_kt -- $CMD > ${P}_kt.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d ${P}_pmc_fetch -- $CMD > ${P}_pmc_fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d ${P}_pmc_write -- $CMD > ${P}_pmc_write.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d ${P}_pmc_sq -- $CMD > ${P}_pmc_sq.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d ${P}_pmc_inst -- $CMD > ${P}_pmc_inst.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d ${P}_pmc_mfma -- $CMD > ${P}_pmc_mfma.log 2>&1
  tail -c 300 ${P}_kt.log
done
