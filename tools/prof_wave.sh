# k_wave_iter under rocprofv3: kernel trace, then counter passes, for the cases of tools/bench_wave.py given by index
# usage (on the GPU box): bash tools/prof_wave.sh <case index> ...   outputs under gpurun_out/prof_wave/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_wave gpurun_out/pmc_wave*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wave/kt -- python3 tools/bench_wave.py one "$@" > gpurun_out/prof_wave.log 2>&1
python3 tools/kt_summary.py gpurun_out/prof_wave/kt 8
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_wave1 -- python3 tools/bench_wave.py one "$@" >> gpurun_out/prof_wave.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_wave2 -- python3 tools/bench_wave.py one "$@" >> gpurun_out/prof_wave.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_wave3 -- python3 tools/bench_wave.py one "$@" >> gpurun_out/prof_wave.log 2>&1
# (a fourth pass with the TCC counters FETCH_SIZE / WRITE_SIZE ran into gpurun's limit on this script's first use: left out)
python3 tools/pmc_summary.py k_wave_iter ${WAVE_FRAMES:-16384}
