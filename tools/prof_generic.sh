# the coverage path's three reference shapes under rocprofv3: kernel trace, then two counter passes each;
# outputs under gpurun_out/prof_gen/<tag>/...   usage (on the GPU box): bash tools/prof_generic.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_gen
run() {
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gen/$tag/kt -- python3 tools/bench_generic_one.py "$@" > gpurun_out/prof_gen_$tag.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/prof_gen/$tag/pmc1 -- python3 tools/bench_generic_one.py "$@" >> gpurun_out/prof_gen_$tag.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_gen/$tag/pmc2 -- python3 tools/bench_generic_one.py "$@" >> gpurun_out/prof_gen_$tag.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d gpurun_out/prof_gen/$tag/pmc3 -- python3 tools/bench_generic_one.py "$@" >> gpurun_out/prof_gen_$tag.log 2>&1
}
run f64_2048 2048 512 1024 16 f64
run f64_512ts 512 100 2048 64 f64 twosided 300
run f32_512ts 512 100 2048 64 f32 twosided 300
python3 tools/prof_generic_summary.py
