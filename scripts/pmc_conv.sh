cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES"
B="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES"
C="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE"
for L in "8 8 128" "16 16 64" "8 32 128"; do
  T=$(echo $L | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $A --output-format csv -d gpurun_out/pmcA_$T -o r -- python scripts/bench_layer.py fwd $L 3 5 > gpurun_out/pmcA_$T.log 2>&1 &&
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $B --output-format csv -d gpurun_out/pmcB_$T -o r -- python scripts/bench_layer.py fwd $L 3 5 > gpurun_out/pmcB_$T.log 2>&1 &&
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmcC_$T -o r -- python scripts/bench_layer.py fwd $L 3 5 > gpurun_out/pmcC_$T.log 2>&1 || exit 1
done
ls gpurun_out | grep pmc
