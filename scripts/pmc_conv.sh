# dev: SQ instruction-mix / stall counters of one conv layer through scripts/bench_layer.py
#   usage: bash scripts/pmc_conv.sh <op> "<cin> <cout> <size>" <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OP=$1; L=$2; T=$3
A="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES"
B="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM"
C="SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_IFETCH"
for P in A B C; do
  eval CN=\$$P
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $CN --output-format csv -d gpurun_out/pmc${P}_$T -o r -- python scripts/bench_layer.py $OP $L 3 5 > gpurun_out/pmc${P}_$T.log 2>&1 || exit 1
done
