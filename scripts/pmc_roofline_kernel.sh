# Matrix-pipe / instruction-mix counters of the roofline kernel (conv3d_fwd_k3_persist<1, true>, 8 -> 8 channels at 128^3)
# through the C ABI (scripts/bench_layer.py); two separate --pmc passes, kernel trace only.  Summary -> profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
B="SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES"
for P in A B; do
  eval CN=\$$P
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $CN --output-format csv -d gpurun_out/pmc${P}_roof -o r -- python scripts/bench_layer.py fwd 8 8 128 3 5 > gpurun_out/pmc${P}_roof.log 2>&1 || exit 1
done
python scripts/pmc_show.py roof persist > gpurun_out/pmc_roof_summary.txt
cat gpurun_out/pmc_roof_summary.txt
