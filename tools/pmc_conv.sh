cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "0" "1" "2"; do
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $pass | cut -d' ' -f1)
    KD6D_CONV_HALO=$cfg rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/r1j/pmc_${cfg}_$tag -o p -- python3 $R/tools/bench_conv.py --kind fwd --only t.head.tower --iters 6 --eager > $R/gpurun_out/r1j/log_${cfg}_$tag.txt 2>&1
  done
done
ls -R $R/gpurun_out/r1j | head -50
