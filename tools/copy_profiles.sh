#!/bin/bash
# Copy the artefacts of an evidence run (tools/final_run.sh <tag>) from gpurun_out/<tag>/ into profiles/ under a round prefix.
# Usage: bash tools/copy_profiles.sh <tag> <prefix, e.g. r04>
T=gpurun_out/$1; P=profiles/$2
for f in bench_pipelined bench_pipelined_again bench_group1 bench_nopipeline bench_linemod13 bench_dense16d bench_darknet_tiny bench_full640 bench_rccl_single_rank bench_rccl_single_rank_overlap; do
  [ -s $T/$f.json ] && cp $T/$f.json ${P}_$f.json
done
for f in kernel_stats.md step_dispatches.md timeline_pipelined.txt timeline_nopipeline.txt bench_norm.md bench_conv.md bench_wgrad_group.txt; do
  [ -s $T/$f ] && cp $T/$f ${P}_$f
done
[ -s $T/pytest_gpu.txt ] && cat $T/box.txt $T/pytest_gpu.txt $T/smoke.txt > ${P}_evidence_run.txt 2>/dev/null
[ -s $T/train.log ] && tail -5 $T/train.log >> ${P}_evidence_run.txt
[ -s gpurun_out/fullsize_parity.json ] && cp gpurun_out/fullsize_parity.json ${P}_fullsize_parity.json
ls -la ${P}_* | wc -l
