#!/bin/bash
# End-of-round evidence run on the GPU box: tests, smoke, the bench lines, timelines, rocprofv3 kernel stats and the
# dispatch list of one replayed step, the norm / conv micro tables.
# Usage (inside gpurun): bash tools/final_run.sh <tag under gpurun_out/> [quick]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-final}
QUICK=${2:-}
mkdir -p $O
cd $R
if [ -z "$QUICK" ]; then
  (hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i unique) > $O/box.txt 2>&1; timeout 3000 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/pytest_gpu.txt
  grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3 >> $O/pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
fi
python bench.py > $O/bench_pipelined.json 2> $O/bench_pipelined.err
python bench.py --no-pipeline --no-cpu-baseline --no-secondary > $O/bench_nopipeline.json 2> $O/bench_nopipeline.err
python bench.py --teacher-group 1 --no-cpu-baseline --no-secondary > $O/bench_group1.json 2> $O/bench_group1.err
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-secondary --timeline > /dev/null 2> $O/timeline_pipelined.txt
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-secondary --no-pipeline --timeline > /dev/null 2> $O/timeline_nopipeline.txt
if [ -z "$QUICK" ]; then
  python bench.py --workload linemod13 --no-cpu-baseline > $O/bench_linemod13.json 2>/dev/null
  python bench.py --workload dense16d > $O/bench_dense16d.json 2>/dev/null
  python bench.py --student darknet_tiny --no-cpu-baseline --no-secondary > $O/bench_darknet_tiny.json 2>/dev/null
  python bench.py --frame full640 --no-cpu-baseline --steps 30 > $O/bench_full640.json 2>/dev/null
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --rccl-single-rank --no-cpu-baseline --no-secondary > $O/bench_rccl_single_rank.json 2> $O/bench_rccl_single_rank.err
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --rccl-single-rank --exchange overlap --no-cpu-baseline --no-secondary > $O/bench_rccl_single_rank_overlap.json 2> $O/bench_rccl_single_rank_overlap.err
  python bench.py --no-cpu-baseline --no-secondary > $O/bench_pipelined_again.json 2>/dev/null
  python tools/bench_norm.py > $O/bench_norm.md 2>&1
  python tools/bench_conv.py --kind all --set all > $O/bench_conv.md 2>&1
  python tools/bench_wgrad_group.py > $O/bench_wgrad_group.txt 2>&1
  timeout 300 python train_kd.py --config_file configs/ape.yaml --config_file_t configs/ape.yaml --backbone darknet_tiny_h --backbone_t darknet53 --kd_weight 5. --working_dir $O/train --synthetic --launch pipeline --teacher_group 3 --max_iters 30 > $O/train.log 2>&1; echo "train rc=$?" >> $O/train.log
  rm -rf $O/train/*.pth
fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o kd -- python3 $R/bench.py --steps 21 --warmup 3 --no-cpu-baseline --no-secondary > $O/prof.log 2>&1
cd $R
DB=$(ls $O/prof/*/kd_results.db $O/prof/kd_results.db 2>/dev/null | tail -1)
python tools/rocprof_summary.py $DB $O/kernel_stats.md 29 > /dev/null
python tools/step_sequence.py $DB $O/step_dispatches.md 3 > $O/step_dispatches_head.txt 2>&1
rm -rf $O/prof
cat $O/pytest_gpu.txt $O/smoke.txt 2>/dev/null; tail -2 $O/train.log 2>/dev/null; cat $O/step_dispatches_head.txt
for f in pipelined pipelined_again group1 nopipeline linemod13 darknet_tiny full640 rccl_single_rank rccl_single_rank_overlap; do [ -s $O/bench_$f.json ] && python -c "import json; d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],3), d['finite'], d['roofline']['frac'] if 'roofline' in d else None)"; done
