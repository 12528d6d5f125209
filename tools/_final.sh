set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
cd $R
python -X faulthandler -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/final/bench_pipelined.json 2> gpurun_out/final/bench_pipelined.err
python bench.py --no-pipeline --no-cpu-baseline > gpurun_out/final/bench_nopipeline.json 2> gpurun_out/final/bench_nopipeline.err
python bench.py --no-cpu-baseline --student darknet_tiny > gpurun_out/final/bench_darknet_tiny.json 2>/dev/null
python bench.py --no-cpu-baseline --frame full640 > gpurun_out/final/bench_full640.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/prof -o kd -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/final/prof.log 2>&1
cd $R
bash tools/pmc_traffic.sh > gpurun_out/final/pmc.log 2>&1
for f in bench_pipelined bench_nopipeline bench_darknet_tiny bench_full640; do tail -1 gpurun_out/final/$f.json | cut -c80-210; done
tail -c 600 gpurun_out/final/pmc.log
