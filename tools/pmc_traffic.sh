#!/bin/bash
# HBM traffic of the conv kernel family per step (roofline.traffic): rocprofv3 PMC, one counter per pass
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950), eager launches, no tracing domains mixed in.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_traffic
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-secondary --no-launch-events > $OUT/log_$c.txt 2>&1
done
python3 $R/tools/pmc_traffic_post.py $OUT
