#!/bin/bash
# HBM traffic of the conv kernel family per step (roofline.traffic): rocprofv3 PMC, one counter per pass
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950), no tracing domains mixed in.
#   bash tools/pmc_traffic.sh            eager launches, one teacher forward per step (rounds 1-3)
#   bash tools/pmc_traffic.sh graph      the default replayed schedule: grouped teacher pass, group 3 -- 30 timed steps; the
#                                        invocation's capture warm-up adds 9 student steps without a teacher and 3 teacher
#                                        passes without a student, i.e. the per-step average is not skewed by it
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_traffic
mkdir -p $OUT
if [ "$1" = "graph" ]; then ARGS="--steps 30 --warmup 3"; else ARGS="--steps 2 --warmup 1 --no-graph"; fi
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-secondary --no-launch-events > $OUT/log_$c.txt 2>&1
done
python3 $R/tools/pmc_traffic_post.py $OUT
