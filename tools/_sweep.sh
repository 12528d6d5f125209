run() { echo "tune: $*"; timeout 600 python bench.py --steps 99 --no-cpu-baseline --no-secondary --no-launch-events "$@" 2>/dev/null | grep '^{' | sed -E 's/.*"value": ([0-9.]+).*/\1/'; }
run
run --tune group_wgs=768
run --tune group_wgs=384
run --tune budget_div=3
run --tune budget_div=6
run --tune streams=5
run --tune group_flush=head_end
run
