reh() { echo "reh: $*"; MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --steps 60 --rccl-single-rank --no-cpu-baseline --no-secondary --no-launch-events "$@" 2>gpurun_out/e.err >gpurun_out/e.out; grep '^{' gpurun_out/e.out | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), d['config']['teacher_group'], d['config']['exchange'], d['config']['exchange_schedule'], d['config']['launch'][:40], d['finite'])" || tail -5 gpurun_out/e.err; }
reh
reh --teacher-group 1
reh --exchange overlap
reh
timeout 600 python bench.py --steps 60 --no-cpu-baseline --no-secondary --no-launch-events 2>/dev/null | grep '^{' | sed -E 's/.*"value": ([0-9.]+).*/no pg: \1/'
