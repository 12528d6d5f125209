timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -k "640" 2>&1 | tail -2
run() { echo "run: $*"; timeout 600 python bench.py --steps 60 --no-cpu-baseline --no-secondary --no-launch-events "$@" 2>/dev/null | grep '^{' | sed -E 's/.*"value": ([0-9.]+).*/\1/'; }
run --frame full640
run --frame full640 --opt conv.smallc_wmax=256
run --frame full640
