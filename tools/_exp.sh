cd /root/repo; mkdir -p gpurun_out/r2q
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_fullsize_gpu.py -q -x > gpurun_out/r2q/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" gpurun_out/r2q/pytest.log | tail -2
python tools/bench_conv.py --kind fwd --set all > gpurun_out/r2q/conv_fwd.md 2>&1
for v in 1 2 3; do
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipelined', round(d['value'],1), round(d['ms_per_step'],3))"
done
python bench.py --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sequential', round(d['value'],1), round(d['ms_per_step'],3))"
