mkdir -p gpurun_out/r2j2
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" 2>&1 | tail -2
python bench.py --no-cpu-baseline > gpurun_out/r2j2/pipe.json 2>gpurun_out/r2j2/pipe.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r2j2/pipe.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["achieved"], r["frac"], r["avg_launch_us"], r["conv_ms_per_step"]); print(r["by_kind"])
PY
