mkdir -p gpurun_out/r2f2
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv or batchnorm" 2>&1 | tail -2
python tools/bench_conv.py --kind fwd --set teacher --only head > gpurun_out/r2f2/fwd.txt 2>&1; cat gpurun_out/r2f2/fwd.txt | grep head
python tools/bench_norm.py 2>&1 | grep "bwd_apply" | awk -F'|' '{printf "%s %s | ",$2,$7}'; echo
for rep in 1 2; do
python bench.py --no-cpu-baseline > gpurun_out/r2f2/pipe_$rep.json 2>/dev/null
python bench.py --no-cpu-baseline --no-pipeline > gpurun_out/r2f2/seq_$rep.json 2>/dev/null
done
python - <<PY
import json,glob
for n in sorted(glob.glob("gpurun_out/r2f2/*.json")):
    d=json.loads(open(n).read().strip().splitlines()[-1]); print(n.split("/")[-1],round(d["value"],1),round(d["ms_per_step"],3))
PY
