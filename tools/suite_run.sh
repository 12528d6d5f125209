#!/bin/bash
# One evidence run of the GPU suite exactly as the driver runs it (`python -m pytest tests/ -x -q -m gpu`), with the box
# it ran on and the full-size parity record beside the log.
# Usage (inside gpurun): bash tools/suite_run.sh <tag under gpurun_out/> [repeat count]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-suite}
N=${2:-1}
mkdir -p $O
cd $R
(hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id"; git rev-parse HEAD 2>/dev/null; md5sum kd-6d-pose-adlp_amd/csrc/libkd6d.so) > $O/box.txt 2>&1
for i in $(seq 1 $N); do
  rm -f gpurun_out/fullsize_parity.json
  python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/run$i.log 2>&1
  echo "run $i rc=$?" >> $O/summary.txt
  tail -1 $O/run$i.log >> $O/summary.txt
  cp gpurun_out/fullsize_parity.json $O/run${i}_fullsize_parity.json 2>/dev/null
  python - >> $O/summary.txt <<PY
import json
try:
    d = json.load(open("$O/run${i}_fullsize_parity.json"))
    recs = {k: v for k, v in d.items() if isinstance(v, dict) and "not_reproducible" in v}
    print("  records:", len(recs), " not_reproducible all empty:", all(v["not_reproducible"] == {} for v in recs.values()),
          " twin losses equal:", all(v.get("twin_losses_equal") for v in recs.values()))
except Exception as e:
    print("  no parity record:", e)
PY
done
cat $O/box.txt $O/summary.txt
