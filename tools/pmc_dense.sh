#!/bin/bash
# SQ / LDS counters of the dense-OT softmin kernels (rocprofv3 PMC, one counter group per pass; no tracing domains mixed in):
#   bash tools/pmc_dense.sh [out dir under gpurun_out]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-pmc_dense}
mkdir -p $OUT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVES" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$tag -o p -- python3 $R/bench.py --workload dense16d --steps 2 --warmup 1 > $OUT/log_$tag.txt 2>&1
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*/")):
    f = glob.glob(d + "*counter_collection.csv")
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "dense_softmin" not in k: continue
        acc[k[:70]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:70], r["Counter_Name"])] += 1
    for k, cs in acc.items():
        print(k); print("   " + "  ".join("%s=%.4g" % (c, v / max(n[(k, c)], 1)) for c, v in sorted(cs.items())))
PY
cat $OUT/summary.txt
rm -rf $OUT/SQ_* $OUT/GRBM_*
