#!/bin/bash
# SQ / LDS / L2 counters of one conv layer (rocprofv3 PMC, one counter group per pass, eager launches):
#   bash tools/pmc_conv.sh <layer name of tools/bench_conv.py> [fwd|dgrad|wgrad] [out dir under gpurun_out] [batch]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
LAYER=${1:-t.init}; KIND=${2:-fwd}; OUT=$R/gpurun_out/${3:-pmc_conv}; BATCH=${4:-16}
mkdir -p $OUT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVES" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$tag -o p -- python3 $R/tools/bench_conv.py --kind $KIND --only $LAYER --batch $BATCH --iters 6 --eager > $OUT/log_$tag.txt 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*/")):
    f = glob.glob(d + "*counter_collection.csv")
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "conv" not in k: continue
        acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:60], r["Counter_Name"])] += 1
    for k, cs in acc.items():
        print(k); print("   " + "  ".join("%s=%.3g" % (c, v / max(n[(k, c)], 1)) for c, v in sorted(cs.items())))
PY
