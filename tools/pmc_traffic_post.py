"""Post-processing of tools/pmc_traffic.sh: per-family HBM bytes per step from the two rocprofv3 PMC passes.

    python tools/pmc_traffic_post.py <dir with FETCH_SIZE/ and WRITE_SIZE/>
"""
import sys
OUT = sys.argv[1]
import csv, glob, json, collections
out = {}
steps_seen = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(OUT + "/%s/*counter_collection.csv" % c)[0]
    per = collections.defaultdict(lambda: [0, 0.0])
    steps_seen[c] = 0
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "clip_adamw" in n:               # one optimizer launch closes every step of the invocation, whatever its
            steps_seen[c] += 1              # mix of warm-up, timed, probe and instrumented steps is
        # the convolution family: every implicit-GEMM kernel, the split-K finalize, the grouped weight gradient with its
        # reduce, and (round 4) the launch that sums the weight-gradient slabs -- rounds 1-3 matched on "conv" alone and
        # left the grouped weight gradient and the finalize launches out
        if not ("conv" in n or "wgrad_group" in n or "splitk_finalize" in n or "grad_acc_resolve" in n) or "pack_dgrad" in n:
            continue
        key = ("smallc" if "smallc" in n else "halo" if "halo" in n else "glds" if "glds" in n
               else "wgrad_group" if "wgrad_group" in n else "wgrad_resolve" if "grad_acc_resolve" in n
               else "wgrad" if "wgrad" in n else "splitk_finalize" if "finalize" in n else "igemm")
        per[key][0] += 1
        per[key][1] += float(r["Counter_Value"])
    out[c] = {k: {"launches": v[0], "counter_sum": v[1]} for k, v in per.items()}
if steps_seen["FETCH_SIZE"] != steps_seen["WRITE_SIZE"] or steps_seen["FETCH_SIZE"] == 0:
    raise SystemExit("the two PMC passes saw %s optimizer launches: not the same bench invocation" % steps_seen)
steps = steps_seen["FETCH_SIZE"]
# MI355X_MICROARCH.md 'HBM': FETCH_SIZE is in KiB-like units of 64 B requests tallied at half size on gfx950:
# bytes = FETCH_SIZE * 1024 * 2 for wide coalesced reads; WRITE_SIZE * 1024 reads exactly for 16-B stores / atomics
res = {"steps_profiled": steps, "per_family": {}, "note": "FETCH_SIZE doubled per the gfx950 correction"}
tot_f = tot_w = 0.0
for fam in set(out["FETCH_SIZE"]) | set(out["WRITE_SIZE"]):
    fr = out["FETCH_SIZE"].get(fam, {"launches": 0, "counter_sum": 0.0})
    wr = out["WRITE_SIZE"].get(fam, {"launches": 0, "counter_sum": 0.0})
    fb = fr["counter_sum"] * 1024.0 * 2.0 / steps
    wb = wr["counter_sum"] * 1024.0 / steps
    res["per_family"][fam] = {"launches_per_step": fr["launches"] / steps, "fetch_MB_per_step": fb / 1e6, "write_MB_per_step": wb / 1e6}
    tot_f += fb; tot_w += wb
res["conv_family_hbm_MB_per_step"] = (tot_f + tot_w) / 1e6
res["conv_family_fetch_MB_per_step"] = tot_f / 1e6
res["conv_family_write_MB_per_step"] = tot_w / 1e6
json.dump(res, open(OUT + "/traffic.json", "w"), indent=1)
print(json.dumps(res)[:1500])
