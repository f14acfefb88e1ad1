#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/profile_bench.sh (gpurun_out/prof_*) into profiles/<tag>_*.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a 16-B/lane coalesced
stream (MI355X_MICROARCH.md, HBM section), so reads are doubled; WRITE_SIZE is exact for 16-B/lane stores."""
import collections, csv, glob, json, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_leapfrog"
kernel = sys.argv[2] if len(sys.argv) > 2 else "k_leapfrog1<8, idhmc::DiagGaussian<8>, 3>"   # the headline variant (7 = gradient-recompute mode)
src = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out"
stats = glob.glob(f"{src}/prof_trace/*/*_kernel_stats.csv")[0]
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
avg_ns = calls = None
for r in csv.DictReader(open(stats)):
    if kernel in r["Name"]:
        avg_ns, calls, name = float(r["AverageNs"]), int(r["Calls"]), r["Name"]
vals = {}
for cname in ("fetch", "write"):
    f = glob.glob(f"{src}/prof_{cname}/*/*_counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        vals[k] = sum(v) / len(v)
fetch_b = vals["FETCH_SIZE"] * 1024 * 2
write_b = vals["WRITE_SIZE"] * 1024
out = {"kernel": name, "calls": calls, "avg_duration_ns": avg_ns,
       "FETCH_SIZE_KiB_mean": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_mean": vals["WRITE_SIZE"],
       "hbm_read_bytes_per_launch": fetch_b, "hbm_write_bytes_per_launch": write_b,
       "hbm_bytes_per_launch": fetch_b + write_b,
       "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads), WRITE_SIZE x1",
       "command": "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 bench.py --steps 200 --warmup 20 --no-cpu"}
json.dump(out, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
