#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/profile_bench.sh <tag> (gpurun_out/prof_<tag>_*) into
  <dst>/<tag>_leapfrog_kernel_stats.csv   the --kernel-trace --stats table of the bench command
  <dst>/<tag>_leapfrog_pmc.json           per-launch duration, HBM bytes, and what the SAME run's bench line said
dst = gpurun_out/ with --stage-only (on the GPU box: the raw per-dispatch CSVs are too big to carry back), else profiles/.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a 16-B/lane coalesced stream
(MI355X_MICROARCH.md, HBM section), so reads are doubled; WRITE_SIZE is exact for 16-B/lane stores."""
import collections, csv, glob, json, os, shutil, sys
args = [a for a in sys.argv[1:] if not a.startswith("--")]
tag = args[0] if args else "r03"
kernel = args[1] if len(args) > 1 else "k_leapfrog1<8, idhmc::DiagGaussian<8>, 3>"   # the headline variant (7 = gradient-recompute mode)
src = "gpurun_out"
dst = "gpurun_out" if "--stage-only" in sys.argv else "profiles"
staged = f"{src}/{tag}_leapfrog_pmc.json"
if dst == "profiles" and os.path.exists(staged):          # summaries made on the box: just copy
    shutil.copy(staged, f"profiles/{tag}_leapfrog_pmc.json")
    shutil.copy(f"{src}/{tag}_leapfrog_kernel_stats.csv", f"profiles/{tag}_leapfrog_kernel_stats.csv")
    print(open(staged).read())
    sys.exit(0)
stats = glob.glob(f"{src}/prof_{tag}_trace/*/*_kernel_stats.csv")[0]
shutil.copy(stats, f"{dst}/{tag}_leapfrog_kernel_stats.csv")
avg_ns = calls = None
other = {}
for r in csv.DictReader(open(stats)):
    if kernel in r["Name"]:
        avg_ns, calls, name = float(r["AverageNs"]), int(r["Calls"]), r["Name"]
    for key in ("k_xchg_sum", "k_placement_probe"):
        if key in r["Name"]:
            other[r["Name"].split("(")[0].replace("void idhmc::", "")] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
vals = {}
for cname in ("fetch", "write"):
    f = glob.glob(f"{src}/prof_{tag}_{cname}/*/*_counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        vals[k] = sum(v) / len(v)
fetch_b = vals["FETCH_SIZE"] * 1024 * 2
write_b = vals["WRITE_SIZE"] * 1024
line = json.loads(open(f"{src}/prof_{tag}_trace.json").read().strip().splitlines()[-1])
pl = line["state_placement"]
alg = line["roofline"]["algorithmic_bytes_per_launch"]
out = {"kernel": name, "calls": calls, "avg_duration_ns": avg_ns,
       "achieved_GBps_from_profile": alg / avg_ns, "frac_of_8000_from_profile": alg / avg_ns / 8000.0,
       "same_run_bench_line": {"kernel_ms_hip_events": line["roofline"]["kernel_ms"], "frac": line["roofline"]["frac"],
                               "value": line["value"], "state_placement": pl,
                               "dense_single_step_kernel_ms_under_the_profiler": (line.get("dense") or {}).get("single_step_sweeps", {}).get("kernel_ms")},
       "placement_mode": "good" if pl["single_array_GBps"] > 0 and pl["probe_GBps"] >= 1.10 * pl["single_array_GBps"] else "bad",
       "FETCH_SIZE_KiB_mean": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_mean": vals["WRITE_SIZE"],
       "hbm_read_bytes_per_launch": fetch_b, "hbm_write_bytes_per_launch": write_b,
       "hbm_bytes_per_launch": fetch_b + write_b, "hbm_over_algorithmic": (fetch_b + write_b) / alg,
       "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads), WRITE_SIZE x1",
       "other_kernels": other,
       "command": "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 bench.py --steps 200 --warmup 20 --no-cpu --no-cfg3"}
json.dump(out, open(f"{dst}/{tag}_leapfrog_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
