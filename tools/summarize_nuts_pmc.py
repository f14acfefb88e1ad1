#!/usr/bin/env python3
"""gpurun_out/nuts_pmc/<tag>/<group>/ (tools/profile_nuts_pmc.sh) -> profiles/<out>.json: per-launch means of every
counter for the k_nuts dispatches, HBM bytes with the guide's gfx950 corrections (FETCH_SIZE KiB x 1024 x 2 for wide
coalesced reads, WRITE_SIZE KiB x 1024), and the derived per-leapfrog figures."""
import collections, csv, glob, json, re, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/nuts_pmc"
out_path = sys.argv[2] if len(sys.argv) > 2 else "profiles/r02_nuts_pmc.json"
kernel = sys.argv[3] if len(sys.argv) > 3 else "k_nuts"
res = {"command": "tools/profile_nuts_pmc.sh: rocprofv3 --kernel-trace --stats | --pmc <group> -- python3 tools/bench_nuts.py "
                  "(65 536 chains x D=1024, shared metric M^-1 = sigma^2, 3 warm-up + 3 timed transitions)",
       "corrections": "FETCH_SIZE x2 (gfx950 counts 128-B requests of wide coalesced reads as 64 B), WRITE_SIZE x1; both KiB",
       "configs": {}}
for tag in sorted(d.split("/")[-1] for d in glob.glob(src + "/*") if not d.endswith((".log", ".txt"))):
    cfg = {}
    log = open("%s/%s.trace.log" % (src, tag)).read()
    m = re.search(r"eps=(\S+) transitions=(\d+) ms/transition=(\S+) leapfrogs=(\d+) steps/s=(\S+) mean depth=(\S+) mean steps=(\S+)", log)
    cfg.update(eps=float(m.group(1)), ms_per_transition_under_trace=float(m.group(3)), leapfrog_steps_per_s_under_trace=float(m.group(5)),
               mean_depth=float(m.group(6)), mean_steps_per_chain=float(m.group(7)))
    stats = glob.glob("%s/%s/trace/*/*_kernel_stats.csv" % (src, tag))
    for r in csv.DictReader(open(stats[0])):
        if kernel in r["Name"]:
            cfg.setdefault("kernels", []).append({"name": r["Name"][:120], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                                   "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])})
    counters = collections.defaultdict(list)
    meta = {}
    for f in glob.glob("%s/%s/*/*/*_counter_collection.csv" % (src, tag)):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
                for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
                    if k in r:
                        meta[k] = r[k]
    cfg["dispatch"] = meta
    cfg["counters_mean_per_launch"] = {k: sum(v) / len(v) for k, v in sorted(counters.items())}
    c = cfg["counters_mean_per_launch"]
    leaves = cfg["mean_steps_per_chain"] * 65536
    d = {}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["hbm_read_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * 2
        d["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
        d["hbm_bytes_per_leapfrog"] = (d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]) / leaves
        t = cfg["kernels"][0]["avg_ns"] * 1e-9 if cfg.get("kernels") else None
        if t:
            d["hbm_GBps"] = (d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]) / t / 1e9
    if "TCC_HIT_sum" in c:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        d["l2_requests_per_leapfrog"] = (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) / leaves
    if "SQ_INSTS_VALU" in c:
        d["valu_insts_per_leapfrog"] = c["SQ_INSTS_VALU"] / leaves
        d["salu_insts_per_leapfrog"] = c.get("SQ_INSTS_SALU", 0) / leaves
        d["lds_insts_per_leapfrog"] = c.get("SQ_INSTS_LDS", 0) / leaves
    if "SQ_WAVE_CYCLES" in c:
        wc = c["SQ_WAVE_CYCLES"]
        d["wave_cycle_shares"] = {k: c[k] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                                       "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS") if k in c}
        d["wave_quadcycles_per_leapfrog"] = wc / leaves
    cfg["derived"] = d
    res["configs"][tag] = cfg
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
