#!/usr/bin/env python3
"""gpurun_out/dense_nuts_pmc/<group>/ (tools/profile_dense_nuts_pmc.sh) -> profiles/r03_dense_nuts_pmc.json: counters of k_nuts<DenseMvnCoop> per launch:
launches 3-5 (single transitions, after 2 warm ones) averaged, and the launch of 20 transitions.  HBM bytes with the guide's gfx950 corrections
(FETCH_SIZE KiB x 1024 x 2, WRITE_SIZE KiB x 1024)."""
import collections, csv, glob, json, re, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/dense_nuts_pmc"
out_path = sys.argv[2] if len(sys.argv) > 2 else "profiles/r03_dense_nuts_pmc.json"
steps = json.loads(re.search(r"leapfrog steps per k_nuts launch: (\[.*\])", open(src + "/trace.log").read()).group(1))
rows = [r for r in csv.DictReader(open(glob.glob(src + "/trace/*/*_kernel_trace.csv")[0])) if "k_nuts" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
assert len(dur) == len(steps) == 6
per = collections.defaultdict(list)      # counter -> per-launch values in launch order
for f in sorted(glob.glob(src + "/*/*/*_counter_collection.csv")):
    byc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_nuts" in r["Kernel_Name"]:
            byc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, v in byc.items():
        per[k] = [x for _, x in sorted(v)]
flop = 2.0 * 256 * 256
def part(idx):
    n = len(idx)
    c = {k: sum(v[i] for i in idx) / n for k, v in per.items() if len(v) == 6}
    st, us = sum(steps[i] for i in idx) / n, sum(dur[i] for i in idx) / n
    d = {"leapfrog_steps": st, "us_under_kernel_trace": us, "leapfrog_steps_per_s": st / us * 1e6, "mfma_TFLOPs": st * flop / us / 1e6,
         "mfma_frac_of_78.6": st * flop / us / 1e6 / 78.6}
    if "FETCH_SIZE" in c:
        d["hbm_bytes_per_leapfrog"] = (c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024) / st
        d["hbm_GBps"] = (c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024) / us / 1e3
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        d["mfma_busy_over_sq_busy_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"]
        d["mfma_mops_f64_per_leapfrog"] = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) / st
        d["wave_cycle_shares"] = {k: c[k] / c["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU") if k in c}
    if "SQ_BUSY_CU_CYCLES" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        # SQ_VALU_MFMA_BUSY_CYCLES sums the SIMDs' busy cycles (64 per v_mfma_f64_16x16x4_f64), SQ_BUSY_CU_CYCLES the CUs' cycles with work
        d["mfma_pipe_busy_share_of_cu_busy_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"])
        d["cu_busy_cycles_per_cu_over_elapsed_GHz"] = c["SQ_BUSY_CU_CYCLES"] / 256.0 / us / 1e3      # = shader clock while every CU is busy
    if "SQ_INSTS_VALU" in c:
        d["valu_insts_per_leapfrog"] = c["SQ_INSTS_VALU"] / st
        d["lds_insts_per_leapfrog"] = c.get("SQ_INSTS_LDS", 0) / st
    return {"counters_per_launch": c, "derived": d}
res = {"command": "tools/profile_dense_nuts_pmc.sh: rocprofv3 --kernel-trace --stats | --pmc <group> -- python3 tools/ubench/dense_nuts_pmc_run.py "
                  "(16 384 chains x D=256 dense MVN, eps 0.05: 5 single-transition launches then one of 20 transitions)",
       "corrections": "FETCH_SIZE x2 (gfx950 counts 128-B requests of wide coalesced reads as 64 B), WRITE_SIZE x1; both KiB",
       "launch_durations_us": dur, "leapfrog_steps_per_launch": steps,
       "single_transition_launches": part([2, 3, 4]), "twenty_transitions_per_launch": part([5])}
json.dump(res, open(out_path, "w"), indent=1)
for k in ("single_transition_launches", "twenty_transitions_per_launch"):
    print(k, json.dumps(res[k]["derived"]))
