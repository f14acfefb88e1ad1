"""one line about the placement a bench run got (gpurun_out/pw.json, pw.err from IDHMC_PLACEMENT_VERBOSE=1)"""
import json, re
d = json.load(open("gpurun_out/pw.json")); sp = d["state_placement"]
err = open("gpurun_out/pw.err").read()
first = err.split("idhmc placement")[1:]
walk = [float(x) for x in re.findall(r"pair walk step \d+ \([\d.]+ GiB held\): [\d.]+ GB/s = ([\d.]+) x", err)]
print("%.4e frac %.4f | %s | probe %.0f single %.0f tried %d create %.0f ms peak %.1f GiB | first context's pair ratios: %s" % (
    d["value"], d["roofline"]["frac"], sp["kind"], sp["probe_GBps"], sp["single_array_GBps"], sp["candidates_tried"], sp["create_ms"],
    sp["peak_transient_bytes"] / 2**30, " ".join("%.2f" % w for w in walk[:10])))
