#!/usr/bin/env python3
"""Print one line per kernel: VGPR/AGPR/SGPR/scratch/LDS/occupancy (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:.*?)Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
    if m and cur:
        cur[m.group(1).strip()] = m.group(2)
        if m.group(1).strip().startswith("LDS Size"):
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"idhmc::", "", name)
            name = re.sub(r"\(.*", "", name)
            if flt in name:
                print(f"{name[:70]:70s} V{cur.get('VGPRs','?'):>4} A{cur.get('AGPRs','?'):>4} S{cur.get('TotalSGPRs','?'):>4} "
                      f"scratch {cur.get('ScratchSize','?'):>5} occ {cur.get('Occupancy','?')} lds {cur.get('LDS Size','?')}")
            cur = {}
