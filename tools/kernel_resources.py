#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of a HIP source (hipcc -Rpass-analysis=kernel-resource-usage), build host only.
usage: tools/kernel_resources.py csrc/rq_scan.hip [name filter]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--offload-device-only", "-c", src,
                    "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None; rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    k, _, v = m.group(1).partition(": ")
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}; rows.append(cur)
    elif cur is not None: cur[k.strip()] = v.strip()
for c in rows:
    if flt in c["name"]:
        print(f'{c["name"][:90]:90s} VGPR {str(c.get("VGPRs")):>4} AGPR {str(c.get("AGPRs")):>3} SGPR {str(c.get("SGPRs")):>3} scratch {str(c.get("ScratchSize [bytes/lane]")):>4} spillV {str(c.get("VGPRs Spill")):>3} occ {str(c.get("Occupancy [waves/SIMD]"))}')
