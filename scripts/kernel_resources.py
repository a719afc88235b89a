#!/usr/bin/env python3
"""Print a per-kernel resource table (VGPR/AGPR/SGPR/spill/occupancy/LDS) from hipcc remarks."""
import re, subprocess, sys
out = subprocess.run(["make", "-s", "-C", sys.argv[1] if len(sys.argv) > 1 else str(__import__("pathlib").Path(__file__).resolve().parent.parent / "lipvq-vae_amd" / "csrc"), "resources"],
                     capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: ([A-Za-z \[\]/]+): (.+?) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:70]}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'occ':>4s} {'LDS':>6s}")
for r in rows:
    print(f"{r['name']:70s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {r.get('VGPRs Spill','?'):>6s} {r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('LDS Size [bytes/block]','?'):>6s}")
