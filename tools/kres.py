"""kernel resource table from hipcc -Rpass-analysis=kernel-resource-usage output on stdin: name, VGPRs, spills, LDS, occupancy"""
import re, subprocess, sys
txt = sys.stdin.read()
cur = None
for l in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().replace("lwp::", "").split("(")[0]}
        continue
    for key, tag in (("VGPRs", "v"), ("VGPRs Spill", "vs"), ("SGPRs Spill", "ss"), ("Occupancy [waves/SIMD]", "occ")):
        m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", l)
        if m and cur is not None: cur[tag] = int(m.group(1))
    if "LDS Size" in l and cur is not None:
        if len(sys.argv) < 2 or sys.argv[1] in cur["name"]:
            print("%-60s vgpr %3d spill %3d sspill %3d occ %s" % (cur["name"][:60], cur.get("v", -1), cur.get("vs", -1), cur.get("ss", -1), cur.get("occ")))
        cur = None
