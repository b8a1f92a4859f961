"""MFMA utilisation per kernel from one rocprofv3 PMC pass:

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d DIR -- python3 bench.py ...
    python tools/pmc_mfma.py DIR out.txt "label"

SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over the chip's 1024 SIMDs (MI355X_MICROARCH.md,
per-instruction constants); GRBM_GUI_ACTIVE counts the cycles the dispatch was active (summed over the 8 XCDs, hence
/ 8).  utilisation = busy / (1024 * active / 8).  The second figure prices the same kernel from its duration in the
kernel trace at 2.4 GHz."""
import csv
import glob
import os
import re
import sys

CLOCK_GHZ, SIMDS, XCDS = 2.4, 1024, 8


def main():
    root, out, label = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    per = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            nm = re.sub(r"^void\s+", "", row["Kernel_Name"]).replace("lwp::", "")
            nm = re.sub(r"\(.*$", "", nm)
            d = per.setdefault(nm, {"mfma": 0.0, "active": 0.0, "n": {}, "ns": 0.0})
            key = row["Counter_Name"]
            v = float(row["Counter_Value"])
            if key == "SQ_VALU_MFMA_BUSY_CYCLES":
                d["mfma"] += v
                d["ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            elif key == "GRBM_GUI_ACTIVE":
                d["active"] += v
            d["n"][key] = d["n"].get(key, 0) + 1
    lines = ["%s\nrocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; utilisation = MFMA busy cycles / (1024 SIMDs x active cycles)" % label,
             "%-44s %9s %12s %14s %14s" % ("kernel", "launches", "avg us", "util (GUI)", "util (2.4GHz)")]
    for nm, d in sorted(per.items(), key=lambda kv: -kv[1]["mfma"]):
        n = max(d["n"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0), 1)
        if d["mfma"] <= 0:
            continue
        u1 = d["mfma"] / (SIMDS * d["active"] / XCDS) if d["active"] > 0 else float("nan")
        u2 = d["mfma"] / (SIMDS * d["ns"] * CLOCK_GHZ) if d["ns"] > 0 else float("nan")
        lines.append("%-44s %9d %12.2f %13.1f%% %13.1f%%" % (nm[:44], n, d["ns"] / n / 1e3, 100 * u1, 100 * u2))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
