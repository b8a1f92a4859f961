"""Summarise rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected in SEPARATE runs with --kernel-trace only) into
per-kernel and per-class HBM traffic per launch.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc/fetch -- python3 bench.py --streams 1 ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc/write -- python3 bench.py --streams 1 ...
    python tools/pmc_traffic.py gpurun_out/pmc/fetch gpurun_out/pmc/write out.json "batch=1 368x656 fp32"

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM): FETCH_SIZE and WRITE_SIZE are in KB;
on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide (16 B/lane) coalesced reads, so it is DOUBLED here
(an upper estimate for narrower accesses); WRITE_SIZE is exact for 16-B-per-lane stores.  Infinity-Cache hits are
counted, so this is fabric-side traffic of the XCD L2s (an upper bound on HBM bytes)."""
import csv
import glob
import json
import os
import re
import sys


def kernel_class(name):
    if "stem" in name:
        return "stem"
    if "dwpw" in name:
        return "fused_dw_pw"
    if "heads_" in name:                           # a stage's two 1x1 head convs in one launch
        return "gemm_1x1"
    if "gemm" in name:
        m = re.search(r"<([^>]*)>", name)
        args = [a.strip() for a in m.group(1).split(",")] if m else []
        if "gemm_bf16_ar" in name:                 # window-resident bf16 kernel: the dense 3x3 convs at large M
            return "dense_3x3"
        if "bf16" in name:                         # shared-tile bf16 kernel: the 1x1 convs (and 3x3 at small M)
            return "gemm_1x1"
        return "dense_3x3" if args and args[-1] == "3" else "gemm_1x1"
    if "dw_kernel" in name or "dw_tiled" in name:
        return "depthwise"
    if any(k in name for k in ("find_peaks", "nms", "score_pairs", "match", "assemble", "publish", "preprocess", "upsample", "resize")):
        return "post"
    return "other"


def collect(root, counter):
    per = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            nm = re.sub(r"^void\s+", "", row["Kernel_Name"]).replace("lwp::", "")
            nm = re.sub(r"\(.*$", "", nm)
            a = per.setdefault(nm, [0.0, 0])
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return per


def main():
    fetch_dir, write_dir, out, note = sys.argv[1], sys.argv[2], sys.argv[3], (sys.argv[4] if len(sys.argv) > 4 else "")
    fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    kernels, classes = {}, {}
    for nm in sorted(set(fe) | set(wr)):
        f_kb, f_n = fe.get(nm, [0.0, 0])
        w_kb, w_n = wr.get(nm, [0.0, 0])
        n = max(f_n, w_n, 1)
        k = {"class": kernel_class(nm), "launches_profiled": n,
             "fetch_bytes_per_launch_raw": f_kb * 1024 / max(f_n, 1), "fetch_bytes_per_launch": 2 * f_kb * 1024 / max(f_n, 1),
             "write_bytes_per_launch": w_kb * 1024 / max(w_n, 1)}
        k["traffic_bytes_per_launch"] = k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]
        kernels[nm] = k
        c = classes.setdefault(k["class"], {"launches_profiled": 0, "fetch": 0.0, "write": 0.0})
        c["launches_profiled"] += n
        c["fetch"] += 2 * f_kb * 1024 * (n / max(f_n, 1)) if f_n else 0.0
        c["write"] += w_kb * 1024 * (n / max(w_n, 1)) if w_n else 0.0
    for c in classes.values():
        c["traffic_bytes_per_launch"] = (c.pop("fetch") + c.pop("write")) / max(c["launches_profiled"], 1)
    json.dump({"note": note, "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes); FETCH_SIZE x2 (gfx950), KB -> bytes",
               "classes": classes, "kernels": kernels}, open(out, "w"), indent=1)
    print(json.dumps(classes, indent=1))


if __name__ == "__main__":
    main()
