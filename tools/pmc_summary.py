"""per-kernel averages of arbitrary rocprofv3 PMC counters (one or more passes):

    rocprofv3 --kernel-trace --pmc C1 C2 ... --output-format csv -d DIR -- python3 bench.py ...
    python tools/pmc_summary.py out.txt DIR [DIR2 ...]

Every counter is averaged per launch of a kernel; "us" is the launch duration of the pass that collected the counter."""
import csv, glob, os, re, sys


def main():
    out, roots = sys.argv[1], sys.argv[2:]
    per, counters = {}, []
    for root in roots:
        for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                nm = re.sub(r"\(.*$", "", re.sub(r"^void\s+", "", row["Kernel_Name"]).replace("lwp::", ""))
                key = row["Counter_Name"]
                if key not in counters:
                    counters.append(key)
                d = per.setdefault(nm, {})
                c = d.setdefault(key, [0.0, 0, 0.0])
                c[0] += float(row["Counter_Value"]); c[1] += 1
                c[2] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    lines = ["%-46s %8s %9s " % ("kernel", "launches", "avg us") + " ".join("%22s" % c[-22:] for c in counters)]
    order = sorted(per.items(), key=lambda kv: -max(v[2] for v in kv[1].values()))
    for nm, d in order:
        any_c = next(iter(d.values()))
        lines.append("%-46s %8d %9.1f " % (nm[:46], any_c[1], any_c[2] / any_c[1] / 1e3) +
                     " ".join("%22.4g" % (d[c][0] / d[c][1]) if c in d else "%22s" % "-" for c in counters))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
