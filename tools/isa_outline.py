"""outline of one kernel's ISA: MFMA / global loads / LDS reads / waitcnts / branches in program order.
usage: isa_outline.py file.s mangled_name_substring [start_label]"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"))
out = []
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"): out.append("END"); break
    if t.startswith("v_mfma"): out.append("M")
    elif t.startswith("s_waitcnt"): out.append("[" + t[9:].strip() + "]")
    elif t.startswith("global_load") or t.startswith("buffer_load"): out.append("G")
    elif t.startswith("global_store") or t.startswith("buffer_store"): out.append("S")
    elif t.startswith("ds_read"): out.append("D")
    elif t.startswith("ds_write"): out.append("W")
    elif t.startswith("scratch_"): out.append("SCR")
    elif t.startswith("s_cbranch") or t.startswith("s_branch"): out.append("<" + t.split()[0][2:] + " " + t.split()[1] + ">")
    elif re.match(r"^\.LBB\d+_\d+:", t): out.append("\n" + t)
    elif t.startswith("s_barrier"): out.append("BAR")
txt = " ".join(out)
txt = re.sub(r"(?:M ){4,}", lambda m: "M*%d " % (len(m.group(0)) // 2), txt)
print(txt)
