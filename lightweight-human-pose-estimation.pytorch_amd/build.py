"""Build recipe for the C-ABI library (hipcc, gfx950 only, in-tree output).

    python lightweight-human-pose-estimation.pytorch_amd/build.py [--force]

Produces ``liblwpose_hip.so`` next to this file.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "liblwpose_hip.so")
SOURCES = ["net_graph.cpp", "net_kernels.hip", "net_kernels_bf16.hip", "net_kernels_tiled.hip", "post_kernels.hip", "capi.cpp"]
HEADERS = ["lwp_internal.h", os.path.join("..", "..", "include", "lwpose.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-result", "-DNDEBUG"]
# the post-processing must reproduce NumPy's separately-rounded float32/float64 arithmetic bit for bit:
# no FMA contraction anywhere in that translation unit (the in-source pragma alone is not honoured for
# packed-math fusion by hipcc 7.2)
EXTRA = {"post_kernels.hip": ["-ffp-contract=off"] + (["-DLWP_ASM_STAMPS"] if os.environ.get("LWP_ASM_STAMPS") else []),
         # LWP_ABLATION=1 at build time adds the ablation instantiations of the hot kernels (tools/ only; never shipped by default)
         "net_kernels_bf16.hip": (["-DLWP_ABLATION"] if os.environ.get("LWP_ABLATION") else []),
         "net_kernels.hip": (["-DLWP_ABLATION"] if os.environ.get("LWP_ABLATION") else []) +
                            (["-DDWPW_PF=%d" % int(os.environ["LWP_DWPW_PF"])] if os.environ.get("LWP_DWPW_PF") else [])}


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    hdr_t = max(hdr_t, os.path.getmtime(os.path.abspath(__file__)))
    # the ablation / experiment defines change the objects without touching a source file: such builds recompile everything
    special = any(os.environ.get(k) for k in ("LWP_ABLATION", "LWP_DWPW_PF", "LWP_ASM_STAMPS")) or os.path.exists(os.path.join(CSRC, ".special"))
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or special or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([hipcc] + FLAGS + EXTRA.get(s, []) + ["-x", "hip", "-c", src, "-o", obj])
    marker = os.path.join(CSRC, ".special")
    if any(os.environ.get(k) for k in ("LWP_ABLATION", "LWP_DWPW_PF", "LWP_ASM_STAMPS")):
        open(marker, "w").close()                      # the next plain build must recompile too
    elif os.path.exists(marker):
        os.remove(marker)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=min(len(jobs), 5) or 1) as ex:      # one hipcc per translation unit, side by side
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs)
    os.replace(LIB + ".tmp", LIB)                      # atomic: a snapshot of the tree never sees a half-written library
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
