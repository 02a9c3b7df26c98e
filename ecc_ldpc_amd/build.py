"""Builds ecc_ldpc_amd/libldpc_hip.so in-tree with hipcc for gfx950 (no JIT cache: the built .so
travels to the GPU box with the repository snapshot)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libldpc_hip.so")
CLI = os.path.join(HERE, "ecc-ldpc-hip")
# (cli_main.cc is the stand-alone executable, built separately below)
SOURCES = ["api.cc", "host.cc", "flood.hip", "fused.hip", "fused_msg.hip", "fused_split.hip", "fused_csr.hip", "sim.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function", "-Wno-unused-value",
         "-ffp-contract=off",  # parity: a*b+c must round twice, like the reference's Double/float_ty ops
         "-fno-fast-math"]


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "ldpc_hip.h"))
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = _headers()
    objs, procs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            # -save-temps=obj keeps the device assembly (<stem>-hip-amdgcn-amd-amdhsa-gfx950.s) next to the object:
            # the input of tools/isa_histogram.py (instruction histogram of the iteration loops -> build/isa_stats.json)
            cmd = [HIPCC] + FLAGS + (["-save-temps=obj"] if s.endswith(".hip") else []) + ["-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    if force or procs or _stale(OUT, objs):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    isa_stats(objdir, force or bool(procs))
    for f in os.listdir(objdir):   # the other -save-temps by-products are not needed by anything
        if f.endswith((".bc", ".hipi", ".out", ".resolution.txt", ".hipfb")) or (f.endswith(".s") and "-host-" in f) or f.endswith("gfx950.o"):
            os.remove(os.path.join(objdir, f))
    # the native command-line driver (host code only; binds the C ABI like any other C/C++ host would)
    cli_src = os.path.join(CSRC, "cli_main.cc")
    if force or _stale(CLI, [cli_src, OUT] + hdrs):
        cmd = [HIPCC, "-O2", "-std=c++17", "-Wall", "-x", "hip", cli_src, "-o", CLI, "-L" + HERE, "-lldpc_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


def isa_stats(objdir, force=False):
    """build/isa_stats.json: per kernel, the instruction histogram of its iteration loop (tools/isa_histogram.py) --
    what bench.py's VALU roofline multiplies by the iterations a run really executed."""
    import glob
    import importlib.util
    import json
    out = os.path.join(objdir, "isa_stats.json")
    asm = sorted(glob.glob(os.path.join(objdir, "fused*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    if not asm or not (force or _stale(out, asm)):
        return out
    tool = os.path.join(HERE, "..", "tools", "isa_histogram.py")
    if not os.path.exists(tool):
        return out
    spec = importlib.util.spec_from_file_location("isa_histogram", tool)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.run(asm)
    for r in res:
        r["source"] = os.path.basename(r["source"])
    json.dump(res, open(out, "w"), indent=1)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
