"""Builds pygcn_amd/csrc/libgcn_spmm.so (the C-ABI of include/gcn_spmm.h) for gfx950 with hipcc.

The library is built IN-TREE so that it travels with a snapshot of the repository; hipcc
cross-compiles without a GPU.  `python -m pygcn_amd.build` or `__graft_entry__.build()`.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRCS = [os.path.join(HERE, "csrc", f) for f in ("gcn_spmm.hip", "gcn_ingest.hip", "gcn_gemm.hip", "gcn_plan.hip", "gcn_pack.hip")]
OUT = os.path.join(HERE, "csrc", "libgcn_spmm.so")
ARCH = "gfx950"


def needs_build():
    if not os.path.exists(OUT):
        return True
    deps = SRCS + [os.path.join(ROOT, "include", "gcn_spmm.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps)


def build(force=False, verbose=True):
    """One hipcc process per source file (in parallel), then one link."""
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # (-Wno-inline-asm: the DMA helper of gcn_gemm.hip names m0 in its clobber list on purpose — one
    #  warning per inlined copy, a thousand per build)
    flags = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-Wno-inline-asm", "-I",
             os.path.join(ROOT, "include")]
    flags[0:0] = os.environ.get("PYGCN_HIPCC_FLAGS", "").split()    # tuning experiments only
    objdir = os.path.join(HERE, "csrc", "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SRCS:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        jobs.append((obj, cmd, subprocess.Popen(cmd)))
    for obj, cmd, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    link = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", OUT + ".tmp"] + [j[0] for j in jobs]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
