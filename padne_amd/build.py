"""Build libpadne_hip.so (gfx950) in-tree with hipcc.  ``python -m padne_amd.build``."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpadne_hip.so")
SOURCES = ["capi.hip", "spmv.hip", "spmm.hip", "pcg.hip", "assemble.hip", "comm.hip", "amg.hip", "generate.hip", "kkt.hip"]
HEADERS = ["common.hpp", os.path.join("..", "..", "include", "padne_hip.h"),
           os.path.join("..", "..", "include", "padne_hip_test.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libpadne_hip.so cannot be built")
    return exe


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    cc = hipcc()
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, "build", src.replace(".hip", ".o"))
        cmd = [cc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs, "-ldl",
           "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
