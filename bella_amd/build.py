"""Builds libbella_hip.so (hand-written gfx950 HIP kernels + the C ABI) in-tree with hipcc."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbella_hip.so")
SOURCES = ["bella_hip.hip"]
HEADERS = ["core.hpp", "util.hpp", "assemble.hpp", "kcount.hpp", "fastq.hpp", "wide.hpp", "spgemm.hpp", "order.hpp", "slotorder.hpp", "writer.hpp", "xdrop.hpp", "xdrop_packed.hpp", "logan.hpp", "comm.hpp", os.path.join("..", "..", "include", "bella_hip.h")]


CLI = os.path.join(HERE, "bin", "bella-hip")
CLI_SOURCES = [os.path.join(HERE, "host", f) for f in ("bella_hip_main.cpp", "bella_hip_driver.hpp")] + [os.path.join(HERE, "..", "include", "bella_hip.h")]


def build_cli(force: bool = False) -> str:
    """bella_amd/bin/bella-hip: the native command line (host/bella_hip_main.cpp), plain g++ against the C ABI; finds the library next
    to itself (rpath $ORIGIN/..)."""
    if not force and os.path.exists(CLI) and all(os.path.getmtime(f) <= os.path.getmtime(CLI) for f in CLI_SOURCES + [LIB]):
        return CLI
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(HERE, "..", "include"), "-I" + os.path.join(HERE, "host"),
                           "-o", CLI, CLI_SOURCES[0], "-L" + HERE, "-lbella_hip", "-Wl,-rpath,$ORIGIN/..", "-lpthread"])
    return CLI


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        build_cli()
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-Wno-unused-result", "-Wno-pass-failed", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    build_cli(force=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
