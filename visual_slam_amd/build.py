"""Builds libORBhip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libORBhip.so"
SOURCES = [CSRC / "orbhip.hip"]
DEPS = SOURCES + sorted(CSRC.glob("*.h")) + [PKG.parent / "include" / "orbhip.h", PKG.parent / "include" / "orb_pattern_31.inc"]
# -ffp-contract=off: float expressions (scale tables, fastAtan2, the rBRIEF rotation) must round
# exactly as written, on the host and on the device.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-result"]


def build(force=False, verbose=False, extra=()):
    if not force and LIB.exists() and all(LIB.stat().st_mtime >= d.stat().st_mtime for d in DEPS):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, *FLAGS, *extra, "-o", str(LIB), *map(str, SOURCES)]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="-f" in sys.argv, verbose=True,
          extra=["-Rpass-analysis=kernel-resource-usage"] if "-r" in sys.argv else ())
    print("built", LIB)
