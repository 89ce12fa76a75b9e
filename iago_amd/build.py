"""Builds iago_amd/libiago_hip.so (the C-ABI library, gfx950 code objects) in-tree.

    python -m iago_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the
GPU box with the repository snapshot.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libiago_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + \
        glob.glob(os.path.join(HERE, "..", "include", "*.h")) + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link the shared library."""
    if not force and not _stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "_obj"), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, "_obj", os.path.basename(src) + ".o")
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", SO] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
