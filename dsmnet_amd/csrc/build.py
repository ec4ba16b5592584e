"""Build libdsmnet_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m dsmnet_amd.csrc.build          # or: python dsmnet_amd/csrc/build.py

The library is built in-tree (next to the sources) so that it travels with the
repository snapshot; it is git-ignored (*.so).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["abi.hip", "corr1d.hip", "cost_volume.hip", "soft_argmin.hip", "conv3d.hip", "conv_f16.hip",
           "conv3d_bwd.hip", "bn3d.hip", "decoder.hip", "spp.hip", "warp.hip"]
LIB = os.path.join(HERE, "libdsmnet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-Wall", "-Wno-unused-function"] + os.environ.get("HIPCC_EXTRA", "").split()


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, stamps=False):
    """stamps=True: diagnostic build with in-kernel s_memtime phase stamps
    (libdsmnet_hip_stamps.so; never loaded by the package)."""
    if stamps:
        lib = os.path.join(HERE, "libdsmnet_hip_stamps.so")
        srcs = [os.path.join(HERE, s) for s in SOURCES]
        extra = ["-DDSM_ABLATE=%s" % os.environ["DSM_ABLATE"]] if os.environ.get("DSM_ABLATE") else []
        cmd = [HIPCC] + FLAGS + ["-DDSM_STAMPS"] + extra + ["-shared", "-o", lib] + srcs
        print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return lib
    hdrs = [os.path.join(HERE, "common.hpp"), os.path.join(HERE, "conv_split.hpp"), os.path.join(HERE, "conv_zs.hpp"), os.path.join(HERE, "conv_common.hpp"),
            os.path.join(HERE, "..", "..", "include", "dsmnet_hip.h")]
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    objs = []
    jobs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, stamps="--stamps" in sys.argv))
