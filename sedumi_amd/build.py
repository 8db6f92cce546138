"""Build libsedumi_hip.so (HIP, gfx950) in-tree.  `python -m sedumi_amd.build`"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsedumi_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(f) > t for f in deps)


def build_phases(verbose=True):
    """Tools-only variant with the in-kernel phase clocks compiled in (-DSDM_PHASES): sedumi_amd/lib/libsedumi_hip_phases.so.
    Never loaded by the package; tools/phase_*.py point capi.use_library at it."""
    os.makedirs(LIBDIR, exist_ok=True)
    out = os.path.join(LIBDIR, "libsedumi_hip_phases.so")
    hipcc, objs = _compile_objects(["-DSDM_PHASES", "-Wno-unused-variable"], os.path.join(LIBDIR, "obj_phases"), verbose)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_variant(tag, flags, verbose=True):
    """Tools-only build with extra compiler flags (a measurement variant): sedumi_amd/lib/libsedumi_hip_<tag>.so.
    Never loaded by the package; the tools point capi.use_library at it (SDM_LIB)."""
    os.makedirs(LIBDIR, exist_ok=True)
    out = os.path.join(LIBDIR, "libsedumi_hip_%s.so" % tag)
    hipcc, objs = _compile_objects(list(flags), os.path.join(LIBDIR, "obj_" + tag), verbose)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def _compile_objects(extra, objdir, verbose):
    """One object per source (hipcc -c), rebuilt only when the source or any header is newer; the compiles run side by side."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libsedumi_hip.so")
    os.makedirs(objdir, exist_ok=True)
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    hdr_t = max(os.path.getmtime(f) for f in hdrs)
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-Wall", "-Wno-unused-result"] + extra +
                        ["-I", CSRC, "-o", obj, src])
    if verbose:
        for j in jobs:
            print(" ".join(j), flush=True)
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(subprocess.check_call, jobs))
    return hipcc, objs


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 into sedumi_amd/lib/libsedumi_hip.so."""
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    if force and os.path.isdir(objdir):
        shutil.rmtree(objdir)
    hipcc, objs = _compile_objects([], objdir, verbose)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if "--phases" in sys.argv:
        build_phases()
    elif "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], sys.argv[i + 2:])
    else:
        build(force="--force" in sys.argv)
