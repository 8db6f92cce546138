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


MEXHOST_DIR = os.path.join(HERE, "mexhost")
MEXHOST_LIB = os.path.join(LIBDIR, "libsdm_mexhost.so")
SHIM_DIR = os.path.join(HERE, "mexshims")
MEX_OUT = os.path.join(LIBDIR, "mex")
SHIMS = ["getada", "getada1", "getada2", "getada3", "blkchol", "fwblkslv", "bwblkslv", "ordmmdmex", "symfctmex", "choltmpsiz", "cholsplit",
         "symbfwblk", "finsymbden", "dpr1fact", "fwdpr1", "bwdpr1", "invcholfac", "incorder", "adendotd", "adenscale"]


def _newer(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


def build_mexhost(verbose=True):
    """sedumi_amd/lib/libsdm_mexhost.so: the MEX C API for running mexFunction binaries without MATLAB / Octave (gcc, no GPU code)."""
    os.makedirs(LIBDIR, exist_ok=True)
    src = os.path.join(MEXHOST_DIR, "mexhost.c")
    if _newer(MEXHOST_LIB, [src, os.path.join(MEXHOST_DIR, "mex.h")]):
        cmd = ["gcc", "-O2", "-fPIC", "-DNDEBUG", "-w", "-I", MEXHOST_DIR, "-shared", "-o", MEXHOST_LIB, src, "-lm"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return MEXHOST_LIB


def build_mexshims(lib=LIB, out=MEX_OUT, verbose=False):
    """Every mexFunction shim of sedumi_amd/mexshims as a shared object of its own (<out>/<name>.so), linked to the C-ABI library
    `lib` and to the MEX host -- g++ only: MATLAB / Octave are not needed (with them: mex / mkoctfile)."""
    from concurrent.futures import ThreadPoolExecutor
    build_mexhost(verbose)
    os.makedirs(out, exist_ok=True)
    inc = ["-I", MEXHOST_DIR, "-I", os.path.join(HERE, "..", "include"), "-I", SHIM_DIR]
    hdrs = [os.path.join(SHIM_DIR, "mexcommon.h"), os.path.join(MEXHOST_DIR, "mex.h"), os.path.join(HERE, "..", "include", "sedumi_hip.h")]
    common = os.path.join(out, "mexcommon.o")
    if _newer(common, [os.path.join(SHIM_DIR, "mexcommon.cpp")] + hdrs):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-Wall", "-c", os.path.join(SHIM_DIR, "mexcommon.cpp"), "-o", common] + inc)
    jobs = []
    for name in SHIMS:
        so = os.path.join(out, name + ".so")
        if _newer(so, [os.path.join(SHIM_DIR, name + ".cpp"), common, lib, MEXHOST_LIB] + hdrs):
            libdir, libname = os.path.dirname(os.path.abspath(lib)), os.path.basename(lib)[3:-3]     # (linked by name: found through the run path)
            jobs.append(["g++", "-O2", "-fPIC", "-Wall", "-shared", os.path.join(SHIM_DIR, name + ".cpp"), common, "-o", so, "-L", libdir, "-l" + libname,
                         "-L", LIBDIR, "-lsdm_mexhost", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + libdir, "-Wl,-rpath," + LIBDIR] + inc)
    if verbose:
        for j in jobs:
            print(" ".join(j), flush=True)
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(subprocess.check_call, jobs))
    return out


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 into sedumi_amd/lib/libsedumi_hip.so; then the MEX host and the mexFunction shims
    (sedumi_amd/lib/mex/<name>.so) linked to it."""
    if not force and not needs_build():
        build_mexshims(LIB, MEX_OUT, False)
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
    build_mexshims(LIB, MEX_OUT, False)
    return LIB


if __name__ == "__main__":
    if "--phases" in sys.argv:
        build_phases()
    elif "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], sys.argv[i + 2:])
    else:
        build(force="--force" in sys.argv)
