"""tests/hipemu/build_emu.py -- TEST INFRASTRUCTURE ONLY.
Compiles the sedumi_amd/csrc kernel sources with g++ against the fiber emulator
(hipemu.h) into tests/hipemu/libsedumi_hipemu.so for CPU-side logic tests."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
CSRC = os.path.join(ROOT, "sedumi_amd", "csrc")
LIB = os.path.join(HERE, "libsedumi_hipemu.so")


def build(force=False):
    """(Serialised across processes: pytest-xdist workers that find the library stale at the same time would otherwise compile and link it on top of
    each other -- `file too short` in the worker that loads it meanwhile.)"""
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build(force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build(force=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "hipemu.*")) + \
        glob.glob(os.path.join(ROOT, "include", "*.h"))
    if not force and os.path.exists(LIB) and all(os.path.getmtime(f) <= os.path.getmtime(LIB) for f in deps):
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    objs, jobs = [], []
    hdr_t = max(os.path.getmtime(f) for f in deps if f.endswith(".h"))
    for s in srcs + [os.path.join(HERE, "hipemu.cpp")]:
        o = os.path.join(HERE, "_obj_" + os.path.basename(s) + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t):
            jobs.append(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-DSDM_EMU", "-x", "c++", "-I", HERE, "-I", CSRC,
                         "-Wall", "-Wno-unknown-pragmas", "-Wno-unused-variable", "-c", s, "-o", o])
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(subprocess.check_call, jobs))
    subprocess.check_call(["g++", "-shared", "-o", LIB + ".tmp"] + objs)
    os.replace(LIB + ".tmp", LIB)                      # (never a half-written library under the name the tests load)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
