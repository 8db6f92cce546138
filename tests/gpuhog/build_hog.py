"""tests/gpuhog/build_hog.py -- TEST INFRASTRUCTURE: builds tests/gpuhog/libgpuhog.so (hipcc, gfx950), the kernel that occupies
compute units for the starvation test of the one-launch factor kernel.  `python tests/gpuhog/build_hog.py`; also run by
__graft_entry__.build(), so the library travels to the GPU box."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libgpuhog.so")


def build():
    src = os.path.join(HERE, "hog.hip")
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(src):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-o", LIB, src])
    return LIB


if __name__ == "__main__":
    print(build())
