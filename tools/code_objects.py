"""tools/code_objects.py [library] -- register / scratch metadata of every gfx950 kernel in libsedumi_hip.so.
The shared object carries one clang offload bundle per HIP source; each bundle entry for gfx950 is an ELF code object whose
NT_AMDGPU_METADATA note lists, per kernel, .vgpr_count, .vgpr_spill_count, .sgpr_spill_count, .private_segment_fixed_size
(llvm-readelf --notes).  Used by tests/test_abi.py (no kernel may spill vector registers)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    """the gfx950 code objects (bytes) embedded in a host shared object"""
    blob = open(path, "rb").read()
    out, pos = [], 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return out
        nent = struct.unpack_from("<Q", blob, i + 24)[0]
        p = i + 32
        for _ in range(nent):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size > 0:
                out.append(blob[i + off:i + off + size])
        pos = i + 24


def kernels(path):
    """{kernel name: {vgpr_count, agpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size}}"""
    res = {}
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        for blk in re.split(r"\n\s*- \.agpr_count", "\n" + notes)[1:]:
            blk = "  - .agpr_count" + blk
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            rec = {}
            for key in ("vgpr_count", "agpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
                mm = re.search(r"\.%s:\s+(\d+)" % key, blk)
                rec[key] = int(mm.group(1)) if mm else 0
            res[name.group(1)] = rec
    return res


def demangle(names):
    import shutil
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not filt:
        return list(names)
    p = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.split("\n") if p.returncode == 0 else list(names)


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "sedumi_amd", "lib", "libsedumi_hip.so")
    ks = kernels(lib)
    names = sorted(ks)
    for n, dn in zip(names, demangle(names)):
        r = ks[n]
        print("%-44s vgpr %3d agpr %3d  vgpr spills %3d  sgpr spills %3d  scratch %4d B  lds %6d B" %
              (dn.split("(")[0][-44:], r["vgpr_count"], r["agpr_count"], r["vgpr_spill_count"], r["sgpr_spill_count"], r["private_segment_fixed_size"], r["group_segment_fixed_size"]))
