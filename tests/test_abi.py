"""The C-ABI library: every symbol declared in include/sedumi_hip.h is exported by libsedumi_hip.so, and the
product loader fails loudly instead of falling back to anything (no compute calls here: no GPU needed)."""
import ctypes
import os
import re
import subprocess

import pytest

from helpers import ROOT

HEADER = os.path.join(ROOT, "include", "sedumi_hip.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sdm_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def hiplib():
    from sedumi_amd import build
    return build.build()


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for name in ("sdm_getada1", "sdm_getada2", "sdm_getada3", "sdm_blkchol", "sdm_fwblkslv", "sdm_bwblkslv",
                 "sdm_ordmmd", "sdm_symfct", "sdm_choltmpsiz", "sdm_cholsplit", "sdm_plan_create", "sdm_plan_getada",
                 "sdm_plan_blkchol", "sdm_plan_ldlsolve"):
        assert name in syms


def test_library_builds_for_gfx950_and_exports_every_symbol(hiplib):
    lib = ctypes.CDLL(hiplib)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in include/sedumi_hip.h but not exported: {missing}"
    lib.sdm_backend.restype = ctypes.c_char_p
    assert lib.sdm_backend() == b"hip-gfx950"


def test_code_object_targets_gfx950_and_uses_fp64_mfma(hiplib, tmp_path):
    """Every device code object of the library is a gfx950 one, and the trailing updates use the FP64 matrix instruction."""
    import shutil
    lib = shutil.copy(hiplib, tmp_path / "lib.so")                   # (the extraction writes next to its input)
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", str(lib)], capture_output=True, text=True, cwd=tmp_path).stdout
    assert "gfx950" in out
    devs = [f for f in os.listdir(tmp_path) if "amdgcn" in f]
    assert devs and all("gfx950" in f for f in devs), devs
    dis = _disassemble_gfx950(hiplib, tmp_path)
    assert dis.count("v_mfma_f64_16x16x4") > 100


def test_panel_launch_publishes_only_after_its_stores_are_acknowledged(hiplib, tmp_path):
    """Inter-workgroup hand-overs inside k_ldl_panel: write-through (sc1) data stores, then an explicit
    `s_waitcnt vmcnt(0)` (SDM_STORES_DONE -- a workgroup-scope fence emits nothing for global stores on gfx950), then
    the relaxed counter increment.  Disassembly check: between every signal add and the nearest write-through store
    before it there is a vmcnt(0) wait."""
    import shutil
    lib = shutil.copy(hiplib, tmp_path / "lib.so")
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    subprocess.run([objdump, "--offloading", str(lib)], capture_output=True, text=True, cwd=tmp_path)
    body = None
    for f in sorted(os.listdir(tmp_path)):
        if "gfx950" not in f:
            continue
        dis = subprocess.run([objdump, "-d", str(tmp_path / f)], capture_output=True, text=True).stdout
        # the kernel and the device functions its workgroup roles live in (panel_role_diag is a real call)
        parts = re.findall(r"^[0-9a-f]+ <[^>]*(?:k_ldl_panel|panel_role_|panel_stage_)[^>]*>:\n(.*?)(?=^[0-9a-f]+ <[^>]*>:|\Z)", dis, flags=re.S | re.M)
        if parts:
            body = [l for part in parts for l in part.split("\n") + ["<function boundary>"]]
            break
    assert body, "k_ldl_panel not found in the gfx950 code objects"
    signals = [i for i, l in enumerate(body) if re.search(r"\bglobal_atomic_add\b", l)]
    assert len(signals) >= 4
    checked = 0
    for i in signals:
        j, waited = i - 1, False
        while j >= 0 and not re.search(r"\bglobal_store\w* .*\bsc1\b", body[j]):
            waited = waited or "s_waitcnt vmcnt(0)" in body[j]
            if re.search(r"\bglobal_atomic_add\b", body[j]) or body[j] == "<function boundary>":
                j = -1                      # an earlier signal (or another function) lies in between: nothing new to publish here
                break
            j -= 1
        if j >= 0:
            assert waited, f"signal add at disassembly line {i} can overtake the write-through store at line {j}"
            checked += 1
    assert checked >= 2


def test_loader_fails_loudly_without_library(tmp_path):
    from sedumi_amd import capi
    capi.use_library(str(tmp_path / "nope.so"))
    try:
        with pytest.raises(capi.SdmError):
            capi.lib()
    finally:
        capi.use_library(None)


def test_no_device_is_an_error_not_a_fallback(hiplib):
    """In the GPU-less container plan creation must fail with a clear message."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    from sedumi_amd import capi
    from sedumi_amd.plan import Plan
    capi.use_library(None)
    with pytest.raises(capi.SdmError):
        Plan(0)


def _disassemble_gfx950(hiplib, tmp_path):
    import shutil
    lib = shutil.copy(hiplib, tmp_path / "lib.so")
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    subprocess.run([objdump, "--offloading", str(lib)], capture_output=True, text=True, cwd=tmp_path)
    return "\n".join(subprocess.run([objdump, "-d", str(tmp_path / f)], capture_output=True, text=True).stdout
                     for f in sorted(os.listdir(tmp_path)) if "gfx950" in f)       # one code object per source file


def test_no_kernel_spills_vector_registers(hiplib, tmp_path):
    """Code-object metadata of every gfx950 kernel in the library (llvm-readelf --notes on the embedded code objects,
    tools/code_objects.py): no kernel spills VGPRs.  k_ldl_panel did (167, round 2) until its workgroup roles became separate
    register allocations.  ONE exception, pinned down function by function in the test below: the blocked row solve that
    workgroup 0 of k_ldl_panel runs on the rows of a partial last block (panel_stage_rows_blocked) keeps five addresses in
    scratch across its 16-row batches -- 3 spilled VGPRs in the kernel's metadata."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("code_objects", os.path.join(ROOT, "tools", "code_objects.py"))
    co = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(co)
    if not os.path.exists(co.READELF):
        pytest.skip("llvm-readelf not found")
    ks = co.kernels(hiplib)
    assert len(ks) >= 40 and any("k_ldl_front" in k for k in ks) and any("k_sfw_diag" in k for k in ks)
    spilling = {k: v["vgpr_spill_count"] for k, v in ks.items() if v["vgpr_spill_count"]}
    # (round 6: 4 -- the kernel's ownership argument of the block-cyclic ranks made its body park one more register around the call of the
    # diagonal role; MAXCUT-4000's factor measured the same 2.465 ms before and after, profiles/r08j_*)
    assert all("k_ldl_panel" in k and n <= 4 for k, n in spilling.items()), f"kernels that spill vector registers: {spilling}"
    scratch = sorted(k for k, v in ks.items() if v["private_segment_fixed_size"])
    assert all("k_ldl_front" in k or "k_ldl_panel" in k for k in scratch), scratch


def test_merged_sweep_kernels_fit_eight_workgroups_per_cu(hiplib):
    """k_sfw_rows_diag / k_sbw_step_diag carry a streaming role that lives on eight workgroups of 256 per CU: at most 64 vector registers, no
    scratch, no spills (with the shared diagonal body called instead of a lean role the kernels were allotted 216 registers
    and a merged launch took twice the time of the two it replaced)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("code_objects", os.path.join(ROOT, "tools", "code_objects.py"))
    co = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(co)
    if not os.path.exists(co.READELF):
        pytest.skip("llvm-readelf not found")
    ks = {k: v for k, v in co.kernels(hiplib).items() if "k_sfw_rows_diag" in k or "k_sbw_step_diag" in k}
    assert len(ks) == 2, list(ks)
    for k, v in ks.items():
        assert v["vgpr_count"] <= 64 and v["agpr_count"] == 0 and v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (k, v)


def test_called_stages_of_the_factor_kernels_touch_scratch_only_at_entry(hiplib, tmp_path):
    """The stages of the two factor kernels are called functions (one register allocation each) that use most of the register
    file.  Under the default convention such a function saves and restores every callee-saved VGPR it touches through scratch
    on each call -- 72 in the diagonal-block stage, 117 in the rows stage (round 3: 2 us per panel of MAXCUT-4000's factor,
    0.7 us per panel of control07's).  Internal functions whose calls are not marked as tail calls are exempt (the caller
    keeps what it needs in registers the callee leaves alone): SDM_NOINLINE carries not_tail_called for that.  And a caller
    that needs something AFTER such a call has to park it in scratch around the call -- so the diagonal role of k_ldl_panel
    is a chain of stages that never return (update -> LDL' -> rows), each call the last thing its caller does.
    Checked in the disassembly: every stage has at most the handful of scratch instructions of its entry (the register that
    holds spilled SGPRs), the chain functions of k_ldl_front none at all; the one stage that does spill is named."""
    dis = _disassemble_gfx950(hiplib, tmp_path)
    found = {}
    names = ("front_diag", "front_rows_diag", "front_rows", "front_update", "panel_role_diag", "panel_stage_update", "panel_stage_block",
             "panel_stage_rows", "panel_stage_rows_few", "panel_stage_rows_blocked", "k_ldl_panel", "k_ldl_front")
    for name in names:
        for sym, part in re.findall(r"^[0-9a-f]+ <([^>]*sdm\d+%s[A-Z][^>]*)>:\n(.*?)(?=^[0-9a-f]+ <[^>]*>:|\Z)" % name, dis, flags=re.S | re.M):
            found[name] = len(re.findall(r"\bscratch_(?:load|store)", part))
    assert set(found) == set(names), found
    assert found["front_rows_diag"] == 0 and found["front_rows"] == 0 and found["front_update"] == 0 and found["k_ldl_front"] == 0, found
    assert found["panel_stage_rows_few"] == 0 and found["panel_stage_rows_blocked"] <= 12, found
    assert all(found[n] <= 4 for n in ("front_diag", "panel_role_diag", "panel_stage_update", "panel_stage_block", "panel_stage_rows")), found
    assert found["k_ldl_panel"] <= 6, found          # (the kernel body: three registers parked around the one call of the diagonal role, see the spill test above)
    # and nothing else in the library touches scratch at all (solves, ADA', dense columns, PSD, PCG: every kernel and every function)
    allowed = set(names) | {"pivot_probe"}
    for sym, part in re.findall(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^[0-9a-f]+ <[^>]*>:|\Z)", dis, flags=re.S | re.M):
        if re.search(r"\bscratch_(?:load|store)", part):
            assert any(re.search(r"sdm\d+%s[A-Z]" % n, sym) for n in allowed), sym
