"""The built library must be free of the two hardware hazards the compiler cannot guard inside inline-asm stores
(vmlmf_amd/csrc/vmlmf_device.h; tools/check_asm_hazards.py disassembles every gfx950 kernel and looks for them).
Runs on the CPU: it reads the code objects, it does not execute them."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "vmlmf_amd", "lib", "libvmlmf_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not present")
def test_no_unguarded_hazard_at_an_asm_store():
    if not os.path.exists(LIB):
        import vmlmf_amd._lib as L
        L.build()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_hazards.py"), LIB], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    first = r.stdout.splitlines()[0]
    assert "0 hazard(s)" in first and int(first.split()[2]) > 1000, first   # it did look at the stores ...
    assert int(first.split(";")[1].split()[0]) > 1000, first                # ... and at the DPP rank reduces


def test_the_checker_finds_a_planted_hazard(tmp_path):
    """The walk over a kernel's instructions, on hand-written listings: a reload of the base right in front of a store, a
    write of a wide store's data register right behind it, and the same two with enough wait states in between."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_asm_hazards as C
    bad_a = ["v_readlane_b32 s29, v184, 14", "global_store_dwordx4 v8, v[4:7], s[28:29] sc0 sc1", "s_nop 1"]
    bad_b = ["s_nop 4", "global_store_dwordx4 v17, v[8:11], s[16:17]", "s_cbranch_vccnz L1", "v_add_u32_e32 v8, 0x800, v16"]
    good = ["v_readlane_b32 s29, v184, 14", "s_nop 4", "global_store_dwordx4 v8, v[4:7], s[28:29]", "s_nop 1", "v_add_u32_e32 v4, 1, v4",
            "v_readlane_b32 s31, v1, 2", "s_mov_b32 s0, 0", "s_mov_b32 s0, 0", "s_mov_b32 s0, 0", "s_mov_b32 s0, 0", "s_mov_b32 s0, 0",
            "global_store_dword v14, v4, s[30:31]", "v_mov_b32_e32 v4, 0"]
    bad_c = ["v_mul_f32_e32 v5, v1, v2", "v_fmac_f32_dpp v9, v5, v7 row_ror:1 row_mask:0xf bank_mask:0xf"]
    good_c = ["v_mul_f32_e32 v5, v1, v2", "s_nop 1", "v_fmac_f32_dpp v9, v5, v7 row_ror:1 row_mask:0xf bank_mask:0xf",
              "v_fmac_f32_dpp v9, v5, v8 row_ror:2 row_mask:0xf bank_mask:0xf"]
    for lines, n in ((bad_a, 1), (bad_b, 1), (good, 0), (bad_c, 1), (good_c, 0)):
        problems = []
        C.check_kernel("k", lines, problems)
        assert len(problems) == n, (lines, problems)


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not present")
def test_headline_kernels_stay_inside_their_register_budget():
    """tools/kernel_budget.py: VGPR counts and scratch instructions of the recurrent kernels of configs A and C, read from the
    built library's code objects.  Their speed moves with the register allocation (+-5 % for unrelated
    edits); a spill into a per-step loop or a crossed occupancy step fails here instead of waiting for a re-measurement."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_budget.py"), LIB], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]
    assert "0 breach(es)" in r.stdout
