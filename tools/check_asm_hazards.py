#!/usr/bin/env python3
"""Static check of the built library for the two gfx940+ hazards that hipcc cannot guard inside inline-asm blocks
(vmlmf_amd/csrc/vmlmf_device.h, "Inline asm: invisible to ..."):

  A. a vector memory instruction that takes its address from scalar registers needs five wait states after a VALU
     instruction wrote those registers (v_readlane / v_readfirstlane / v_cmp ... -> s[N:N+1]);
  B. a store of more than 8 bytes needs two wait states before a VALU instruction overwrites its data registers;
  C. a DPP instruction (v_fmac_f32_dpp ... row_ror, the rank reduce of the recurrent kernels, also inline asm) needs two
     wait states after a VALU instruction wrote the register it reads across lanes.

The compiler inserts the wait states for the memory instructions it emits itself; the stores written as inline asm
(scalar base + 32-bit vector offset: st4_sv, st1_sv and their write-through variants) are invisible to it.  This script
disassembles every gfx950 code object of the library and walks each kernel linearly: for every global_store with a
scalar base it looks back five wait states for a VALU write of the base pair (A) and, for dwordx3/x4 stores, forward two
wait states for a VALU write of a data register (B).  s_nop N counts N + 1 wait states, every other instruction one;
a branch target in between is treated as "unknown" and ends the search in that direction (labels only occur between
blocks; the stores sit in straight-line code behind their LDS reads).

Usage: check_asm_hazards.py [path/to/libvmlmf_hip.so]      exit status 1 when a hazard is found.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
STORE = re.compile(r"^\s*global_store_(dword|dwordx2|dwordx3|dwordx4|short|byte)\s+(v\d+|v\[\d+:\d+\]),\s*(v\d+|v\[\d+:\d+\]),\s*(s\[(\d+):(\d+)\])")
SALU_OR_OTHER = ("s_", "ds_", "global_", "buffer_", "flat_", "scratch_", "v_nop")


def regs(tok):
    m = re.match(r"([sva])\[(\d+):(\d+)\]", tok)
    if m:
        return {m.group(1) + str(i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([sva])(\d+)$", tok)
    return {tok} if m else set()


def wait_states(ins):
    m = re.match(r"\s*s_nop\s+(\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def dest_regs(ins):
    """registers a VALU instruction writes (first operand; v_cmp*/v_readlane*/v_readfirstlane may write scalars)"""
    s = ins.strip()
    if not s.startswith("v_"):
        return set()
    ops = s.split(None, 1)
    if len(ops) < 2:
        return set()
    first = ops[1].split(",")[0].strip()
    out = regs(first)
    if first == "vcc":
        out = {"vcc"}
    # v_cmpx / carry-out forms: second operand can be a scalar pair too
    parts = [p.strip() for p in ops[1].split(",")]
    if s.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_div_scale", "v_mad_u64", "v_mad_i64")) and len(parts) > 1:
        out |= regs(parts[1])
    return out


DPP = re.compile(r"^\s*(v_\w+_dpp)\s+(v\d+),\s*(v\d+)")


def check_kernel(name, lines, problems):
    n = len(lines)
    for i, ins in enumerate(lines):
        d = DPP.match(ins)
        if d:   # C: a DPP instruction reads its first source from other lanes: two wait states after a VALU write of it
            src, ws, j = d.group(3), 0, i - 1
            while j >= 0 and ws < 2:
                prev = lines[j]
                if prev.endswith(":"):
                    break
                if src in dest_regs(prev):
                    problems.append(f"{name}: VALU write of {src} {ws} wait state(s) before the DPP read `{ins.strip()}`: `{prev.strip()}`")
                    break
                ws += wait_states(prev)
                j -= 1
            continue
        m = STORE.match(ins)
        if not m:
            continue
        base = {"s" + str(k) for k in range(int(m.group(5)), int(m.group(6)) + 1)}
        data = regs(m.group(3))
        wide = m.group(1) in ("dwordx3", "dwordx4")
        # A: look back five wait states
        ws, j = 0, i - 1
        while j >= 0 and ws < 5:
            prev = lines[j]
            if prev.endswith(":"):     # label: another path joins here
                break
            if dest_regs(prev) & base:
                problems.append(f"{name}: VALU write of the scalar base {m.group(4)} {ws} wait state(s) before `{ins.strip()}`: `{prev.strip()}`")
                break
            ws += wait_states(prev)
            j -= 1
        # B: look forward two wait states
        if wide:
            ws, j = 0, i + 1
            while j < n and ws < 2:
                nxt = lines[j]
                if nxt.endswith(":"):
                    break
                if dest_regs(nxt) & data:
                    problems.append(f"{name}: VALU write of a data register {ws} wait state(s) behind `{ins.strip()}`: `{nxt.strip()}`")
                    break
                ws += wait_states(nxt)
                j += 1


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="vmlmf_hz_")
    try:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib, copy)
        subprocess.run([OBJDUMP, "--offloading", copy], check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" in f:
                r = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--no-leading-addr", os.path.join(tmp, f)], check=True, capture_output=True, text=True)
                yield f, r.stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main(lib):
    problems, kernels, stores, dpps = [], 0, 0, 0
    for _, text in disassemble(lib):
        name, cur = None, []
        for raw in text.splitlines():
            line = raw.split("//")[0].rstrip()
            if not line.strip():
                continue
            m = re.match(r"^<?([A-Za-z_$][\w$.]*)>?:$", line.strip())
            if m and not line.startswith((" ", "\t")):
                if name is not None:
                    kernels += 1
                    stores += sum(1 for x in cur if STORE.match(x))
                    dpps += sum(1 for x in cur if DPP.match(x))
                    check_kernel(name, cur, problems)
                name, cur = m.group(1), []
            elif name is not None:
                s = line.strip()
                if re.match(r"^<[\w$.]+>:$", s):          # local label inside a function
                    cur.append(s[1:-2] + ":")
                else:
                    cur.append(s)
        if name is not None:
            kernels += 1
            stores += sum(1 for x in cur if STORE.match(x))
            dpps += sum(1 for x in cur if DPP.match(x))
            check_kernel(name, cur, problems)
    print(f"{kernels} functions, {stores} stores with a scalar base checked, {len(problems)} hazard(s); {dpps} DPP reads checked")
    for p in problems[:50]:
        print("  " + p)
    return 1 if problems else 0


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "vmlmf_amd", "lib", "libvmlmf_hip.so")))
