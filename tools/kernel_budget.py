#!/usr/bin/env python3
"""Register / scratch budget of the kernels on the headline path, read from the built library's code-object metadata
(llvm-readelf --notes of every gfx950 code object): no GPU needed.

Why: the recurrent kernels are bound by instruction issue on a dependent chain, and their speed moves by several per cent with
the register allocation ("0.164 <-> 0.172 ms for edits that only touched the worker half").  A scalar or vector
register spilled into scratch inside a per-step loop, or a VGPR count that crosses an occupancy step (168 / 256), is a silent
regression; this list makes it a failing CPU test instead of something only a re-measurement on the GPU finds.

    python tools/kernel_budget.py [libvmlmf_hip.so]        prints every kernel's (vgpr, sgpr, scratch); exit 1 on a breach
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
DEMANGLE = shutil.which("c++filt") or "c++filt"

# kernel (demangled prefix) -> (max VGPRs, max scratch_load / scratch_store INSTRUCTIONS in its code: the metadata's
# private_segment_fixed_size also counts stack objects no instruction touches).  Config A = rec_fwd_kernel<16,1,false,256,3,true> (forward, with
# the x-projection wave: its callee-saved registers are the scratch) and rec3_bwd_kernel<16,1> (backward with the riding
# workers); the values are what round 3 shipped with a margin of one allocation granule (8 registers).
BUDGET = {
    "void rec_fwd_kernel<16, 1, false, 256, 3, true>(": (256, 0),     # (its x-projection wave is a callee of its own: below)
    "void rec_fwd_kernel<16, 1, false, 256, 3, false>(": (128, 0),
    "void rec3_bwd_kernel<16, 0>(": (136, 0),
    "void rec3_bwd_kernel<16, 1>(": (256, 0),      # (riding forms: one workgroup per CU by their LDS request; the direct prologue peaks at ~190)
    "void rec3_bwd_kernel<16, 2>(": (256, 0),
    "void rec3_bwd_kernel<8, 0>(": (136, 0),
    "void rec_bwd_kernel<16, 1, false, 256, 3, 0>(": (128, 0),
    "void wf_fwd_kernel<24, 4, 1, 256, false>(": (168, 0),
    "void wf_bwd_kernel<24, 4, 1, 256, false>(": (168, 0),
}


def kernels(lib):
    tmp = tempfile.mkdtemp(prefix="vmlmf_kb_")
    out = {}
    try:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib, copy)
        subprocess.run([OBJDUMP, "--offloading", copy], check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([READELF, "--notes", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            cur = {}
            for line in txt.splitlines():
                m = re.match(r"\s*\.(name|vgpr_count|sgpr_count|private_segment_fixed_size|agpr_count):\s*(\S+)", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2)
                if k == "name":
                    cur = {"name": v}
                    out[v] = cur
                else:
                    cur[k] = int(v)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # scratch instructions per kernel, from the disassembly (the hazard checker's walk over the same code objects)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import check_asm_hazards as C
    for _, text in C.disassemble(lib):
        name = None
        for raw in text.splitlines():
            line = raw.split("//")[0].rstrip()
            m = re.match(r"^<?([A-Za-z_$][\w$.]*)>?:$", line.strip())
            if m and not line.startswith((" ", "\t")):
                name = m.group(1)
            elif name in out and line.strip().startswith("scratch_"):
                out[name]["scratch_insts"] = out[name].get("scratch_insts", 0) + 1
    names = list(out)
    dem = subprocess.run([DEMANGLE], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return {d: out[n] for n, d in zip(names, dem)}


def main(lib):
    ks = kernels(lib)
    bad = []
    for prefix, (vmax, smax) in BUDGET.items():
        hit = [(d, k) for d, k in ks.items() if d.startswith(prefix)]
        if not hit:
            bad.append(f"{prefix}...: not in the library (renamed? update tools/kernel_budget.py)")
            continue
        for d, k in hit:
            v = k.get("vgpr_count", 0) + k.get("agpr_count", 0)
            s = k.get("scratch_insts", 0)
            line = (f"{prefix[5:-1]}: {v} VGPRs (budget {vmax}), {k.get('sgpr_count', 0)} SGPRs, {s} scratch instructions (budget {smax}; "
                    f"{k.get('private_segment_fixed_size', 0)} B private segment)")
            print(line)
            if v > vmax or s > smax:
                bad.append(line)
    print(f"{len(ks)} kernels in the library, {len(BUDGET)} budgeted, {len(bad)} breach(es)")
    for b in bad:
        print("  BREACH:", b)
    return 1 if bad else 0


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "vmlmf_amd", "lib", "libvmlmf_hip.so")))
