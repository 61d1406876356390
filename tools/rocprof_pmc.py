"""Merge two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; results .db of each) into profiles/<name>.json:
HBM bytes per launch of every library kernel, with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE
under-reports wide coalesced reads by exactly 2x; checked against the known tape sizes of rec_bwd_kernel).

    python tools/rocprof_pmc.py fetch.db write.db out.json "<provenance line>"
"""
import json
import sqlite3
import sys

OURS = ("pack_kernel", "xproj_kernel", "rec_fwd_kernel", "rec_bwd_kernel", "rec3_fwd_kernel", "rec3_bwd_kernel", "rec4_bwd_kernel", "reduce_cg_many_kernel", "nll_grad_kernel", "nll_finish_kernel", "embed_bwd_kernel", "adam_gate_kernel", "dqx_dx_kernel", "wgrad_mfma_kernel", "wgrad_ring_kernel",
        "reduce_cg_kernel", "finish2_kernel", "finish_kernel", "adam_fused_kernel", "head_fwd_kernel", "head_bwd_kernel", "ce_fwd_kernel", "ce_bwd_kernel",
        "adam_kernel", "wf_fwd_kernel", "wf_bwd_kernel", "pack_stack_kernel", "wgrad_mfma_stack_kernel", "wgrad4_stack_kernel",
        "reduce_cg_stack_kernel", "finish_stack_kernel", "rb_fwd_kernel", "rb_bwd_kernel", "rb_pack_kernel",
        "xexp_mfma_kernel", "gemm_skinny_kernel", "rbx_fwd_kernel", "rbx_bwd_kernel", "rbx_zero_kernel", "rb_pack_stack_kernel")


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    out = {}
    q = ("select kernel_name, avg(s) from (select kernel_name, dispatch_id, sum(value) as s from counters_collection "
         "where counter_name = ? group by kernel_name, dispatch_id) group by kernel_name")
    for name, v in c.execute(q, (counter,)):
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("<")[0].split("::")[-1].strip()
        if short in OURS:
            out[short] = v
    return out


def main(fdb, wdb, out, source):
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        fk, wk = f.get(k, 0.0), w.get(k, 0.0)
        kernels[k] = {"FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1),
                      "hbm_bytes_per_launch": int(round((2 * fk + wk) * 1024))}
    doc = {"source": source,
           "units": "counter values are KiB per dispatch (mean over dispatches, summed over XCDs); gfx950 correction "
                    "per MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by exactly 2x -> "
                    "read_bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE used as is",
           "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
