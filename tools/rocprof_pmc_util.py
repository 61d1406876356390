"""Merge rocprofv3 --pmc passes of SQ / GRBM counters (rocpd .db files) into profiles/<name>.json: per library kernel the
mean counter value per dispatch and the utilisation figures derived from them (north_star: "rocprof-reported ... MFMA
utilisation").

    python tools/rocprof_pmc_util.py out.json "<provenance>" pass1.db [pass2.db ...]

Derived (MI355X_MICROARCH.md: SQ_*_CYCLES except SQ_VALU_MFMA_BUSY_CYCLES count quad-cycles, summed over waves / SEs):
  valu_insts_per_wave, mfma_insts_per_wave, lds_insts_per_wave
  mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)        matrix pipe busy while the CU is busy
  valu_active_frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES                      share of wave time issuing VALU
  wait_frac        = SQ_WAIT_ANY / SQ_WAVE_CYCLES                              share parked on s_waitcnt / s_barrier
  lds_conflict_frac= SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (or / SQ_ACTIVE_INST_LDS)
"""
import json
import sqlite3
import sys


def short_name(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("<")[0].split("::")[-1].strip()


def tables(c):
    return [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]


def per_kernel(db):
    c = sqlite3.connect(db)
    out = {}
    view = "counters_collection"
    if view not in tables(c):
        raise SystemExit(f"{db}: no {view} view (tables: {tables(c)[:8]} ...)")
    q = (f"select kernel_name, counter_name, avg(s), count(*) from (select kernel_name, counter_name, dispatch_id, sum(value) as s "
         f"from {view} group by kernel_name, counter_name, dispatch_id) group by kernel_name, counter_name")
    for name, counter, v, n in c.execute(q):
        k = short_name(name)
        out.setdefault(k, {})[counter] = v
        out[k]["_dispatches"] = n
    return out


def main(out, source, *dbs):
    merged = {}
    for db in dbs:
        for k, d in per_kernel(db).items():
            merged.setdefault(k, {}).update(d)
    keep = {}
    for k, d in sorted(merged.items()):
        if not k.endswith("_kernel"):
            continue
        g = lambda n: d.get(n)
        der = {}
        waves = g("SQ_WAVES")
        if waves:
            for cn, dn in (("SQ_INSTS_VALU", "valu_insts_per_wave"), ("SQ_INSTS_MFMA", "mfma_insts_per_wave"),
                           ("SQ_INSTS_VALU_MFMA_MOPS_F32", "mfma_mops_f32_per_wave"), ("SQ_INSTS_LDS", "lds_insts_per_wave"),
                           ("SQ_INSTS_SALU", "salu_insts_per_wave")):
                if g(cn) is not None:
                    der[dn] = round(g(cn) / waves, 1)
        if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("SQ_BUSY_CU_CYCLES"):
            der["mfma_busy_frac"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * g("SQ_BUSY_CU_CYCLES")), 4)
        wc = g("SQ_WAVE_CYCLES")
        if wc:
            for cn, dn in (("SQ_ACTIVE_INST_VALU", "valu_active_frac"), ("SQ_WAIT_ANY", "wait_frac"),
                           ("SQ_WAIT_INST_ANY", "issue_stall_frac"), ("SQ_ACTIVE_INST_ANY", "active_frac"),
                           ("SQ_ACTIVE_INST_LDS", "lds_active_frac")):
                if g(cn) is not None:
                    der[dn] = round(g(cn) / wc, 4)
        den = g("SQ_LDS_IDX_ACTIVE") or g("SQ_ACTIVE_INST_LDS")
        if g("SQ_LDS_BANK_CONFLICT") is not None and den:
            der["lds_conflict_frac"] = round(g("SQ_LDS_BANK_CONFLICT") / den, 4)
        keep[k] = {"counters": {n: round(v, 1) for n, v in d.items() if not n.startswith("_")},
                   "dispatches": d.get("_dispatches"), "derived": der}
    doc = {"source": source,
           "units": "counter values are means per dispatch, summed over all SEs/XCDs; SQ cycle counters are quad-cycles "
                    "(x4 = shader cycles) except SQ_VALU_MFMA_BUSY_CYCLES (cycles); see tools/rocprof_pmc_util.py for the derived fields",
           "kernels": keep}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: v["derived"] for k, v in keep.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], *sys.argv[3:])
