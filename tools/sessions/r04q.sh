#!/bin/bash
# r04q: PMC passes of a config E layer with wgrad_ring_kernel (roofline row: counter bytes, MFMA busy), the bench line of the tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04q2; mkdir -p $O
R=$GRAFT_REPO_ROOT
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
run_pmc() { local name=$1; shift; local ctr=$1; shift; ( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $ctr -d $R/$O/$name -o p -- "$@" ) > $O/$name.log 2>&1 < /dev/null; }
db() { find $O/$1 -name "*.db" | head -1; }
E1="python3 $R/tools/run_e.py --nograph"
run_pmc e1 "$P1" $E1; run_pmc e2 "$P2" $E1; run_pmc ef "FETCH_SIZE" $E1; run_pmc ew "WRITE_SIZE" $E1
python tools/rocprof_pmc_util.py $O/r04_pmc_util_config_e.json "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes) -- python3 tools/run_e.py --nograph; one config E layer (V4 group, H 650, ranks 32/[32,32], B 256, T 35), round 4, with wgrad_ring_kernel" $(db e1) $(db e2) > /dev/null 2>&1
python tools/rocprof_pmc.py $(db ef) $(db ew) $O/r04_pmc_traffic_config_e.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/run_e.py --nograph; one config E layer, round 4, with wgrad_ring_kernel" > /dev/null 2>&1
rm -rf $O/e1 $O/e2 $O/ef $O/ew
python - <<'PY'
import json
for f in ("r04_pmc_traffic_config_e.json","r04_pmc_util_config_e.json"):
    try:
        k=json.load(open("gpurun_out/r04q2/"+f))["kernels"]
        for n,v in k.items():
            print(f, n, v.get("hbm_bytes_per_launch"), json.dumps(v.get("derived"))[:300] if "derived" in v else "")
    except Exception as e:
        print(f, "missing", e)
PY
timeout 900 python bench.py > $O/r04_bench.json 2> $O/bench.err < /dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04q2/r04_bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("bench", d["value"], d["ms_per_step"], d.get("ms_per_step_kept_images"), d["eager_ms_per_step"], d.get("train_step_ms"), r["kernel"], r["frac"], d["kernels_us"], {k:v.get("ms_per_step") for k,v in d["other_configs"].items()})
PY
