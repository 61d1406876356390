#!/bin/bash
# r02z: final numbers of the round: full bench, configs, LM step, strong-scaling points on one GPU, kernel stats + PMC
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02z; mkdir -p $O
R=$GRAFT_REPO_ROOT
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --repack --no-cpu-baseline > $O/bench_repack.json 2>/dev/null
for gb in 512 256 128; do python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null; done > $O/bench_strong_1gpu.jsonl
python tools/bench_configs.py > $O/configs.jsonl 2>/dev/null
python tools/bench_lm.py > $O/lm.jsonl 2>/dev/null
python tools/bench_rb.py e32 > $O/e32.jsonl 2>/dev/null
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
run_pmc() { local name=$1; shift; local ctr=$1; shift; ( cd /tmp && rocprofv3 --kernel-trace --pmc $ctr -d $R/$O/$name -o p -- "$@" ) > $O/$name.log 2>&1; }
BENCH="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph"
run_pmc a1 "$P1" $BENCH; run_pmc a2 "$P2" $BENCH; run_pmc af "FETCH_SIZE" $BENCH; run_pmc aw "WRITE_SIZE" $BENCH
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocprof_pmc_util.py $O/r02_pmc_util.json "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes) -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph; config A (B=64 T=128 H=180 r=16)" $(db a1) $(db a2) > /dev/null 2>&1
python tools/rocprof_pmc.py $(db af) $(db aw) $O/r02_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph; config A; merged by tools/rocprof_pmc.py" > /dev/null 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/ks -o k -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline ) > $O/ks.log 2>&1
python tools/rocprof_summary.py $(db ks) $O/r02_kernel_stats.csv "bench.py --steps 50 --warmup 10 --no-cpu-baseline (config A; eager region + hipGraph replays + untimed breakdown pass): rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/ke -o k -- python3 $R/tools/run_e.py --nograph ) > $O/ke.log 2>&1
python tools/rocprof_summary.py $(db ke) $O/r02_config_e_layer_kernel_stats.csv "config E layer (V4 group, H=650, ranks 32/[32,32], B=256, T=35), clusters of 16 on the row-block kernels: rocprofv3 --kernel-trace --stats -- python3 tools/run_e.py --nograph" > /dev/null 2>&1
rm -rf $O/a1 $O/a2 $O/af $O/aw $O/ks $O/ke
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02z/bench.json")); print("bench", d["value"], d["ms_per_step"], d["eager_ms_per_step"], d["train_step_ms"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["speedup_vs_cpu"])
for f in ("bench_strong_1gpu.jsonl","configs.jsonl","lm.jsonl","e32.jsonl"):
    for l in open("gpurun_out/r02z/"+f):
        l=l.strip()
        if l.startswith("{"):
            j=json.loads(l); print(f, {k:j[k] for k in j if k in ("config","shape","B","ms_per_step","ms_hipgraph","ms_eager","ms_fwd_bwd","ms_per_step_eager","fused_loss_and_update")} if "metric" not in j else (j["config"]["batch_per_gpu"], j["ms_per_step"]))
PY
head -14 $O/r02_kernel_stats.csv
