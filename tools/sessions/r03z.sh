#!/bin/bash
# r03z: round-3 numbers: full bench, strong-scaling points on one GPU, every config, LM step, kernel stats, PMC counters of the same command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03z; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
for gb in 512 256 128; do timeout 200 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null < /dev/null; done > $O/bench_strong_1gpu.jsonl
timeout 900 python tools/bench_configs.py > $O/configs.jsonl 2>/dev/null < /dev/null
timeout 600 python tools/bench_lm.py > $O/lm.jsonl 2>/dev/null < /dev/null
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
run_pmc() { local name=$1; shift; local ctr=$1; shift; ( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $ctr -d $R/$O/$name -o p -- "$@" ) > $O/$name.log 2>&1 < /dev/null; }
BENCH="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extra"
run_pmc a1 "$P1" $BENCH; run_pmc a2 "$P2" $BENCH; run_pmc af "FETCH_SIZE" $BENCH; run_pmc aw "WRITE_SIZE" $BENCH
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocprof_pmc_util.py $O/r03_pmc_util.json "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes) -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extra; config A (B=64 T=128 H=180 r=16), round 3 kernels (rec_fwd_kernel, rec3_bwd_kernel with the riding workers)" $(db a1) $(db a2) > /dev/null 2>&1
python tools/rocprof_pmc.py $(db af) $(db aw) $O/r03_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extra; config A, round 3 kernels; merged by tools/rocprof_pmc.py" > /dev/null 2>&1
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ks -o k -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra ) > $O/ks.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db ks) $O/r03_kernel_stats.csv "bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra (config A; eager region + hipGraph replays + untimed breakdown pass), round 3: rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/kse -o k -- python3 $R/tools/run_e.py --nograph ) > $O/kse.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db kse) $O/r03_config_e_layer_kernel_stats.csv "config E layer (V4 group, H=650, ranks 32/[32,32], B=256, T=35), round 3: rocprofv3 --kernel-trace --stats -- python3 tools/run_e.py --nograph" > /dev/null 2>&1
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ksc -o k -- python3 $R/tools/run_c.py ) > $O/ksc.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db ksc) $O/r03_config_c_kernel_stats.csv "config C (2 x 256, rank 24, B 128, T 24, I 77, fp32) through the wavefront launches, round 3: rocprofv3 --kernel-trace --stats -- python3 tools/run_c.py" > /dev/null 2>&1
rm -rf $O/a1 $O/a2 $O/af $O/aw $O/ks $O/kse $O/ksc
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03z/bench.json")); r=d["roofline"]
print("bench", d["value"], d["ms_per_step"], d.get("ms_per_step_kept_images"), d["eager_ms_per_step"], d.get("train_step_ms"), r["kernel"], r["achieved"], r["frac"], r["launch_us"], r.get("traffic"), d["cpu_baseline"]["value"], d.get("speedup_vs_cpu"))
print("other", d.get("other_configs"))
for f in ("bench_strong_1gpu.jsonl","configs.jsonl","lm.jsonl"):
    for l in open("gpurun_out/r03z/"+f):
        l=l.strip()
        if l.startswith("{"):
            j=json.loads(l); print(f, {k:j[k] for k in j if k in ("config","shape","B","ms_per_step","ms_hipgraph","ms_eager","value","train_step_ms","fused_loss_and_update")})
PY
head -12 $O/r03_kernel_stats.csv | cut -c1-110
