#!/bin/bash
# r02zz4: numbers after the weight-gradient workers moved into the backward launch and the asm-store hazards were closed:
# full bench, every config, strong-scaling points on one GPU, kernel stats, PMC counters of the same command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02zz4; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
timeout 300 python bench.py --repack --no-cpu-baseline --no-extra > $O/bench_repack.json 2>/dev/null < /dev/null
for gb in 512 256 128; do timeout 200 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null < /dev/null; done > $O/bench_strong_1gpu.jsonl
timeout 900 python tools/bench_configs.py > $O/configs.jsonl 2>/dev/null < /dev/null
timeout 600 python tools/bench_lm.py > $O/lm.jsonl 2>/dev/null < /dev/null
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
run_pmc() { local name=$1; shift; local ctr=$1; shift; ( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $ctr -d $R/$O/$name -o p -- "$@" ) > $O/$name.log 2>&1 < /dev/null; }
BENCH="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extra"
run_pmc a1 "$P1" $BENCH; run_pmc a2 "$P2" $BENCH; run_pmc af "FETCH_SIZE" $BENCH; run_pmc aw "WRITE_SIZE" $BENCH
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocprof_pmc_util.py $O/r02_zz4_pmc_util.json "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes) -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extra; config A (B=64 T=128 H=180 r=16), weight-gradient workers inside rec_bwd_kernel" $(db a1) $(db a2) > /dev/null 2>&1
python tools/rocprof_pmc.py $(db af) $(db aw) $O/r02_zz4_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extra; config A, weight-gradient workers inside rec_bwd_kernel; merged by tools/rocprof_pmc.py" > /dev/null 2>&1
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ks -o k -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra ) > $O/ks.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db ks) $O/r02_zz4_kernel_stats.csv "bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra (config A; eager region + hipGraph replays + untimed breakdown pass), weight-gradient workers inside rec_bwd_kernel: rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
rm -rf $O/a1 $O/a2 $O/af $O/aw $O/ks
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02zz4/bench.json")); r=d["roofline"]
print("bench", d["value"], d["ms_per_step"], d["eager_ms_per_step"], d.get("train_step_ms"), r["kernel"], r["achieved"], r["frac"], r["launch_us"], d["cpu_baseline"]["value"], d.get("speedup_vs_cpu"))
print("other", d.get("other_configs"))
for f in ("bench_repack.json","bench_strong_1gpu.jsonl","configs.jsonl","lm.jsonl"):
    for l in open("gpurun_out/r02zz4/"+f):
        l=l.strip()
        if l.startswith("{"):
            j=json.loads(l); print(f, {k:j[k] for k in j if k in ("config","shape","B","ms_per_step","ms_hipgraph","ms_eager","value","train_step_ms","fused_loss_and_update")})
PY
head -12 $O/r02_zz4_kernel_stats.csv | cut -c1-110
python -c "
import json
for f in ('r02_zz4_pmc_traffic.json','r02_zz4_pmc_util.json'):
    d=json.load(open('gpurun_out/r02zz4/'+f)); k=d['kernels']
    for n in ('rec_fwd_kernel','rec_bwd_kernel'):
        print(f, n, json.dumps(k.get(n))[:400])
"
