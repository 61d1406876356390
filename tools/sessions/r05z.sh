#!/bin/bash
# r05z: round-4 numbers: full bench, strong-scaling points on one GPU, every config, LM step, config E lines, kernel stats, PMC counters
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05z; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
for gb in 512 256 128; do timeout 200 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null < /dev/null; done > $O/bench_strong_1gpu.jsonl
for gb in 512 256 128; do VMLMF_INROW=0 timeout 200 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null < /dev/null; done > $O/bench_strong_1gpu_inrow_off.jsonl
timeout 900 python tools/bench_configs.py > $O/configs.jsonl 2>/dev/null < /dev/null
timeout 600 python tools/bench_lm.py > $O/lm.jsonl 2>/dev/null < /dev/null
timeout 600 python tools/bench_lm.py 256 --dropout 0.5 > $O/lm_dropout.jsonl 2>/dev/null < /dev/null
timeout 300 python bench.py --force-collective --steps 100 --warmup 10 --no-cpu-baseline --no-extra > $O/bench_forced_collective_1rank.json 2>/dev/null < /dev/null
timeout 600 python bench.py --config E --force-collective --steps 20 --warmup 5 > $O/config_e_lm_1gpu_forced_collective.json 2>/dev/null < /dev/null
for bp in 128 64; do timeout 600 python bench.py --config E --batch-per-gpu $bp --steps 20 --warmup 5 2>/dev/null < /dev/null; done > $O/config_e_lm_1gpu_b128_b64.jsonl
timeout 600 python bench.py --config E --steps 20 --warmup 5 > $O/config_e_lm_1gpu.json 2>/dev/null < /dev/null
timeout 600 python bench.py --config E --batch-per-gpu 32 --steps 20 --warmup 5 > $O/config_e_lm_1gpu_b32.json 2>/dev/null < /dev/null
VMLMF_BENCH_REHEARSAL=1 timeout 600 python bench.py --config E --gpus 2 --steps 10 --warmup 3 > $O/rehearsal_config_e_2ranks.json 2>/dev/null < /dev/null
VMLMF_BENCH_REHEARSAL=1 timeout 600 python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline --no-extra > $O/rehearsal_plain2.json 2>/dev/null < /dev/null
VMLMF_BENCH_REHEARSAL=1 timeout 600 python bench.py --gpus 2 --global-batch 512 --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $O/rehearsal_strong2.json 2>/dev/null < /dev/null
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
run_pmc() { local name=$1; shift; local ctr=$1; shift; ( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $ctr -d $R/$O/$name -o p -- "$@" ) > $O/$name.log 2>&1 < /dev/null; }
db() { find $O/$1 -name "*.db" | head -1; }
BENCH="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extra"
run_pmc a1 "$P1" $BENCH; run_pmc a2 "$P2" $BENCH; run_pmc af "FETCH_SIZE" $BENCH; run_pmc aw "WRITE_SIZE" $BENCH
python tools/rocprof_pmc_util.py $O/r05_pmc_util.json "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes) -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extra; config A (B=64 T=128 H=180 r=16), round 5 (rec_fwd_kernel, rec3_bwd_kernel with the riding workers)" $(db a1) $(db a2) > /dev/null 2>&1
python tools/rocprof_pmc.py $(db af) $(db aw) $O/r05_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extra; config A, round 5; merged by tools/rocprof_pmc.py" > /dev/null 2>&1
rm -rf $O/a1 $O/a2 $O/af $O/aw
# the same at 256 rows per GPU (BASELINE configs[3] at N = 2): rec4_bwd_kernel, weight gradients inside the rows' workgroups - and the form it replaces
B256="python3 $R/bench.py --global-batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extra"
run_pmc b1 "$P1" $B256; run_pmc b2 "$P2" $B256; run_pmc bf "FETCH_SIZE" $B256; run_pmc bw "WRITE_SIZE" $B256
python tools/rocprof_pmc_util.py $O/r05_pmc_util_b256.json "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes) -- python3 bench.py --global-batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extra; 256 rows on one GPU, round 5 (rec_fwd_kernel, rec4_bwd_kernel)" $(db b1) $(db b2) > /dev/null 2>&1
python tools/rocprof_pmc.py $(db bf) $(db bw) $O/r05_pmc_traffic_b256.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --global-batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extra; 256 rows on one GPU, round 5 (rec4_bwd_kernel: no dpre tape, no weight-gradient launch)" > /dev/null 2>&1
rm -rf $O/b1 $O/b2 $O/bf $O/bw
export VMLMF_INROW=0
run_pmc cf "FETCH_SIZE" $B256; run_pmc cw "WRITE_SIZE" $B256
python tools/rocprof_pmc.py $(db cf) $(db cw) $O/r05_pmc_traffic_b256_inrow_off.json "the same command with VMLMF_INROW=0 (round 3's form: rec3_bwd_kernel writes dpre, wgrad_mfma_kernel reads it back)" > /dev/null 2>&1
unset VMLMF_INROW
rm -rf $O/cf $O/cw
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ks -o k -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra ) > $O/ks.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db ks) $O/r05_kernel_stats.csv "bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra (config A; eager region + hipGraph replays + untimed breakdown pass), round 5: rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ksb -o k -- python3 $R/bench.py --global-batch 256 --steps 50 --warmup 10 --no-cpu-baseline --no-extra ) > $O/ksb.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db ksb) $O/r05_kernel_stats_b256.csv "bench.py --global-batch 256 --steps 50 --warmup 10 --no-cpu-baseline --no-extra (256 rows on one GPU: rec4_bwd_kernel), round 5: rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/kse -o k -- python3 $R/tools/run_e.py --nograph ) > $O/kse.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db kse) $O/r05_config_e_layer_kernel_stats.csv "config E layer (V4 group, H=650, ranks 32/[32,32], B=256, T=35), round 5: rocprofv3 --kernel-trace --stats -- python3 tools/run_e.py --nograph" > /dev/null 2>&1
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ksl -o k -- python3 $R/tools/bench_lm.py ) > $O/ksl.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db ksl) $O/r05_lm_step_kernel_stats.csv "tools/bench_lm.py: whole LM steps at config E's shape (13 steps each of: head in place V3 / group, two-call loss, stock, group), round 5: rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ksd -o k -- python3 $R/tools/bench_lm.py 256 --dropout 0.5 ) > $O/ksd.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db ksd) $O/r05_lm_step_dropout_kernel_stats.csv "tools/bench_lm.py 256 --dropout 0.5: 13 LM steps each at p = 0, at p = 0.5 with the package's mask-free dropout, at p = 0.5 with nn.Dropout (Model.stock_dropout), group layers, round 5: rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
rm -rf $O/ksd
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ksc -o k -- python3 $R/tools/run_c.py ) > $O/ksc.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db ksc) $O/r05_config_c_kernel_stats.csv "config C (2 x 256, rank 24, B 128, T 24, I 77, fp32) through the wavefront launches, round 5: rocprofv3 --kernel-trace --stats -- python3 tools/run_c.py" > /dev/null 2>&1
rm -rf $O/ks $O/ksb $O/kse $O/ksl $O/ksc
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05z/bench.json")); r=d["roofline"]
print("bench", d["value"], d["ms_per_step"], d.get("ms_per_step_kept_images"), d["eager_ms_per_step"], d.get("train_step_ms"), r["kernel"], r["achieved"], r["frac"], r["launch_us"], r.get("traffic"), d["cpu_baseline"]["value"], d.get("speedup_vs_cpu"))
print("other", d.get("other_configs"))
for f in ("bench_strong_1gpu.jsonl","bench_strong_1gpu_inrow_off.jsonl","configs.jsonl","lm.jsonl","lm_dropout.jsonl","bench_forced_collective_1rank.json","config_e_lm_1gpu_forced_collective.json","config_e_lm_1gpu_b128_b64.jsonl","config_e_lm_1gpu.json","config_e_lm_1gpu_b32.json","rehearsal_config_e_2ranks.json","rehearsal_plain2.json","rehearsal_strong2.json"):
    try:
        for l in open("gpurun_out/r05z/"+f):
            l=l.strip()
            if l.startswith("{"):
                j=json.loads(l); print(f, {k:j[k] for k in j if k in ("config","shape","B","ms_per_step","ms_hipgraph","ms_eager","value","train_step_ms","fused_loss_and_update","head_in_place","ms_per_step_eager","dropout","dropout_launches","ms_p0","ms_package","ms_nn_dropout","allreduce_alone_ms")} if "workload" not in str(j.get("config")) else (j["config"].get("batch_per_gpu"), j["ms_per_step"], j.get("train_step_ms"), j["n_gpus"]))
    except OSError as e:
        print(f, "missing")
for f in ("r05_pmc_traffic.json","r05_pmc_traffic_b256.json","r05_pmc_traffic_b256_inrow_off.json"):
    try:
        k=json.load(open("gpurun_out/r05z/"+f))["kernels"]; print(f, {n:round(v["hbm_bytes_per_launch"]/1e6,1) for n,v in k.items()})
    except Exception as e:
        print(f, "missing", e)
PY
head -12 $O/r05_kernel_stats_b256.csv | cut -c1-110
