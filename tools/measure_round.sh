#!/bin/bash
# Every number of a round's record (profiles/rNN_*), from the tree as it stands, on one MI355X box:
#   bash tools/measure_round.sh r06          (through gpurun: results land in gpurun_out/r06z, copy what is to be judged into profiles/)
# bench line, strong-scaling points on one GPU, every BASELINE config, LM steps, config E data-parallel lines (1 rank + 2-rank rehearsals),
# rocprofv3 kernel stats of the headline / 256 rows / config C / config E layer / the clustered stack at 32 rows / LM steps, and the
# PMC passes (SQ counters in two passes, FETCH_SIZE, WRITE_SIZE - each its own run, --kernel-trace only) for config A, 256 rows and
# config C and the config E kernels (clustered stack at 32 rows + the single layer at 256 rows).
RN=${1:-r06}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${RN}z; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
ks() { local name=$1; shift; local out=$1; shift; local title=$1; shift
  ( cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $R/$O/$name -o k -- "$@" ) > $O/$name.log 2>&1 < /dev/null
  python tools/rocprof_summary.py $(db $name) $O/$out "$title" > /dev/null 2>&1; rm -rf $O/$name; }
run_pmc() { local name=$1; shift; local ctr=$1; shift; ( cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --pmc $ctr -d $R/$O/$name -o p -- "$@" ) > $O/$name.log 2>&1 < /dev/null; }
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
pmc4() { local tag=$1; shift; local what=$1; shift   # four passes of one command -> ${RN}_pmc_util_$tag.json, ${RN}_pmc_traffic_$tag.json
  run_pmc ${tag}1 "$P1" "$@"; run_pmc ${tag}2 "$P2" "$@"; run_pmc ${tag}f "FETCH_SIZE" "$@"; run_pmc ${tag}w "WRITE_SIZE" "$@"
  python tools/rocprof_pmc_util.py $O/${RN}_pmc_util$tag.json "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes) -- $what" $(db ${tag}1) $(db ${tag}2) > /dev/null 2>&1
  python tools/rocprof_pmc.py $(db ${tag}f) $(db ${tag}w) $O/${RN}_pmc_traffic$tag.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) -- $what; merged by tools/rocprof_pmc.py" > /dev/null 2>&1
  rm -rf $O/${tag}1 $O/${tag}2 $O/${tag}f $O/${tag}w; }

timeout 900 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
for gb in 512 256 128; do timeout 200 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null < /dev/null; done > $O/bench_strong_1gpu.jsonl
timeout 900 python tools/bench_configs.py > $O/configs.jsonl 2>/dev/null < /dev/null
timeout 600 python tools/bench_lm.py > $O/lm.jsonl 2>/dev/null < /dev/null
timeout 600 python tools/bench_lm.py 32 > $O/lm_b32.jsonl 2>/dev/null < /dev/null
timeout 600 python tools/bench_lm.py 256 --dropout 0.5 > $O/lm_dropout.jsonl 2>/dev/null < /dev/null
timeout 600 python tools/bench_lm.py 32 --dropout 0.5 > $O/lm_dropout_b32.jsonl 2>/dev/null < /dev/null
timeout 300 python tools/probes/rbx_probe.py 32 64 128 > $O/config_e_two_layers_stacked_vs_chained.jsonl 2>/dev/null < /dev/null
timeout 300 python tools/probes/rbx_probe.py 32 64 128 --plain >> $O/config_e_two_layers_stacked_vs_chained.jsonl 2>/dev/null < /dev/null
timeout 300 python bench.py --force-collective --steps 100 --warmup 10 --no-cpu-baseline --no-extra > $O/bench_forced_collective_1rank.json 2>/dev/null < /dev/null
timeout 600 python bench.py --config E --steps 20 --warmup 5 > $O/config_e_lm_1gpu.json 2>/dev/null < /dev/null
for bp in 128 64 32; do timeout 600 python bench.py --config E --batch-per-gpu $bp --steps 20 --warmup 5 2>/dev/null < /dev/null; done > $O/config_e_lm_1gpu_b128_b64_b32.jsonl
VMLMF_BENCH_REHEARSAL=1 timeout 600 python bench.py --config E --gpus 2 --steps 10 --warmup 3 > $O/rehearsal_config_e_2ranks.json 2>/dev/null < /dev/null
VMLMF_BENCH_REHEARSAL=1 timeout 600 python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline --no-extra > $O/rehearsal_plain2.json 2>/dev/null < /dev/null
VMLMF_BENCH_REHEARSAL=1 timeout 600 python bench.py --gpus 2 --global-batch 512 --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $O/rehearsal_strong2.json 2>/dev/null < /dev/null
VMLMF_BENCH_REHEARSAL=1 timeout 600 python bench.py --gpus 2 --transport p2p --steps 50 --warmup 10 --no-cpu-baseline --no-extra > $O/rehearsal_p2p2.json 2>/dev/null < /dev/null
timeout 300 python tools/probes/p2p_latency.py 2>/dev/null | tail -1 > $O/p2p_latency_world1.txt

BENCH="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extra"
pmc4 "" "python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extra; config A (B=64 T=128 H=180 r=16), $RN" $BENCH
B256="python3 $R/bench.py --global-batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extra"
pmc4 "_b256" "python3 bench.py --global-batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extra; 256 rows on one GPU (rec_fwd_kernel, rec4_bwd_kernel), $RN" $B256
pmc4 "_config_e" "python3 tools/probes/run_e.py --nograph; one PTB group layer, B=256 T=35 (rb_fwd_kernel / rb_bwd_kernel on clusters of 16), $RN" python3 $R/tools/probes/run_e.py --nograph
pmc4 "_config_c" "python3 tools/probes/run_c.py; config C (2 x 256, rank 24, B 128, T 24, fp32: wf_fwd_kernel / wf_bwd_kernel / wgrad4_stack_kernel), $RN" python3 $R/tools/probes/run_c.py
pmc4 "_config_e_stack32" "python3 tools/probes/rbx_probe.py 32 --stacked-only; two PTB group layers at 32 rows in one launch per direction (rbx_fwd_kernel / rbx_bwd_kernel), $RN" python3 $R/tools/probes/rbx_probe.py 32 --stacked-only

ks ks ${RN}_kernel_stats.csv "bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra (config A; eager region + hipGraph replays + untimed breakdown pass), $RN: rocprofv3 --kernel-trace --stats" python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra
ks ksb ${RN}_kernel_stats_b256.csv "bench.py --global-batch 256 --steps 50 --warmup 10 --no-cpu-baseline --no-extra (256 rows on one GPU: rec4_bwd_kernel), $RN: rocprofv3 --kernel-trace --stats" python3 $R/bench.py --global-batch 256 --steps 50 --warmup 10 --no-cpu-baseline --no-extra
ks kse ${RN}_config_e_layer_kernel_stats.csv "config E layer (V4 group, H=650, ranks 32/[32,32], B=256, T=35), $RN: rocprofv3 --kernel-trace --stats -- python3 tools/probes/run_e.py --nograph" python3 $R/tools/probes/run_e.py --nograph
ks kss ${RN}_config_e_stack_b32_kernel_stats.csv "two PTB group layers at 32 rows, both in ONE launch per direction (csrc/vmlmf_rbx.hip), $RN: rocprofv3 --kernel-trace --stats -- python3 tools/probes/rbx_probe.py 32 --stacked-only" python3 $R/tools/probes/rbx_probe.py 32 --stacked-only
ks ksh ${RN}_config_e_chained_b32_kernel_stats.csv "the same two layers at 32 rows as chained per-layer launches (round 5's form), $RN: rocprofv3 --kernel-trace --stats -- python3 tools/probes/rbx_probe.py 32 --chained-only" python3 $R/tools/probes/rbx_probe.py 32 --chained-only
ks ksl ${RN}_lm_step_kernel_stats.csv "tools/bench_lm.py: whole LM steps at config E's shape (13 steps each of: head in place V3 / group, two-call loss, stock, group), $RN: rocprofv3 --kernel-trace --stats" python3 $R/tools/bench_lm.py
ks ksl32 ${RN}_lm_step_b32_kernel_stats.csv "tools/bench_lm.py 32: whole LM steps at 32 rows (configs[4]'s share of one GPU of eight), $RN: rocprofv3 --kernel-trace --stats" python3 $R/tools/bench_lm.py 32
ks ksc ${RN}_config_c_kernel_stats.csv "config C (2 x 256, rank 24, B 128, T 24, I 77, fp32) through the wavefront launches, $RN: rocprofv3 --kernel-trace --stats -- python3 tools/probes/run_c.py" python3 $R/tools/probes/run_c.py
python - "$O" <<'PY'
import json, sys, glob, os
O = sys.argv[1]
d = json.loads(open(O + "/bench.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("bench", d["value"], d["ms_per_step"], d.get("ms_per_step_kept_images"), d["eager_ms_per_step"], d.get("train_step_ms"), r["kernel"], r["achieved"], r["frac"], r["launch_us"], r.get("traffic"), d["cpu_baseline"]["value"], d.get("harness"))
print("other", {k: v.get("ms_per_step") for k, v in d.get("other_configs", {}).items()})
for f in sorted(glob.glob(O + "/*.json*")):
    if f.endswith("bench.json") or "pmc" in f: continue
    for l in open(f):
        l = l.strip()
        if l.startswith("{"):
            j = json.loads(l)
            keys = ("config", "shape", "B", "plain", "ms_per_step", "ms_hipgraph", "ms_eager", "value", "train_step_ms", "fused_loss_and_update", "head_in_place", "ms_per_step_eager", "dropout", "dropout_launches", "chained_eager_ms", "stacked_eager_ms", "chained_graph_ms", "stacked_graph_ms", "n_gpus")
            cfg = j.get("config")
            print(os.path.basename(f), {k: (j[k] if k != "config" or not isinstance(cfg, dict) else cfg.get("batch_per_gpu")) for k in keys if k in j})
for f in sorted(glob.glob(O + "/*pmc_traffic*.json")):
    k = json.load(open(f))["kernels"]; print(os.path.basename(f), {n: round(v["hbm_bytes_per_launch"] / 1e6, 1) for n, v in k.items()})
PY
ls $O
