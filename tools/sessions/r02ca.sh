#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02ca; mkdir -p $O
BENCH_NOCPU=1 timeout 900 python tools/bench_configs.py > $O/configs.jsonl 2>/dev/null < /dev/null
python - <<'PY'
import json
new={}
for l in open("gpurun_out/r02ca/configs.jsonl"):
    l=l.strip()
    if l.startswith("{"):
        j=json.loads(l); new[j["config"]]=j
old={}
for l in open("profiles/r02_zz_configs.jsonl"):
    l=l.strip()
    if l.startswith("{"):
        j=json.loads(l); old[j["config"]]=j
for k,j in new.items():
    o=old.get(k,{})
    print("%-62s graph %.4f (was %s) eager %.4f (was %s)" % (k[:62], j.get("ms_hipgraph",0), o.get("ms_hipgraph"), j.get("ms_eager",0), o.get("ms_eager")))
PY
