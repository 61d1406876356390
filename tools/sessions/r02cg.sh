cd $GRAFT_REPO_ROOT
for gb in 32 80 96 112; do for w in 1 0; do echo -n "B=$gb WRIDE=$w: "; VMLMF_WRIDE=$w timeout 200 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null < /dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done; done
