# the whole GPU suite + config C same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06f; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $O/tests.txt; cat $O/tests.txt
bash tools/probes/r06_ab_c.sh > $O/ab_c.txt 2>&1; cat $O/ab_c.txt
