# same-box A/B of config C (wavefront launches): round-5 tree against this tree, interleaved
for i in 1 2 3; do
  (cd gpurun_tmp/r05 && python tools/bench_configs.py 2>/dev/null | grep "C(fp32)" | cut -c1-140 | sed 's/^/r05 /')
  python tools/bench_configs.py 2>/dev/null | grep "C(fp32)" | cut -c1-140 | sed 's/^/r06 /'
done
