cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02cp; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ke -o k -- python3 $R/tools/run_e.py --nograph ) > $O/ke.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db ke) $O/r02_zz4_config_e_layer_kernel_stats.csv "config E layer (V4 group, H=650, ranks 32/[32,32], B=256, T=35), end of round 2: rocprofv3 --kernel-trace --stats -- python3 tools/run_e.py --nograph" > /dev/null 2>&1
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/kc -o k -- python3 $R/tools/run_c.py ) > $O/kc.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db kc) $O/r02_zz4_config_c_kernel_stats.csv "config C (2 x 256, rank 24, B 128, T 24, I 77, fp32) through the wavefront launches, end of round 2: rocprofv3 --kernel-trace --stats -- python3 tools/run_c.py" > /dev/null 2>&1
rm -rf $O/ke $O/kc
head -12 $O/r02_zz4_config_e_layer_kernel_stats.csv | cut -c1-100; head -12 $O/r02_zz4_config_c_kernel_stats.csv | cut -c1-100
