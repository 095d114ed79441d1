# durations of the binning kernels (last 6 forward renders of the bench scene), counting path and radix path
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_bint; mkdir -p $O
for v in ${V:-count radix}; do
  REPS=6 timeout 90 rocprofv3 --kernel-trace -d $O/$v -o s --output-format csv -- python tools/probe/bin_probe.py $([ $v = radix ] && echo --radix) > $O/$v.log 2>&1
  echo "== $v"; python tools/probe/trace_avg.py $O/$v/s_kernel_trace.csv 6 gp_bin gp_duplicate gp_tile_ranges gp_scan_block | head -8
  python tools/probe/trace_avg.py $O/$v/s_kernel_trace.csv 24 gp_radix | head -8
done
