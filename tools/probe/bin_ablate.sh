# ablation of gp_bin_scatter_kernel's stages (gp_debug_option(6, bits): 1 no counting pass, 2 no placing pass, 4 no stores,
# (averages are taken over the LAST launches only: the scene set-up renders run before the option is set)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_abl; mkdir -p $O
for a in ${ABL:-0 1 2 3 4}; do
  REPS=4 timeout 90 rocprofv3 --kernel-trace -d $O/a$a -o s --output-format csv -- python tools/probe/bin_probe.py --ablate $a > $O/a$a.log 2>&1
  echo "ablate $a: $(python tools/probe/trace_avg.py $O/a$a/s_kernel_trace.csv 4 gp_bin_scatter)"
done
