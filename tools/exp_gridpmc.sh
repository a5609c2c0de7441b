cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for l in libgapro_hip.so libgapro_hip_grid.so; do
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $c -d /tmp/pm -o x --output-format csv -- python3 $R/tools/bench_fit.py --sizes 384 --fits 256 --reps 1 --lib $l > /tmp/pm.log 2>&1
    echo "== $l $c"; python3 $R/tools/pmc_summary.py /tmp/pm | grep "k_svgp_fit<" | cut -c40-200
  done
done
