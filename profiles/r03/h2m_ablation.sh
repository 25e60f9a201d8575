cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in main nochain nosort; do
  if [ $v = main ]; then unset QM_LIB_OVERRIDE; else export QM_LIB_OVERRIDE=$GRAFT_REPO_ROOT/rapmap_amd/variants/$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_h2m_abl/$v -o s -- python bench.py --sel-aln --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
  f=$(find gpurun_out/r03_h2m_abl/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "qm_h2m|qm_read_kernel|qm_sel_align|qm_sel_plan" $f | cut -d, -f1,2,4 | cut -c1-120
done
