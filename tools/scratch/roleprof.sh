cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export SVAE_AMD_LIB=$GRAFT_REPO_ROOT/tests/_variants/$v.so
  rm -rf /tmp/prof_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o bench -- python $GRAFT_REPO_ROOT/tools/bench_train_path.py 512 200 10 1 > /dev/null 2>&1
  echo -n "== $v: "; find /tmp/prof_$v -name "*kernel_stats.csv" | head -1 | xargs grep "sweep1_prod" | cut -d, -f4
done
