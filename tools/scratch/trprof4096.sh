cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o bench -- python $GRAFT_REPO_ROOT/tools/bench_train_path.py 4096 200 10 1 > /dev/null 2>&1
find /tmp/prof_x -name "*kernel_stats.csv" | head -1 | xargs head -9 | cut -c1-150
