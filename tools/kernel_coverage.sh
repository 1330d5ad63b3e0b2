# The GPU suite under rocprofv3 (kernel trace + stats only), then tools/kernel_coverage.py over the stats: kernels no test launches.
# The three tests that start further processes are left out (no child processes under the profiler's preloaded library).
out=gpurun_out/cov; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o c -- python3 -m pytest tests -q -m gpu -x -k "not config3 and not two_ranks and not group_of_one" > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $out/pytest.log
S=$(find $out -name "*kernel_stats.csv" | head -1)
python tools/kernel_coverage.py $S > $out/coverage.txt
find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete
head -40 $out/coverage.txt
