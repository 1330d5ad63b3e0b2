# Round-4 evidence, one GPU call: bench line (+ extras), rocprofv3 kernel trace of the same command and its summaries, the PMC traffic
# passes (separate --pmc runs, MI355X_MICROARCH.md), the training step's steady-state profile, the two-rank rehearsal through
# `bench.py --gpus 2` itself.  Everything lands under gpurun_out/fin; copy what is to be judged into profiles/.
set -o pipefail
out=gpurun_out/fin; mkdir -p $out/prof $out/pmc_f $out/pmc_w $out/proft
python bench.py > $out/bench_n1.json 2> $out/bench_n1.err || exit 1
cp gpurun_out/bench_extras.json $out/bench_extras.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o b -- python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $out/bench_prof.json 2> $out/bench_prof.err || exit 2
T=$(find $out/prof -name "*kernel_trace.csv" | head -1); S=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python tools/summarize_trace.py $T --steps 10 > $out/steady_state_b16.csv
python tools/step_sequence.py $T > $out/step_sequence.txt
python tools/graph_queues.py $T 19 > $out/forked_replay_queues.txt
python tools/graph_queues.py $T 32 > $out/single_stream_replay.txt
cp $S $out/kernel_stats_whole_process.csv
find $out/prof -name "*.csv" -delete; find $out/prof -name "*.db" -delete
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_f -o f -- python3 tools/pmc_kernels.py > $out/pmc_f.log 2>&1 || exit 4
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_w -o w -- python3 tools/pmc_kernels.py > $out/pmc_w.log 2>&1 || exit 5
python tools/pmc_kernels.py --summarize $(find $out/pmc_f -name "*counter_collection.csv" | head -1) $(find $out/pmc_w -name "*counter_collection.csv" | head -1) > $out/pmc_traffic.json || exit 6
find $out/pmc_f $out/pmc_w -name "*.csv" -delete; find $out/pmc_f $out/pmc_w -name "*.db" -delete
echo "pmc done"
rocprofv3 --kernel-trace --output-format csv -d $out/proft -o t -- python3 tools/bench_train.py > $out/train_prof.log 2>&1 || exit 7
TT=$(find $out/proft -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $TT --steps 5 --periodic > $out/train_steady_b24.csv
find $out/proft -name "*.csv" -delete; find $out/proft -name "*.db" -delete
echo "train profile done"
python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --no-heavy-extras > $out/two_rank.json 2> $out/two_rank.err || exit 3
python bench.py --group-of-one --backend nccl --no-cpu-baseline --no-extras --steps 20 --warmup 3 > $out/rccl_group_of_one.json 2> $out/rccl_group_of_one.err || exit 8
head -c 300 $out/bench_n1.json; echo; head -1 $out/steady_state_b16.csv; tail -1 $out/step_sequence.txt; tail -1 $out/forked_replay_queues.txt; head -1 $out/train_steady_b24.csv; head -c 200 $out/two_rank.json
