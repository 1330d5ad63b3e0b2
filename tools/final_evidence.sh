set -o pipefail
out=gpurun_out/fin; mkdir -p $out/prof
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
GDM_FORCE_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 10 --warmup 3 --backend gloo --no-heavy-extras > $out/two_rank.json 2> $out/two_rank.err || exit 3
head -c 400 $out/bench_n1.json; echo; head -1 $out/steady_state_b16.csv; tail -1 $out/step_sequence.txt; tail -1 $out/forked_replay_queues.txt; tail -1 $out/single_stream_replay.txt; tail -c 600 $out/two_rank.json
