#!/bin/bash
# Folds gpurun_out/round/* (written by tools/profile_round.sh on the GPU box) into the committed profiles/rNN_* files.
# usage: bash tools/profile_fold.sh 01
set -eu
N=${1:?round number, e.g. 01}
O=gpurun_out/round
P=profiles
mkdir -p $P
tail -1 $O/bench.json > $P/r${N}_bench.json
for m in 0 1; do
  cp $O/trace_math$m/p_kernel_stats.csv $P/r${N}_kernel_stats_math$m.csv
  grep -E "^\s+[0-9]+ \S+\s+[0-9.]+ ms/step|sum of kernels" $O/trace_math$m.err > $P/r${N}_bench_dump_ops_math$m.txt
  python tools/trace_to_ops.py $O/trace_math$m/p_kernel_trace.csv $O/trace_math$m.err 20 > $P/r${N}_per_op_math$m.csv
done
cp $P/r${N}_per_op_math0.csv $P/r${N}_per_op.csv
python tools/pmc_to_traffic.py $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv > $P/r${N}_traffic.json
cp $O/fear_m_trace/p_kernel_stats.csv $P/r${N}_fear_m_kernel_stats.csv
python tools/trace_to_ops.py $O/fear_m_trace/p_kernel_trace.csv $O/fear_m_trace.err 10 > $P/r${N}_fear_m_per_op.csv
python tools/pmc_to_traffic.py $O/fear_m_pmc_FETCH_SIZE/p_counter_collection.csv $O/fear_m_pmc_WRITE_SIZE/p_counter_collection.csv > $P/r${N}_fear_m_traffic.json
cp $O/train_trace/p_kernel_stats.csv $P/r${N}_train_kernel_stats.csv
{ grep -h "mode=\|ms  " $O/train_plain.err; python tools/train_traffic.py $O $O/train_trace 7 4; } > $P/r${N}_train_traffic.txt
python tools/pmc_summary.py $O/sq_pass1 $O/sq_pass2 $O/sq_pass3 > $P/r${N}_sq_counters.txt
python tools/pmc_summary.py --all $O/train_sq_pass1 $O/train_sq_pass2 $O/train_sq_pass3 > $P/r${N}_train_sq_counters.txt
python -c "import json,sys; print(json.dumps(json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['latency_batch1']))" $P/r${N}_bench.json > $P/r${N}_latency_batch1.json
tail -1 $O/bench_force_dist.json > $P/r${N}_bench_force_dist_1gpu.json
ls -la $P
