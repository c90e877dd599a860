set -x
python tools/fused_trace.py abx/lib_trace.so --tail 2>&1 | tee gpurun_out/r06_tail_trace1.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-other-configs --fdsa-pair > gpurun_out/r06_a_bench_pair_$i.json 2> gpurun_out/r06_a_bench_pair_$i.err
python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/r06_a_bench_tail_$i.json 2> gpurun_out/r06_a_bench_tail_$i.err
done
python -m pytest tests -m gpu -x -q -k "fdsa or end_to_end or configs" 2>&1 | tail -15 | tee gpurun_out/r06_a_tests.txt
