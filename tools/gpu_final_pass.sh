set -e
mkdir -p gpurun_out/fin
timeout -k 10 900 python -m pytest tests -m gpu -q -s > gpurun_out/fin/gpu_tests.log 2>&1
tail -2 gpurun_out/fin/gpu_tests.log
timeout -k 10 900 python tools/make_traffic_json.py profiles/r03 > gpurun_out/fin/traffic.log 2>&1
cp profiles/r03/traffic.json gpurun_out/fin/traffic.json
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > gpurun_out/fin/bench.json 2> gpurun_out/fin/bench.err
tail -c 600 gpurun_out/fin/bench.json
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/fin/prof -o b -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/fin/bench_under_rocprof.json 2> $ROOT/gpurun_out/fin/rocprof.err
cd $ROOT
find gpurun_out/fin/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/fin/bench_kernel_stats.csv \;
find gpurun_out/fin/prof -name "*kernel_trace.csv" -delete
ls gpurun_out/fin
