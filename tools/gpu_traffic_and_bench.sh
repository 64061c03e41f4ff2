set -e
mkdir -p gpurun_out/fin
timeout -k 10 1000 python tools/make_traffic_json.py profiles/r03 > gpurun_out/fin/traffic.log 2>&1
cp profiles/r03/traffic.json gpurun_out/fin/traffic.json
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > gpurun_out/fin/bench.json 2> gpurun_out/fin/bench.err
tail -c 1500 gpurun_out/fin/bench.json
