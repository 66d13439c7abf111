set -o pipefail
mkdir -p gpurun_out/r05m
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r05m/suite.log 2>&1; echo "suite rc $?" >> gpurun_out/r05m/suite.log
tail -3 gpurun_out/r05m/suite.log
python3 bench.py > gpurun_out/r05m/bench_default.json 2> gpurun_out/r05m/bench_default.err
python3 tools/perf_expectations.py gpurun_out/r05m/bench_default.json
