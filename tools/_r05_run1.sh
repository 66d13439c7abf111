set -o pipefail
mkdir -p gpurun_out/r05a
python3 bench.py --frames 12 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-probe --sustain-seconds 0 > gpurun_out/r05a/cold12.json 2> gpurun_out/r05a/cold12.err
python3 bench.py --frames 12 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-probe --sustain-seconds 0 --schedule serial > gpurun_out/r05a/cold12_serial.json 2>> gpurun_out/r05a/cold12.err
python3 bench.py --frames 12 --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-probe --sustain-seconds 0 > gpurun_out/r05a/warm12.json 2>> gpurun_out/r05a/cold12.err
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05a/suite.log 2>&1; echo "suite rc $?" >> gpurun_out/r05a/suite.log
tail -5 gpurun_out/r05a/suite.log
