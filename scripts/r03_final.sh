#!/bin/bash
# final validation of a build: GPU suite in both modes, random parity sweep, 2-rank rehearsal of bench.py, then the measurement round
set -o pipefail
tag=${1:?tag}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1; echo "default mode: $(tail -1 $O/tests_gpu.log)"
MBV_CONV_SPLITK=1 timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests_gpu_splitk.log 2>&1; echo "low-latency mode: $(tail -1 $O/tests_gpu_splitk.log)"
timeout -k 10 900 python tests/fuzz_parity.py 300 53 > $O/fuzz_parity.txt 2>&1; tail -3 $O/fuzz_parity.txt
timeout -k 10 600 python bench.py --gpus 2 --batch 32 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_gpus2_rehearsal.json 2> $O/bench_gpus2.err; echo "rehearsal rc $?"
bash scripts/measure_round.sh $tag skip
