#!/bin/bash
# One measurement pass of the current build on the GPU box: tests, bench (+ the sharded path on one rank), the other
# configurations, single-utterance latency, per-kernel statistics and the MfmaUtil / HBM-traffic counter passes
# (each counter in its own run, program directly after `--`).  Everything lands in gpurun_out/<tag>/.
# usage (inside gpurun): bash scripts/measure_round.sh <tag> [skip-tests]
set -o pipefail
tag=${1:?tag}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
if [ -z "$2" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1 || { tail -5 $O/tests_gpu.log; exit 1; }
  tail -1 $O/tests_gpu.log
fi
timeout -k 10 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
MBV_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_n1_force_dist.json 2>> $O/bench_n1.err || exit 1
: > $O/other_configs.jsonl
for cb in "ljs_ms_istft_vits 64" "uudb_ms_istft_vits_ms 32" "ljs_mini_mb_istft_vits 64" "ljs_istft_vits 64" "ljs_mini_istft_vits 64" "ljs_mb_istft_vits 256"; do
  set -- $cb
  timeout -k 10 400 python bench.py --config $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline >> $O/other_configs.jsonl 2>> $O/bench_n1.err || exit 1
done
timeout -k 10 300 python bench.py --ragged --steps 5 --warmup 2 --no-cpu-baseline >> $O/other_configs.jsonl 2>> $O/bench_n1.err || exit 1
{ echo "== MBV_CONV_SPLITK=0"; MBV_CONV_SPLITK=0 timeout -k 10 200 python scripts/latency_b1.py; echo "== MBV_CONV_SPLITK=1"; MBV_CONV_SPLITK=1 timeout -k 10 200 python scripts/latency_b1.py; } 2>&1 | grep -v amdgpu.ids > $O/latency_b1.txt || exit 1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 $R/scripts/run_infer.py ljs_mb_istft_vits 64 5 > $O/ks.log 2>&1 || exit 1
rm -f $O/ks/*kernel_trace.csv
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/mfma_infer -o m -- python3 $R/scripts/run_infer.py ljs_mb_istft_vits 64 2 > $O/mfma_infer.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/mfma_iso -o m -- python3 $R/scripts/prof_kernels.py conv 3 > $O/mfma_iso.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o m -- python3 $R/scripts/prof_kernels.py istft 5 > $O/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o m -- python3 $R/scripts/prof_kernels.py istft 5 > $O/write.log 2>&1 || exit 1
rm -f $O/*/*kernel_trace.csv
cd $R
cat $O/bench_n1.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stage_ms'), d['roofline']['frac'])"
cat $O/latency_b1.txt
