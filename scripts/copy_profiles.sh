#!/bin/bash
# copy the summaries of a measurement round (gpurun_out/<tag>/, written by scripts/r03_final.sh / measure_round.sh) into profiles/
set -e
t=${1:?tag}; R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/$t; P=$R/profiles
for f in bench_n1.json bench_n1_force_dist.json bench_gpus2_rehearsal.json other_configs.jsonl latency_b1.txt fuzz_parity.txt; do
  [ -f $O/$f ] && cp $O/$f $P/${t}_$f
done
cp $O/ks/ks_kernel_stats.csv $P/${t}_kernel_stats.csv
cp $O/mfma_infer/m_counter_collection.csv $P/${t}_mfma_util_infer.csv
cp $O/mfma_iso/m_counter_collection.csv $P/${t}_mfma_util_isolated.csv
cp $O/fetch/m_counter_collection.csv $P/${t}_istft_pmc_FETCH_SIZE.csv
cp $O/write/m_counter_collection.csv $P/${t}_istft_pmc_WRITE_SIZE.csv
python3 $R/scripts/pmc_to_json.py $t $O/fetch $O/write $O/mfma_iso $O/mfma_infer
ls $P | grep "^${t}_"
