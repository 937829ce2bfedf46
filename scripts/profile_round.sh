#!/bin/bash
# rocprofv3 passes behind profiles/rNN_*: kernel stats of the bench, FETCH / WRITE of the one-launch steps (one stream fp32; eight streams
# bf16 weights + bf16 KV cache, with fp32 and with bf16 activations), MFMA counters of the prefill and of the Perceiver, ContentVec / Perceiver kernel stats.
# (every counter pass runs under `timeout`: a pass that hangs must not eat the box's time limit)
# usage (on the GPU box, from the repo root): bash scripts/profile_round.sh r06
R=${1:-r06}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python $OLDPWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-cold > $OUT/${R}_bench_under_rocprof.json 2> $OUT/${R}_bench_under_rocprof.err
cp $(find $OUT/prof_bench -name '*kernel_stats.csv' | head -1) $OUT/${R}_rocprofv3_kernel_stats_bench.csv
cp $(find $OUT/prof_bench -name '*domain_stats.csv' | head -1) $OUT/${R}_rocprofv3_domain_stats_bench.csv 2>/dev/null
rm -rf $OUT/prof_bench
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/prof_$ctr -- python $OLDPWD/scripts/time_decode.py 1 13 8 > /dev/null 2>&1
  echo "== $ctr, scripts/time_decode.py 1 13 8" >> $OUT/${R}_pmc_fetch_write_decode.txt
  python $OLDPWD/scripts/pmc_fetch.py $OUT/prof_$ctr k_decode_persist >> $OUT/${R}_pmc_fetch_write_decode.txt
  rm -rf $OUT/prof_$ctr
  for W in bf16_kv bf16_act; do
    WEIGHTS=$W timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/prof_$ctr -- python $OLDPWD/scripts/time_decode.py 8 13 24 > /dev/null 2>&1
    echo "== $ctr, WEIGHTS=$W scripts/time_decode.py 8 13 24" >> $OUT/${R}_pmc_fetch_write_rows_$W.txt
    python $OLDPWD/scripts/pmc_fetch.py $OUT/prof_$ctr k_rows_persist >> $OUT/${R}_pmc_fetch_write_rows_$W.txt
    rm -rf $OUT/prof_$ctr
  done
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/prof_mfma -- python $OLDPWD/scripts/time_prefill.py 5 75 > $OUT/${R}_prefill_5x110_under_pmc.txt 2>&1
python $OLDPWD/scripts/pmc_mfma.py $OUT/prof_mfma > $OUT/${R}_pmc_mfma_prefill.csv
rm -rf $OUT/prof_mfma
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_hub -- python $OLDPWD/scripts/time_hubert.py 16000 1 > $OUT/${R}_time_hubert.txt 2>&1
cp $(find $OUT/prof_hub -name '*kernel_stats.csv' | head -1) $OUT/${R}_rocprofv3_kernel_stats_contentvec.csv
rm -rf $OUT/prof_hub
