OUT=$PWD/gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_F -- python $R/scripts/time_decode.py 1 13 8 > /dev/null 2>&1
echo "== FETCH_SIZE (second try), scripts/time_decode.py 1 13 8" >> $OUT/r06_pmc_fetch_write_decode.txt
python $R/scripts/pmc_fetch.py $OUT/prof_F k_decode_persist >> $OUT/r06_pmc_fetch_write_decode.txt
rm -rf $OUT/prof_F
WEIGHTS=bf16_act timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_W -- python $R/scripts/time_decode.py 8 13 24 > /dev/null 2>&1
echo "== WRITE_SIZE (second try), WEIGHTS=bf16_act scripts/time_decode.py 8 13 24" >> $OUT/r06_pmc_fetch_write_rows_bf16_act.txt
python $R/scripts/pmc_fetch.py $OUT/prof_W k_rows_persist >> $OUT/r06_pmc_fetch_write_rows_bf16_act.txt
rm -rf $OUT/prof_W
