for cold in 0 64; do
 echo "== cold=$cold"; GVC_PROBE_COLD=$cold python scripts/time_gemm.py 48 110 2>&1 | grep "^M=" | sed 's/tiled.*skinny/skinny/'
done
python scripts/time_prefill.py 1 13; python scripts/time_prefill.py 1 75; python scripts/time_hubert.py 16000 1
