for cold in 0 64; do for touch in 0 1; do
 echo "== cold=$cold touch=$touch"; GVC_PROBE_COLD=$cold GVC_PROBE_TOUCH=$touch python scripts/time_gemm.py 48 110 2>&1 | grep "^M=" | sed 's/tiled.*skinny/skinny/'
done; done
echo "== cold=64 SK=4 (mlp c_proj as shipped)"; GVC_PROBE_COLD=64 GVC_PROBE_SK=4 python scripts/time_gemm.py 48 2>&1 | grep "^M=" | sed 's/tiled.*skinny/skinny/'
echo "== cold=64 SK=4 touch"; GVC_PROBE_COLD=64 GVC_PROBE_SK=4 GVC_PROBE_TOUCH=1 python scripts/time_gemm.py 48 2>&1 | grep "^M=" | sed 's/tiled.*skinny/skinny/'
