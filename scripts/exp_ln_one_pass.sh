# VERDICT round 3 item 4(a), the buildable form: LayerNorm statistics of the one-stream step in ONE pass (sum and sum of squares reduced side by side)
t() { python scripts/time_decode.py 1 13 64 2>&1 | tail -2 | head -1 | sed 's/.*decode //'; }
for i in 1 2 3; do
echo "two-pass (shipped): $(GVC_PERSIST_LN_ONE_PASS=0 t)"
echo "one-pass          : $(GVC_PERSIST_LN_ONE_PASS=1 t)"
done
GVC_PERSIST_LN_ONE_PASS=1 python -m pytest tests/test_gpu_gpt.py -m gpu -q -k "full or golden or reference" 2>&1 | tail -3
