t() { python scripts/time_decode.py 1 $1 $2 2>&1 | tail -2 | head -1 | sed 's/.*decode //'; }
for i in 1 2; do
echo "XCD=2 (q|k|v local too): $(GVC_PERSIST_XCD=2 t 13 64)"
echo "XCD=1 (hidden units local): $(GVC_PERSIST_XCD=1 t 13 64)"
done
echo "XCD=2 110-250 keys: $(GVC_PERSIST_XCD=2 t 75 141)"
echo "XCD=1 110-250 keys: $(GVC_PERSIST_XCD=1 t 75 141)"
