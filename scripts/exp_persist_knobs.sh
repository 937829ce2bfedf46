# A/B of the one-launch steps' loader depth (fills in flight per loader wave); us per generation step incl. sampler
t() { python scripts/time_decode.py $1 $2 64 2>&1 | tail -2 | head -1 | sed 's/.*decode //'; }
for i in 1 2 3; do
echo "one stream depth 2: $(GVC_PERSIST_LOADER_DEPTH=2 t 1 13)"
echo "one stream depth 1: $(GVC_PERSIST_LOADER_DEPTH=1 t 1 13)"
done
echo "one stream 110-174 keys depth 2: $(GVC_PERSIST_LOADER_DEPTH=2 t 1 75)"
echo "one stream 110-174 keys depth 1: $(GVC_PERSIST_LOADER_DEPTH=1 t 1 75)"
for i in 1 2; do
echo "rows B=8 depth 2: $(GVC_ROWS_LOADER_DEPTH=2 t 8 13)"
echo "rows B=8 depth 1: $(GVC_ROWS_LOADER_DEPTH=1 t 8 13)"
echo "rows B=16 P=109 depth 2: $(GVC_ROWS_LOADER_DEPTH=2 t 16 75)"
echo "rows B=16 P=109 depth 1: $(GVC_ROWS_LOADER_DEPTH=1 t 16 75)"
done
