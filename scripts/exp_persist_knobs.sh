t() { python scripts/time_decode.py 1 13 64 2>&1 | tail -2 | head -1 | sed 's/.*decode //'; }
echo "base: $(t)"
echo "base: $(t)"
for b in 1 0; do echo "POLL_B=$b: $(GVC_PERSIST_POLL_B=$b t)"; done
for h in 1 0; do echo "POLL_H=$h: $(GVC_PERSIST_POLL_H=$h t)"; done
echo "DEPTH=3: $(GVC_PERSIST_LOADER_DEPTH=3 t)"
echo "DEPTH=3: $(GVC_PERSIST_LOADER_DEPTH=3 t)"
echo "DEPTH=1: $(GVC_PERSIST_LOADER_DEPTH=1 t)"
echo "DEPTH=3 POLL_H=1 POLL_B=1: $(GVC_PERSIST_LOADER_DEPTH=3 GVC_PERSIST_POLL_H=1 GVC_PERSIST_POLL_B=1 t)"
for h in 4 16; do echo "rows H=$h B=8: $(HEADS=$h python scripts/time_decode.py 8 13 64 2>&1 | tail -2 | head -1 | sed 's/.*decode //')"; echo "rows H=$h B=16 P=109: $(HEADS=$h python scripts/time_decode.py 16 75 64 2>&1 | tail -2 | head -1| sed 's/.*decode //')"; done
