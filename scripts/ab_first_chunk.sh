# Same-box A/B of two trees' headline legs (first-chunk latency, utterances/s).  Prepare in the build container (the GPU box has no .git):
#   mkdir .ab_old && git archive <old commit> bench.py genvc_amd oracle tests/golden __graft_entry__.py include | tar -x -C .ab_old && cp -r genvc_amd/lib .ab_old/genvc_amd/
#   gpurun -- 'bash scripts/ab_first_chunk.sh'; rm -rf .ab_old
# Round 6 (conditioning chain enqueued behind the first segment's ContentVec + DVAE): old 7.41 / 7.47 / 7.38 ms, new 7.18 / 7.22 / 7.13 ms.
mkdir -p gpurun_out
F="--steps 12 --warmup 3 --no-cpu-baseline --no-offline --no-harness --no-extra --no-cold"
for i in 1 2 3; do
  (cd .ab_old && python bench.py $F 2>/dev/null | tail -1) > gpurun_out/ab_old_$i.json
  python bench.py $F 2>/dev/null | tail -1 > gpurun_out/ab_new_$i.json
done
python - <<'PY'
import json,glob
for k in ("old","new"):
    for f in sorted(glob.glob(f"gpurun_out/ab_{k}_*.json")):
        d=json.loads(open(f).read())
        print(k, d["value"], d.get("first_chunk_latency_ms"), d.get("first_chunk_latency_ms_registered_speaker"))
PY
python -m pytest tests/test_gpu_round6.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -3
