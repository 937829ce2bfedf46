export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python bench.py > gpurun_out/bench_e2e.json 2> gpurun_out/bench_e2e.err; tail -c 400 gpurun_out/bench_e2e.json
python bench.py --streams 8 --no-cpu-baseline --no-offline --no-harness > gpurun_out/bench_streams8.json 2>/dev/null; head -c 400 gpurun_out/bench_streams8.json
python -c "import __graft_entry__ as g; g.smoke()"
