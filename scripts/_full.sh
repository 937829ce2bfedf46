export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python bench.py > gpurun_out/bench_e2e.json 2> gpurun_out/bench_e2e.err; tail -c 3000 gpurun_out/bench_e2e.json
