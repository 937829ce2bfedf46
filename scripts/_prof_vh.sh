export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/prof_stream
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stream -o st --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-offline --no-harness > $R/gpurun_out/prof_stream/bench.json 2>/dev/null
cd $R
tail -c 300 gpurun_out/prof_stream/bench.json
