export TMPDIR=/tmp
R=$PWD
python -m pytest tests/test_gpu_shell.py -x -q -m gpu -k "vocod" 2>&1 | tail -3
python scripts/time_prefill.py 1 13; python scripts/time_prefill.py 1 75
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_pf -o pf --output-format csv -- python $R/scripts/time_prefill.py 1 13 > /dev/null 2>&1
cd $R
