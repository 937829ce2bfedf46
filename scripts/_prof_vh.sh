export TMPDIR=/tmp
R=$PWD
python -m pytest tests/test_gpu_frontend.py tests/test_gpu_shell.py -x -q -m gpu -k "hifigan or vocod or harness or stream" 2>&1 | tail -5
python scripts/time_vocoder.py 8 1; python scripts/time_vocoder.py 8 2; python scripts/time_vocoder.py 48 1
GVC_VOCODER_GRAPH=2 python scripts/time_vocoder.py 8 1
GVC_VOCODER_GRAPH=0 python scripts/time_vocoder.py 8 1
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_voc -o voc --output-format csv -- python $R/scripts/time_vocoder.py 8 1 > /dev/null 2>&1
cd $R
