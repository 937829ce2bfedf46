export TMPDIR=/tmp
R=$PWD
python -m pytest tests/test_gpu_frontend.py tests/test_gpu_shell.py -x -q -m gpu 2>&1 | tail -5
python scripts/time_hubert.py 16000 1; python scripts/time_hubert.py 160000 1; python scripts/time_hubert.py 96000 8
