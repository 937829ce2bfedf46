python scripts/time_hubert.py 16000 1; python scripts/time_hubert.py 160000 1; python scripts/time_hubert.py 96000 8; python scripts/time_hubert.py 64000 8
python -m pytest tests/test_gpu_frontend.py -x -q -m gpu -k hubert 2>&1 | tail -2
