timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_action_layout.py 2>&1 | tail -3
