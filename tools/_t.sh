cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "beside_another_launch" 2>&1 | tail -15
