cd /root/repo
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
bash scripts/gpu_variants.sh a b c
