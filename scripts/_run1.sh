cd /root/repo
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
VAR_ARGS="--skip-config2 --err 0" bash scripts/gpu_variants.sh e0a e0b
VAR_ARGS=--skip-config2 bash scripts/gpu_variants.sh e1a
