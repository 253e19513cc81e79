#!/bin/bash
set -o pipefail
MC_INGEST_DEBUG=1 PROBE_NO_HINT=1 timeout -k 10 300 python scripts/long_probe.py 2>&1 | grep "^lib\|count\]" | head -12
timeout -k 10 600 python -m pytest tests/test_gpu_long_records.py -x -q 2>&1 | tail -3
