#!/bin/bash
# What rocFFT's run-time compilation costs the valley / ridge tests: the same test file with an empty and
# then a warm kernel cache.   usage (GPU box, repository root): bash tools/rocfft_cold_cache.sh
export ROCFFT_RTC_CACHE_PATH=/tmp/rocfft_cold_cache.db
rm -f $ROCFFT_RTC_CACHE_PATH
S=$SECONDS; python -m pytest tests/test_gpu_valley_ridge.py -m gpu -x -q 2>&1 | grep -E "passed|failed"; echo "cold cache: $((SECONDS-S)) s"
S=$SECONDS; python -m pytest tests/test_gpu_valley_ridge.py -m gpu -x -q 2>&1 | grep -E "passed|failed"; echo "warm cache: $((SECONDS-S)) s"
ls -la /tmp/rocfft_cold_cache.db 2>&1 | head -2
