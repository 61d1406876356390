#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_dropout.py -m gpu -q -k "captured or stock" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_rb.py tests/test_gpu_modules.py -m gpu -q -x 2>&1 | tail -5
