#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -q -x -s -k "psfpt" 2>&1 | tail -6
