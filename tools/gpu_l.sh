#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_model_gpu.py -q -m gpu --timeout 900 -k "resnet or video" 2>&1 | tail -25
