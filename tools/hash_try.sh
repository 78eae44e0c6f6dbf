#!/bin/bash
# the sparse (row-piece hash) table against the dense one: parity tests, then registration / k-NN times on the configs[4] map
python -m pytest tests/test_gpu_hash.py tests/test_gpu_knn.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hash or sparse or knn" 2>&1 | tail -3
python tools/knn_sweep.py --occupancy --k-normals 32 ${SWEEP_ARGS:-} 2>&1 | grep -v amdgpu | tail -30
