#!/bin/bash
# the last GPU seconds of round 2: the one-sample path (parts of the file, merge, shared-out filter_mismappers) with the real kernels
mkdir -p gpurun_out
timeout 150 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_sample or in_parts" > gpurun_out/r02k_pytest.log 2>&1
echo "pytest exit $?"; tail -15 gpurun_out/r02k_pytest.log
