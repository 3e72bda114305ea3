#!/bin/bash
# SAMDet (f4): kernel + module + end-to-end parity on the GPU
O=gpurun_out/r3; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_samdet.py -x -q -s -m gpu > $O/samdet_tests.log 2>&1; echo "samdet tests rc=$?"
tail -30 $O/samdet_tests.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" > $O/gemm_tests13.log 2>&1; echo "gemm tests rc=$?"
tail -3 $O/gemm_tests13.log
