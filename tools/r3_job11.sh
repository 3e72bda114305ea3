#!/bin/bash
# round-3 GPU job 11: whole GPU suite (no -x)
cd "$(dirname "$0")/.."
O=gpurun_out/r3; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q --timeout=900 > $O/suite11.log 2>&1
tail -25 $O/suite11.log
