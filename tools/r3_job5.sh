#!/bin/bash
# round-3 GPU job 5: raw-buffer range-check semantics probe; gemm_s2 (branch-free epilogue, tickets, L2 prefetch): check, time, trace
cd "$(dirname "$0")/.."
O=gpurun_out/r3
mkdir -p $O
( cd tools/probes && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/buf_probe buf_probe.hip && /tmp/buf_probe ) > $O/buf_probe.log 2>&1
cat $O/buf_probe.log
timeout 600 python tools/gemm_s2_exp.py check > $O/s2_check5.log 2>&1
grep -c "^OK" $O/s2_check5.log; grep -E "FAIL|ALL|SOME|part|fault" $O/s2_check5.log | head -20
timeout 600 python tools/gemm_s2_exp.py time > $O/s2_time5.log 2>&1
tail -8 $O/s2_time5.log
timeout 300 python tools/gemm_s2_exp.py trace > $O/s2_trace5.log 2>&1
grep -E "tiles|histogram" $O/s2_trace5.log
