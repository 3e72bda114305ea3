#!/bin/bash
# Round 5, first GPU call: the paths round 4 finished on the lane-level emulator only -- first run on an MI355X, then A/B
# of every switch on ONE model in ONE process (tools/ab_bench.py), ViT-H and ViT-B.  Results in gpurun_out/r5/job1/.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5/job1
mkdir -p $O
t0=$(date +%s)
RSP_UNMEASURED=1 timeout 420 python -m pytest -m gpu -q -s tests/test_gpu_baseline_configs.py tests/test_gpu_kernels.py \
  tests/test_gpu_encoder.py -k "unmeasured or t2i_fold or folded_token or graph_replay" > $O/unmeasured.log 2>&1
echo "[first GPU run of the emulator-verified kernels] rc=$? $(( $(date +%s) - t0 )) s: $(tail -n 1 $O/unmeasured.log)"
grep -E "PASSED|FAILED|ERROR|passed|failed" $O/unmeasured.log | tail -n 12
timeout 400 python tools/ab_bench.py --kernels --steps 4 --rounds 3 "base:t2i_fold=False,upscale_fused=False,graph=False" \
  "fold0:t2i_fold=True,t2i_fold_variant=0" "fold1:t2i_fold_variant=1" "fold2:t2i_fold_variant=2" "fold3:t2i_fold_variant=3" \
  "up:t2i_fold=False,t2i_fold_variant=0,upscale_fused=True" "fold0+up:t2i_fold=True" "fold0+up+graph:graph=True" \
  > $O/ab_huge.txt 2> $O/ab_huge.err
echo "[ab huge] rc=$? $(( $(date +%s) - t0 )) s"; cat $O/ab_huge.txt | cut -c1-220; tail -n 3 $O/ab_huge.err
timeout 300 python tools/ab_bench.py --arch base --steps 6 --rounds 3 "base:graph=False" "graph:graph=True" > $O/ab_base.txt 2> $O/ab_base.err
echo "[ab base] rc=$? $(( $(date +%s) - t0 )) s"; cat $O/ab_base.txt | cut -c1-220; tail -n 3 $O/ab_base.err
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma16_probe tools/probes/mfma16_probe.hip && /tmp/mfma16_probe > $O/mfma16_probe.txt 2>&1; tail -n 3 $O/mfma16_probe.txt
echo "[done] $(( $(date +%s) - t0 )) s"
