#!/bin/bash
# First GPU call of round 5: the kernels round 4 finished on the lane-level emulator after its GPU budget was spent.
#   1. their first run on an MI355X (tests that are skipped until RSP_UNMEASURED=1),
#   2. the bench with each of them on / off on the same box (A/B), incl. the encoder replayed as a hipGraph,
#   3. rocprofv3 kernel statistics of the bench with all of them on.
# Usage: gpurun --timeout 1200 -- 'bash tools/r5_first.sh' (about 15 minutes of box time: 10 bench runs of 50-70 s, one of them
# under rocprofv3, two short pytest selections, two micro tools); results in gpurun_out/r5/first/.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5/first
mkdir -p $O
t0=$(date +%s)
line() { python - "$1" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = {n: v['ms'] for n, v in r['kernels'].items() if any(s in n for s in ('t2i', 'i2t', 'upscale', 'convT'))}
    print(r['value'], r['unit'], r['ms_per_step'], 'ms/step, canary', (r.get('parity_canary') or {}).get('ok'), k)
except Exception as e:
    print('no bench line:', e)
PY
}
RSP_UNMEASURED=1 timeout 300 python -m pytest -m gpu -q -s tests/test_gpu_baseline_configs.py tests/test_gpu_kernels.py \
  tests/test_gpu_encoder.py -k "unmeasured or t2i_fold or folded_token or graph_replay" > $O/unmeasured.log 2>&1
echo "[first GPU run of the emulator-verified kernels] rc=$? $(( $(date +%s) - t0 )) s: $(tail -n 1 $O/unmeasured.log)"
for cfg in "base:" "fold:--t2i-fold on" "fold2:--t2i-fold on --t2i-fold-variant 2" "fold3:--t2i-fold on --t2i-fold-variant 3" \
           "up:--upscale-fused on" "both:--t2i-fold on --upscale-fused on"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  timeout 90 python bench.py --no-cpu-baseline $flags > $O/bench_$name.json 2> $O/bench_$name.err
  echo "[bench $name] rc=$? $(( $(date +%s) - t0 )) s: $(line $O/bench_$name.json)"
done
# the encoder as a replayed hipGraph: ViT-H (GPU-bound: expect no change) and ViT-B at batch 8 (configs[1]: interpreter-bound)
for cfg in "graph_h:--encoder-graph on" "b8:--arch base" "graph_b8:--arch base --encoder-graph on"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  timeout 120 python bench.py --no-cpu-baseline $flags > $O/bench_$name.json 2> $O/bench_$name.err
  echo "[bench $name] rc=$? $(( $(date +%s) - t0 )) s: $(line $O/bench_$name.json)"
done
bash tools/gpu_job.sh r5/first "p:both:--t2i-fold on --upscale-fused on"
timeout 120 python tools/decoder_tail_micro.py 800 10 > $O/decoder_tail_micro.txt 2>&1; echo "[decoder tail micro] rc=$? $(( $(date +%s) - t0 )) s"; cat $O/decoder_tail_micro.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma16_probe tools/probes/mfma16_probe.hip && /tmp/mfma16_probe > $O/mfma16_probe.txt 2>&1; tail -n 2 $O/mfma16_probe.txt
