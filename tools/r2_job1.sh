# round-2 GPU call 1: baseline of the headline configuration (ViT-H, batch 8) + PMC passes on its dominant kernels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 400 python bench.py --arch huge --steps 3 --warmup 1 --no-cpu-baseline --shapes > gpurun_out/r2/bench_vith_base.json 2> gpurun_out/r2/bench_vith_base.err
bash tools/pmc_round2.sh r2_base
