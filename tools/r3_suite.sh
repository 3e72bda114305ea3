# round-3 closing run: the whole GPU suite, smoke(), then the evidence run (tools/r3_final.sh)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
bash tools/r3_final.sh
