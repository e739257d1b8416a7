cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r07
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r07/pytest_gpu.log 2>&1
tail -4 gpurun_out/r07/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
