cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r07
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r07/pytest_gpu.log 2>&1
grep -n "passed\|failed" gpurun_out/r07/pytest_gpu.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash profiles/tools/r03_campaigns.sh r07_campaigns 60 40 40
