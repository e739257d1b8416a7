cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r07
timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -x -q -m gpu -k expand > gpurun_out/r07/pytest_expand.log 2>&1
tail -5 gpurun_out/r07/pytest_expand.log | cut -c1-200
bash profiles/tools/r07_expand_ab.sh r07_expand_ab3 2
