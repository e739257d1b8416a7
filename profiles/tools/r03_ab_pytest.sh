# same-box A/B of era-zk_evm_amd/ab_*.so (driver + default command, N rounds), then the GPU suite on libzkw.so
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
bash profiles/tools/r02_ab_libs.sh $1 ${2:-2} > /dev/null 2>&1
cat $T/ab_libs.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; tail -4 $T/pytest.log | head -3
