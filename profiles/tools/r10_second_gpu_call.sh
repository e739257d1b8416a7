#!/bin/bash
# The A/B round 6 could not run: -DZKW_SHORT_CLASS (the short cycle reads an instruction's class from bits packed into its ISA entry:
# 212 -> 187 instructions per short NOP in the assembly, profiles/r10_short_cycle_census.txt) against the default build, same box.
# HERE, before the call (the libraries travel with the snapshot):
#   python profiles/tools/build_ab.py a_base WORK; python profiles/tools/build_ab.py b_short_class WORK -DZKW_SHORT_CLASS
#   python profiles/tools/build_ab.py c_short_stack WORK -DZKW_SHORT_CLASS -DZKW_SHORT_STACK      (the census says this one does not pay)
# then   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash profiles/tools/r10_second_gpu_call.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r10_ab_short_class; mkdir -p $OUT
cp era-zk_evm_amd/libzkw.so /tmp/libzkw_keep.so
cp era-zk_evm_amd/ab_b_short_class.so era-zk_evm_amd/libzkw.so
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3) > $OUT/pytest_short_class.log
cp /tmp/libzkw_keep.so era-zk_evm_amd/libzkw.so
bash profiles/tools/r02_ab_libs.sh r10_ab_short_class 3
cat $OUT/pytest_short_class.log
