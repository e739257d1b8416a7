# Where a wave of the cycle kernel waits: a -DZKW_WAITPROF build (built on the box into a scratch copy of libzkw.so) times
# every explicit vmcnt / lgkmcnt wait of one wave over its 256 cycles.  usage: r03_waitprof.sh <tag> [extra defines]
cd $GRAFT_REPO_ROOT
T=gpurun_out/$1; mkdir -p $T
cp era-zk_evm_amd/libzkw.so /tmp/libzkw_keep.so
python -c "
import sys, os; sys.path.insert(0,'.')
import era_zk_evm_amd
from era_zk_evm_amd import build as b
b.build_lib(force=True, extra_flags=['-DZKW_WAITPROF'] + '$2'.split())"
ZKW_BENCH_NO_OTHER_CONFIGS=1 python bench.py --no-cpu-baseline --repeats 0 --steps 40 --warmup 20 --fuse 20 --streams 1 2>&1 | grep "ZKWWAIT\|ZKWPROF" | tail -40 > $T/waitprof.txt
cp /tmp/libzkw_keep.so era-zk_evm_amd/libzkw.so
cat $T/waitprof.txt
