# Phase timing of the cycle kernel: a profiling build (-DZKW_PROFILE, built on the box into a scratch copy of libzkw.so)
# prints shader clocks per phase / opcode for one wave.  usage: r02_phase.sh <tag>
cd $GRAFT_REPO_ROOT
T=gpurun_out/$1; mkdir -p $T
cp era-zk_evm_amd/libzkw.so /tmp/libzkw_keep.so
python -c "
import sys, os; sys.path.insert(0,'.')
import era_zk_evm_amd
from era_zk_evm_amd import build as b
b.build_lib(force=True, extra_flags=['-DZKW_PROFILE'] + os.environ.get('ZKW_PROFILE_EXTRA', '').split())"
for F in 20; do echo "fuse $F" >> $T/phase.txt; ZKW_BENCH_NO_OTHER_CONFIGS=1 timeout 300 python bench.py --no-cpu-baseline --repeats 0 --steps $((F*2)) --warmup $F --fuse $F --streams 1 2>&1 | grep ZKWPROF | tail -40 >> $T/phase.txt; done
cp /tmp/libzkw_keep.so era-zk_evm_amd/libzkw.so
cat $T/phase.txt
