# Phase timing of the cycle kernel: a profiling build (-DZKW_PROFILE, built locally into
# profiles/tools/libzkw_profile_build.so, git-ignored) prints the shader-clock time wave 28 of batch 0 spent in the
# four phases of its VM cycles.  Run through gpurun; the product library is restored afterwards.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cp era-zk_evm_amd/libzkw.so /tmp/libzkw_orig.so
cp profiles/tools/libzkw_profile_build.so era-zk_evm_amd/libzkw.so
for F in 1 16; do echo "fuse $F"; python bench.py --no-cpu-baseline --steps $((F*2)) --warmup $F --fuse $F 2>&1 | grep ZKW_PROFILE | tail -14; done
cp /tmp/libzkw_orig.so era-zk_evm_amd/libzkw.so
