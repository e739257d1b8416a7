set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
python bench.py --steps 20 --warmup 5 > gpurun_out/r02a/driver_std.json 2> gpurun_out/r02a/driver_std.err
python bench.py --steps 64 --warmup 16 --lanes 16 --no-cpu-baseline > gpurun_out/r02a/l16_std.json 2>> gpurun_out/r02a/driver_std.err
cp era-zk_evm_amd/libzkw.so /tmp/std.so
cp era-zk_evm_amd/libzkw_lb2.so era-zk_evm_amd/libzkw.so
python bench.py --steps 64 --warmup 16 --lanes 16 --no-cpu-baseline > gpurun_out/r02a/l16_lb2.json 2>> gpurun_out/r02a/driver_std.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02a/driver_lb2.json 2>> gpurun_out/r02a/driver_std.err
cp /tmp/std.so era-zk_evm_amd/libzkw.so
tail -c 600 gpurun_out/r02a/*.json
