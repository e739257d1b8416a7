#!/bin/bash
# Round 6 (no GPU): is the gfx950 code of this tree the code a GPU has already run?  Compiles every HIP translation unit of a git
# revision and of the working tree to assembly (the build's own flags, device pass only) and compares them, ignoring the
# compilation-unit id (a hash of the source text).  usage: profiles/tools/r10_asm_identity.sh <git revision>   (run from the repo root)
REV=${1:-7a62963}
T=$(mktemp -d)
git archive $REV era-zk_evm_amd/csrc include | tar -x -C $T
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -mllvm -structurizecfg-skip-uniform-regions -S --cuda-device-only"
echo "device assembly: working tree ($(git rev-parse --short HEAD)) against $REV"
for f in zkw_kernels zkw_commit zkw_expand zkw_blake2s zkw_pack; do
  hipcc $FLAGS -I $T/include -I $T/era-zk_evm_amd/csrc -o $T/$f.old.s $T/era-zk_evm_amd/csrc/$f.hip 2>/dev/null &
  hipcc $FLAGS -I include -I era-zk_evm_amd/csrc -o $T/$f.new.s era-zk_evm_amd/csrc/$f.hip 2>/dev/null &
done
wait
for f in zkw_kernels zkw_commit zkw_expand zkw_blake2s zkw_pack; do
  n=$(diff <(grep -v __hip_cuid $T/$f.old.s) <(grep -v __hip_cuid $T/$f.new.s) | grep -c '^[<>]')
  lines=$(grep -vc '^\s*;' $T/$f.new.s)
  if [ "$n" = "0" ]; then echo "  $f.hip: IDENTICAL ($lines lines of assembly)"; else echo "  $f.hip: differs in $n lines"; fi
done
# the host pass of the same translation units (hipcc compiles every .hip twice: a tree whose device pass is fine can still fail to build)
for f in zkw_kernels zkw_commit zkw_expand zkw_blake2s zkw_pack; do
  if hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c --cuda-host-only -I include -I era-zk_evm_amd/csrc -o $T/$f.host.o era-zk_evm_amd/csrc/$f.hip 2>$T/$f.host.log; then echo "  $f.hip: host pass compiles"; else echo "  $f.hip: HOST PASS FAILS"; grep " error" $T/$f.host.log | head -3; fi
done
echo "cycle kernel code object (working tree):"
awk '/\.name: *_Z16zkw_cycle_kernel/{f=1} f&&/sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size|\.vgpr_count|\.sgpr_count|agpr_count/{print "  " $0} /\.name:/{if(f&&!/zkw_cycle_kernel/)f=0}' $T/zkw_kernels.new.s | sort -u
rm -rf $T
