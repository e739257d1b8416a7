#!/bin/bash
# Static check of the cycle kernel's assembly before a change goes to the GPU (round 4: code layout moves the compiler's waits around —
# one edit put an `s_waitcnt vmcnt(0)` at the head of the group loop, i.e. every iteration waited for the cycle's stream stores):
# vmcnt waits in front of the group-loop head (must be 0), spill counts, v_readlane / scratch sites.
# usage: r06_asm_check.sh era-zk_evm_amd/csrc/zkw_kernels.hip [extra hipcc flags]
SRC=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -structurizecfg-skip-uniform-regions -I $(dirname $0)/../../include -I $(dirname $0)/../../era-zk_evm_amd/csrc "$@" -S --cuda-device-only -o /tmp/zkw_chk.s $SRC 2>/dev/null || { echo COMPILE FAILED; exit 1; }
python3 - <<'PY'
import re
ls=open('/tmp/zkw_chk.s').read().split('\n')
s=[i for i,l in enumerate(ls) if l.startswith('_Z16zkw_cycle_kernel')][0]
# group-loop head: the first readlane whose lane select is an SGPR (broadcast of the leader's instruction word)
for i in range(s,len(ls)):
    if re.search(r'v_readlane_b32 s\d+, v\d+, s\d+', ls[i]):
        blk=[l for l in ls[i-30:i+1] if l.strip() and not l.strip().startswith(';')]
        print('group-loop head at', i, 'vmcnt waits in the 30 lines before:', sum('s_waitcnt vmcnt' in l for l in blk)); break
PY
grep "sgpr_spill_count\|vgpr_spill_count\|private_segment_fixed_size:" /tmp/zkw_chk.s | head -3 | tr '\n' ' '; echo
S=$(grep -n "^_Z16zkw_cycle_kernel" /tmp/zkw_chk.s | cut -d: -f1); E=$(grep -n "^_Z16zkw_reset_kernel" /tmp/zkw_chk.s | cut -d: -f1)
echo "cycle kernel: vmcnt waits $(sed -n "${S},${E}p" /tmp/zkw_chk.s | grep -c 's_waitcnt vmcnt'), readlane $(sed -n "${S},${E}p" /tmp/zkw_chk.s | grep -c 'v_readlane'), writelane $(sed -n "${S},${E}p" /tmp/zkw_chk.s | grep -c 'v_writelane'), scratch $(sed -n "${S},${E}p" /tmp/zkw_chk.s | grep -c 'scratch_'), lines $((E-S))"
