# the GPU suite + the driver / default bench lines + cfg 3 (lone batch, 128 fused) on libzkw.so
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
timeout 2400 python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; tail -4 $T/pytest.log | head -3
for A in "--steps 20 --warmup 5" ""; do python bench.py $A --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('[$A] value G', round(j['value']/1e9,2), 'ms/step', round(j['ms_per_step'],4), 'kernel_ms', round(j['kernel_ms'],3), 'frac', round(j['roofline']['frac'],3))" | tee -a $T/bench.txt; done
for F in 1 128; do python bench.py --no-cpu-baseline --cfg 3 --commit-mask 0 --fuse $F --steps $((F*2)) --warmup $F --streams 1 2>&1 | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print('cfg3 fuse', j['config']['batches_per_fused_launch'], 'ms/step', round(j['ms_per_step'],3), 'kernel_ms', round(j['kernel_ms'],3), 'msg GB/s', round(j['value']/1e9,1), 'keccak-f/s %.3g' % r['keccak_f_per_s'])" | tee -a $T/bench.txt; done
python profiles/tools/precompile_campaign.py 0x3400 16 2>&1 | tail -1
