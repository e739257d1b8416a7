# process-to-process spread of the driver's command on one box, with the engine / memory clocks sampled while it runs
cd $GRAFT_REPO_ROOT
T=gpurun_out/$1; mkdir -p $T
for i in 1 2 3 4 5 6; do
  ( for k in 1 2 3 4 5 6 7 8 9 10 11 12; do rocm-smi --showclocks 2>/dev/null | grep -i 'sclk\|mclk\|fclk' | head -3 | tr '\n' ' '; echo; sleep 1; done ) > $T/clocks_$i.txt 2>&1 &
  SMI=$!
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('run $i', round(j['value']/1e9,3), 'kernel_ms', round(j['kernel_ms'],4))" | tee -a $T/runs.txt
  wait $SMI
  sort $T/clocks_$i.txt | uniq -c | sort -rn | head -3
done
rocm-smi --showpower --showtemp 2>/dev/null | grep -i 'power\|temp' | head -6
