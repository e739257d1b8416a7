# round 5, third pass (unused GPU minutes): 1200 further shared-tape seeds through the short cycle, 400 far-call plans
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
timeout 2400 python profiles/tools/fuzz_campaign.py 0x9000 ${2:-1200} uniform > $T/uniform_fuzz_campaign3.txt 2>&1; tail -1 $T/uniform_fuzz_campaign3.txt
timeout 300 python profiles/tools/far_call_campaign.py 0x9800 ${3:-400} > $T/far_call_campaign3.txt 2>&1; tail -1 $T/far_call_campaign3.txt
