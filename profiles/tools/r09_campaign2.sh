# round 5, second pass with other seed ranges (unused GPU minutes): shared tapes through the short cycle, divergent tapes, far calls
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
timeout 1500 python profiles/tools/fuzz_campaign.py 0x7000 ${2:-600} uniform > $T/uniform_fuzz_campaign2.txt 2>&1; tail -1 $T/uniform_fuzz_campaign2.txt
timeout 1500 python profiles/tools/fuzz_campaign.py 0x7800 ${3:-160} > $T/fuzz_campaign2.txt 2>&1; tail -1 $T/fuzz_campaign2.txt
timeout 300 python profiles/tools/far_call_campaign.py 0x7c00 ${4:-300} > $T/far_call_campaign2.txt 2>&1; tail -1 $T/far_call_campaign2.txt
timeout 300 python profiles/tools/precompile_campaign.py 0x7e00 ${5:-200} > $T/precompile_campaign2.txt 2>&1; tail -1 $T/precompile_campaign2.txt
