# round 5: differential campaigns on the final kernel — shared tapes through the short cycle (synth.uniform_fuzz), then the divergent ones
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
timeout 1500 python profiles/tools/fuzz_campaign.py 0x6000 ${2:-240} uniform > $T/uniform_fuzz_campaign.txt 2>&1; tail -1 $T/uniform_fuzz_campaign.txt; grep -c MISMATCH $T/uniform_fuzz_campaign.txt
timeout 1200 python profiles/tools/fuzz_campaign.py 0x6400 ${3:-120} > $T/fuzz_campaign.txt 2>&1; tail -1 $T/fuzz_campaign.txt
timeout 300 python profiles/tools/precompile_campaign.py 0x6800 ${4:-80} > $T/precompile_campaign.txt 2>&1; tail -1 $T/precompile_campaign.txt
timeout 300 python profiles/tools/far_call_campaign.py 0x6c00 ${5:-120} > $T/far_call_campaign.txt 2>&1; tail -1 $T/far_call_campaign.txt
