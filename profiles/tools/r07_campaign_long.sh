# a longer differential campaign on the round's final kernel: other seed ranges than r03_campaigns.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
timeout 1500 python profiles/tools/fuzz_campaign.py 0x5000 ${2:-160} > $T/fuzz_campaign_long.txt 2>&1; tail -1 $T/fuzz_campaign_long.txt
timeout 300 python profiles/tools/precompile_campaign.py 0x5400 ${3:-160} > $T/precompile_campaign_long.txt 2>&1; tail -1 $T/precompile_campaign_long.txt
timeout 300 python profiles/tools/far_call_campaign.py 0x5800 ${4:-160} > $T/far_call_campaign_long.txt 2>&1; tail -1 $T/far_call_campaign_long.txt
