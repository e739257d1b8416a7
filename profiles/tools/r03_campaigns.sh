# differential campaigns on the final kernel of a round (r03, r06): fuzz tapes (traces + commitments), precompile workloads, far-call plans with slot reuse + page read-back
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
python profiles/tools/fuzz_campaign.py 0x4000 ${2:-60} > $T/fuzz_campaign.txt 2>&1; tail -2 $T/fuzz_campaign.txt
python profiles/tools/precompile_campaign.py 0x4400 ${3:-40} > $T/precompile_campaign.txt 2>&1; tail -1 $T/precompile_campaign.txt
python profiles/tools/far_call_campaign.py 0x4800 ${4:-40} > $T/far_call_campaign.txt 2>&1; tail -1 $T/far_call_campaign.txt
