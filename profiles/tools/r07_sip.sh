cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r07
hipcc --offload-arch=gfx950 -O3 profiles/tools/store_issue_probe.hip -o /tmp/sip 2>/dev/null && timeout 120 /tmp/sip | tee gpurun_out/r07/store_issue_probe.txt
