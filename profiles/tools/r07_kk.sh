cd $GRAFT_REPO_ROOT/profiles/tools/kk
hipcc --offload-arch=gfx950 -O3 -o /tmp/kk2 kk2.hip 2>/dev/null && /tmp/kk2
