# per-instruction-class issue costs of a lone wave (profiles/tools/issue_probe.hip), one process per kernel
cd $GRAFT_REPO_ROOT
T=gpurun_out/$1; mkdir -p $T
hipcc --offload-arch=gfx950 -O3 profiles/tools/issue_probe.hip -o /tmp/issue_probe 2>/dev/null
for K in k_valu_dep k_valu_ind k_salu_dep k_readlane k_cmp_cnd k_carry k_branch k_cbranch_nt k_saveexec k_lds_dep k_lds128_dep k_gpridx k_mov8 k_gload_dep k_sload_dep k_code_8k k_code_32k k_code_128k; do
  timeout 60 /tmp/issue_probe $K >> $T/issue_probe.txt 2>&1 || echo "$K: failed" >> $T/issue_probe.txt
done
cat $T/issue_probe.txt
