cd $GRAFT_REPO_ROOT
SHARES="16384 65536" bash tools/gpu_session.sh r06_hist4 chembl_all_share
mv gpurun_out/r06_hist4/chembl_all_share.txt gpurun_out/r06_hist4/share_sweep.txt
SHARES="8192" EXTRA_ENV="NVMK_BFGS_TEAM=656 NVMK_BFGS_HISTORY=1" bash tools/gpu_session.sh r06_hist4 chembl_all_share
mv gpurun_out/r06_hist4/chembl_all_share.txt gpurun_out/r06_hist4/team656_forced.txt
