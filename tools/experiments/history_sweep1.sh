cd $GRAFT_REPO_ROOT
SHARES="512 1024 4096 8192" bash tools/gpu_session.sh r06_hist3 chembl_all_share
mv gpurun_out/r06_hist3/chembl_all_share.txt gpurun_out/r06_hist3/share_sweep.txt
SHARES="2048" EXTRA_ENV="NVMK_BFGS_TEAM=656 NVMK_BFGS_HISTORY=1" bash tools/gpu_session.sh r06_hist3 chembl_all_share
mv gpurun_out/r06_hist3/chembl_all_share.txt gpurun_out/r06_hist3/team656_forced.txt
SHARES="1024" EXTRA_ENV="NVMK_BFGS_TEAM=500 NVMK_BFGS_HISTORY=1" bash tools/gpu_session.sh r06_hist3 chembl_all_share
mv gpurun_out/r06_hist3/chembl_all_share.txt gpurun_out/r06_hist3/team500_forced.txt
