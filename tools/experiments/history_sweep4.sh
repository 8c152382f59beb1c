cd $GRAFT_REPO_ROOT
SHARES="4096" bash tools/gpu_session.sh r06_hist16 chembl_all_share
mv gpurun_out/r06_hist16/chembl_all_share.txt gpurun_out/r06_hist16/default.txt
SHARES="4096" EXTRA_ENV="NVMK_BFGS_TEAM=656 NVMK_BFGS_HISTORY=1" bash tools/gpu_session.sh r06_hist16 chembl_all_share
mv gpurun_out/r06_hist16/chembl_all_share.txt gpurun_out/r06_hist16/team656_forced.txt
SHARES="4096" EXTRA_ENV="NVMK_BFGS_TEAM=400 NVMK_BFGS_HISTORY=1" bash tools/gpu_session.sh r06_hist16 chembl_all_share
mv gpurun_out/r06_hist16/chembl_all_share.txt gpurun_out/r06_hist16/team400_forced.txt
SHARES="8192" EXTRA_ENV="NVMK_BFGS_TEAM=500 NVMK_BFGS_HISTORY=1" bash tools/gpu_session.sh r06_hist16 chembl_all_share
mv gpurun_out/r06_hist16/chembl_all_share.txt gpurun_out/r06_hist16/team500_forced_share8192.txt
