cd $GRAFT_REPO_ROOT
SHARES="2048 4096 8192 16384" bash tools/gpu_session.sh r06_hist17 chembl_all_share
