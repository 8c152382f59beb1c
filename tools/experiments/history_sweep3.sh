cd $GRAFT_REPO_ROOT
bash tools/gpu_session.sh r06_hist5 team_tests
SHARES="1024 2048 4096 8192" bash tools/gpu_session.sh r06_hist5 chembl_all_share
