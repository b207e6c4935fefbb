cd $GRAFT_REPO_ROOT
python tools/diag_x3_grouped.py 2>&1 | grep -v amdgpu.ids
