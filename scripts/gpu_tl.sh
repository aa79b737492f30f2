cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export UPK_LIB=$GRAFT_REPO_ROOT/upgpt_amd/libupk_dev.so
python scripts/timeline_mlp.py 8192 224 32 2>&1 | grep "first w0"
