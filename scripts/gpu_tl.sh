cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_astat_gpu.py -x -q 2>&1 | tail -2
python scripts/ws_bench.py ff1_M8192 qkv_M8192 ff1_M2048 qkv_M2048 ff1_M512 2>&1 | grep -E "^==|as[0-9]|best" | awk '/^==/{n=0} {if(/^==/||/best/||n<3)print; n++}'
export UPK_LIB=$GRAFT_REPO_ROOT/upgpt_amd/libupk_dev.so
for c in "as8x2p7 2" "as4x2p7 4"; do
echo "=== $c"; python scripts/timeline_as.py 8192 224 1792 $c geglu 2>&1 | grep -v amdgpu.ids | grep "first w0" | tail -3
done
