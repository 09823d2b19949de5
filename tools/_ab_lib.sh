set -e
cd "$GRAFT_REPO_ROOT"
for lib in "" build/variants/hmono.so build/variants/hmonopls.so ""; do
  echo "== lib=${lib:-default}"
  DES_HIP_LIB=$lib python tools/time_shard.py 200 --ranks 1,8 --kernels 2>&1 | grep -v "^\[W\|Gloo\|amdgpu.ids"
done
