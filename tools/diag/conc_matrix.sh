cd $GRAFT_REPO_ROOT
for kind in uniform lidar; do
for ch in 0 2 3 4; do
  echo -n "$kind chains=$ch: "
  python3 tools/diag/api_block.py $kind 100 --chains $ch 2>/dev/null | tail -n 1
done; done
