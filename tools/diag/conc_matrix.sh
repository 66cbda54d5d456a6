cd $GRAFT_REPO_ROOT
for q in 4 8 16; do
for ch in 4 8 9; do
  echo -n "graph queues=$q chains=$ch: "
  DEBUG_HIP_FORCE_GRAPH_QUEUES=$q python3 tools/diag/api_block.py uniform 100 --chains $ch | tail -n 1
done; done
