cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tree in . tmp_r05; do
  for extra in "" "--no-plans"; do
    echo -n "tree=$tree $extra: "; (cd $tree && timeout 300 python examples/gcn_custom.py --dataset pubmed --n-hidden 128 --epochs 200 $extra 2>&1 | grep "epochs=")
  done
done
done
