cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O; rm -f $O/gcn_epochs.log
for ds in pubmed com-amazon-sbm reddit-like reddit-sbm; do
  for extra in "" "--no-plans" "--graph-capture"; do
    echo "== $ds hidden 128 epochs 100 $extra" >> $O/gcn_epochs.log
    timeout 600 python examples/gcn_custom.py --dataset $ds --n-hidden 128 --epochs 100 $extra 2>&1 | grep -v "amdgpu\|^W2026" | tail -2 >> $O/gcn_epochs.log
  done
done
cat $O/gcn_epochs.log
