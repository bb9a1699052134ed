# config-4 GCN, plans on: is the current-vs-round-5 gap per epoch or a one-time cost, and does it follow the library or the Python layer?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
rm -rf /tmp/mixD && cp -r tmp_r05 /tmp/mixD && cp gespmm_amd/lib/libgespmm.so gespmm_amd/lib/_gespmm_torch*.so /tmp/mixD/gespmm_amd/lib/
{
for rep in 1 2; do
  for ep in 49 200 800; do
    for tree in tmp_r05 . /tmp/mixD; do
      echo -n "rep=$rep tree=$tree epochs=$ep: "; (cd $tree && timeout 300 python examples/gcn_custom.py --dataset pubmed --n-hidden 128 --epochs $ep 2>&1 | grep "epochs=")
    done
  done
done
} > gpurun_out/r06/gcn_ab2.log 2>&1
cat gpurun_out/r06/gcn_ab2.log
