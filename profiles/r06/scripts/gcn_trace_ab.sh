# kernel trace of the config-4 GCN epoch loop (pubmed shape, hidden 128, plans on) in the current tree and in the round-5 tree
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
for tree in . tmp_r05; do
  name=$(echo $tree | tr -d './'); name=${name:-cur}
  (cd $tree && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gcntrace_$name -o t -- python examples/gcn_custom.py --dataset pubmed --n-hidden 128 --epochs 200 2>&1 | grep "epochs=")
  f=$(find /tmp/gcntrace_$name -name '*kernel_stats.csv' | head -1)
  echo "== tree=$tree $f"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6, "kernels", sum(int(r["Calls"]) for r in rows))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%8d calls %9.1f us avg %6.1f%%  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:110]))
PY
done > gpurun_out/r06/gcn_trace_ab.log 2>&1
cat gpurun_out/r06/gcn_trace_ab.log
