# HIP API trace of the config-4 GCN epoch loop, current tree against the round-5 tree: which host calls differ
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
for tree in . tmp_r05; do
  name=$(echo $tree | tr -d './'); name=${name:-cur}
  (cd $tree && timeout 400 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/gcnhip_$name -o t -- python examples/gcn_custom.py --dataset pubmed --n-hidden 128 --epochs 200 2>&1 | grep "epochs=")
  f=$(find /tmp/gcnhip_$name -name '*hip_api_stats.csv' | head -1)
  echo "== tree=$tree $f"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total hip api ms", tot / 1e6, "calls", sum(int(r["Calls"]) for r in rows))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
    print("%8d calls %9.2f us avg %9.2f ms total  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Name"][:80]))
PY
done > gpurun_out/r06/gcn_hiptrace_ab.log 2>&1
cat gpurun_out/r06/gcn_hiptrace_ab.log
