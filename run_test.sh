#!/bin/bash
# The reference's run_test.sh (run_test.sh:1-29) for this build: run the spmm_test driver
# over every ./data/snap/<name>/<name>.mtx (drop SNAP files there; none can be downloaded
# here) and over the bundled citation graphs, one CSV line per matrix in spmm_test_out.out:
#   data,K=128-vendor,K=128-gespmm,K=256-vendor,K=256-gespmm,K=512-vendor,K=512-gespmm,
# (the vendor column is rocSPARSE where the reference has cuSPARSE). The merge-spmm
# (GraphBLAST) baseline of the reference's script is an un-vendored third-party tree and
# is not run. Build first:  python -c "import __graft_entry__ as g; g.build()"
device=${1:-0}
here="$(cd "$(dirname "$0")" && pwd)"
drv="$here/gespmm_amd/lib/spmm_test"
rm -f spmm_test_out.out
echo "data,K=128-rocsparse-gflops,K=128-gespmm-gflops,K=256-rocsparse-gflops,K=256-gespmm-gflops,K=512-rocsparse-gflops,K=512-gespmm-gflops," >> spmm_test_out.out
for i in ./data/snap/*/; do
    [ -d "$i" ] || continue
    ii=$(basename "$i")
    [ -f "./data/snap/${ii}/${ii}.mtx" ] || continue
    echo -n "$ii," >> spmm_test_out.out
    "$drv" "./data/snap/${ii}/${ii}.mtx" "$device"
    echo >> spmm_test_out.out
done
for i in "$here"/tests/golden/*.mtx ./data/misc/*.mtx; do
    [ -f "$i" ] || continue
    echo -n "$(basename "$i" .mtx)," >> spmm_test_out.out
    "$drv" "$i" "$device"
    echo >> spmm_test_out.out
done
cat spmm_test_out.out
