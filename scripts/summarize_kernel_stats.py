"""Aggregates a rocprofv3 kernel_stats CSV by short kernel name: python scripts/summarize_kernel_stats.py file.csv [top]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time %.3f ms, %d launches" % (tot / 1e6, sum(int(r['Calls']) for r in rows)))
def short(n):
    m = re.search(r'gespmm::(?:\(anonymous namespace\)::)?(\w+(<[^>(]*>)?)', n)
    if m: return m.group(1)
    for k in ('radix_sort_onesweep', 'merge_sort_block_merge', 'radix_sort_block_sort', 'merge_mergepath', 'scan_impl', 'init_lookback',
              'lookback_scan_state', 'transform', 'onesweep_histograms', 'copyBuffer', 'fillBuffer', 'histogram'):
        if k in n: return 'lib/' + k
    return n[:50]
agg = {}
for r in rows:
    a = agg.setdefault(short(r['Name']), [0, 0.0]); a[0] += int(r['Calls']); a[1] += float(r['TotalDurationNs'])
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
    print("%-44s calls %6d  total %9.3f ms  avg %9.1f us" % (k, c, t / 1e6, t / c / 1e3))
