"""The plan's POLICY (csrc/plan_policy.cpp) asked through gespmm_plan_policy — host only, no device: one table of
(shape, analysis numbers) -> decisions for the BASELINE shapes and the hold-out graphs of profiles/r04/holdout_audit.log.
A threshold that moves shows up here as a changed row, with the log that moved it."""
import pytest

from gespmm_amd import _lib

SPLIT, STRICT, SHALLOW = _lib.FLAG_SPLIT_LONG_ROWS, _lib.FLAG_STRICT_ORDER, _lib.FLAG_SHALLOW_UNROLL

# name, (M, nnz, N, max_degree, hits_before, hits_after, staged_fraction), expected subset of the answer
TABLE = [
    # ---- BASELINE configs (the stand-ins' measured analysis numbers)
    ("C2a com-amazon-sbm N=128", (334863, 1851744, 128, 120, 0.018, 0.651, 0.750),  # round 5: staged-rows on short rows too (92.7 vs 107 us)
     dict(analyse=1, dense_try=0, keep_clustered=1, task_entries=40, group_task_entries=16, build_staged=1, keep_staged=1, shallow_unroll=1,
          segmented=0, launch_flags=STRICT, sddmm_route=2, narrow_vec4=0)),
    ("planted communities, mean degree 6, N=128: share 0.527 on short rows wins x1.09 (the record stream, not the LDS)", (600000, 3600000, 128, 40, 0.01, 0.60, 0.527),
     dict(keep_clustered=1, build_staged=1, keep_staged=1)),
    ("planted communities, mean degree 4, N=128: share 0.448 x1.05", (600000, 2400000, 128, 30, 0.01, 0.55, 0.448), dict(build_staged=1, keep_staged=1)),
    ("planted communities, mean degree 3, N=128: level, not built", (600000, 1800000, 128, 30, 0.01, 0.55, 0.0), dict(build_staged=0)),
    ("planted communities, mean degree 4, N=256: level, not built", (600000, 2400000, 256, 30, 0.01, 0.55, 0.0), dict(build_staged=0)),
    ("C2a com-amazon-like N=128", (334863, 1851744, 128, 499, 0.025, 0.176, 0.0),
     dict(analyse=1, keep_clustered=1, task_entries=40, build_staged=0, shallow_unroll=0, segmented=0, sddmm_route=1)),
    ("C2a com-amazon-sbm N=32", (334863, 1851744, 32, 120, 0.09, 0.70, 0.0),
     dict(analyse=1, keep_clustered=1, task_entries=160, shallow_unroll=1, segmented=0, narrow_vec4=1)),  # V = 4: 37.1 vs 44.6 us; 160 entries: 34.9 vs 36.7
    ("C2a com-amazon-like N=32 (misses: one lane per column)", (334863, 1851744, 32, 499, 0.092, 0.222, 0.0), dict(keep_clustered=1, narrow_vec4=0)),
    ("LFR mu=0.1 N=32 (rows of 16: one lane per column)", (300000, 4717400, 32, 306, 0.144, 0.786, 0.0), dict(keep_clustered=1, narrow_vec4=0, segmented=0)),
    ("C2a com-amazon-sbm N=512", (334863, 1851744, 512, 120, 0.01, 0.60, 0.0),
     dict(analyse=1, keep_clustered=1, task_entries=32, shallow_unroll=0, build_staged=1)),  # 256-column tiles since round 4
    ("C2a id-local storage order", (334863, 1851744, 128, 499, 0.60, 0.62, 0.0), dict(analyse=1, keep_clustered=0, segmented=0)),
    ("C2b reddit-like N=128 (dense, no structure)", (232965, 114615892, 128, 21000, 0.10, 0.30, 0.0),
     dict(analyse=1, dense_try=1, keep_clustered=0)),
    ("C2b reddit-sbm N=128 (dense, communities)", (232965, 114615892, 128, 2000, 0.10, 0.72, 0.0),
     dict(analyse=1, dense_try=1, keep_clustered=1, segmented=1)),
    ("C3 products-sbm N=128", (2449029, 123718280, 128, 1446, 0.003, 0.840, 0.638),
     dict(analyse=1, dense_try=0, keep_clustered=1, task_entries=255, build_staged=1, keep_staged=1, segmented=0, model_sample=1 << 22)),
    # round 5: the lane-group form of the staged kernel at N = 32 / 64 (1185 vs 1302 us, 1865 vs 2089 us)
    ("C3 products-sbm N=32", (2449029, 123718280, 32, 1446, 0.01, 0.85, 0.707), dict(keep_clustered=1, build_staged=1, keep_staged=1, segmented=0)),
    ("C3 products-sbm N=64", (2449029, 123718280, 64, 1446, 0.01, 0.85, 0.651), dict(keep_clustered=1, build_staged=1, keep_staged=1, segmented=0)),
    ("C3 products-sbm N=16: streaming kernels", (2449029, 123718280, 16, 1446, 0.01, 0.85, 0.0), dict(keep_clustered=1, build_staged=0)),
    ("geometric N=32: share 0.93", (600000, 7175884, 32, 30, 0.02, 0.93, 0.931), dict(build_staged=1, keep_staged=1)),
    ("small-world N=32 through a default-life plan: share 0.844 wins (~125 vs 153 us)", (1000000, 11001376, 32, 18, 0.019, 0.807, 0.844), dict(build_staged=1, keep_staged=1)),
    ("small-world N=64 through a default-life plan: share 0.831", (1000000, 11001376, 64, 18, 0.011, 0.805, 0.831), dict(build_staged=1, keep_staged=1)),
    ("LFR mu=0.1 N=32: share 0.71 on rows of 16 loses (105 vs 69 us)", (300000, 4717400, 32, 306, 0.144, 0.786, 0.709), dict(build_staged=1, keep_staged=0)),
    ("com-amazon-sbm N=64: short rows stay with the streaming kernels", (334863, 1851744, 64, 120, 0.05, 0.68, 0.0), dict(build_staged=0)),
    ("C3 products-sbm N=512", (2449029, 123718280, 512, 1446, 0.001, 0.833, 0.503), dict(keep_clustered=1, build_staged=1, keep_staged=1, task_entries=102)),
    # ---- hold-out graphs (profiles/r04/holdout_audit.log): the rows that moved thresholds in round 4
    ("LFR mu=0.1 N=128: share 0.552 with 112-row blocks wins 3-6 % (round 4's walk lost 21 % at 0.565)", (300000, 4717400, 128, 306, 0.041, 0.768, 0.552),
     dict(keep_clustered=1, build_staged=1, keep_staged=1, segmented=0)),
    ("LFR mu=0.3 N=128: share 0.409 loses 17 %", (300000, 4759166, 128, 305, 0.041, 0.534, 0.409), dict(build_staged=1, keep_staged=0)),
    ("LFR mu=0.1 N=256: share 0.44 level (x0.99; x0.93 at 512) without the nt marks", (300000, 4717400, 256, 306, 0.021, 0.763, 0.440), dict(build_staged=1, keep_staged=1)),
    ("LFR mu=0.3 N=256: share 0.33 loses 13 % staged", (300000, 4759166, 256, 305, 0.021, 0.521, 0.332), dict(build_staged=1, keep_staged=0, segmented=1)),
    ("geometric N=128: share 0.94 wins", (600000, 7175884, 128, 30, 0.010, 0.910, 0.937), dict(build_staged=1, keep_staged=1)),
    ("small-world N=256: share 0.77 wins", (1000000, 11001376, 256, 18, 0.003, 0.815, 0.767), dict(build_staged=1, keep_staged=1)),
    ("dense LFR mu=0.3 N=128: batch kernel at 0.52 hits", (300000, 15383642, 128, 619, 0.035, 0.519, 0.27), dict(keep_staged=0, segmented=0)),
    ("dense LFR mu=0.3 N=32", (300000, 15383642, 32, 619, 0.136, 0.572, 0.0), dict(segmented=0)),
    ("dense LFR mu=0.3 N=256: segmented at mid hit rates", (300000, 15383642, 256, 619, 0.018, 0.487, 0.196), dict(keep_staged=0, segmented=1)),
    ("dense LFR mu=0.3 N=512: segmented at mid hit rates", (300000, 15383642, 512, 619, 0.018, 0.487, 0.196), dict(keep_staged=0, segmented=1)),
    ("LFR mu=0.5 N=256: segmented (mean degree 15.9)", (300000, 4777356, 256, 309, 0.021, 0.269, 0.199), dict(keep_staged=0, segmented=1)),
    ("LFR mu=0.1 N=256 with a low share: batch kernel at 0.76 hits", (300000, 4717400, 256, 306, 0.021, 0.763, 0.30), dict(keep_staged=0, segmented=0)),
    ("Holme-Kim m=5 N=128 (hubs: long-row pass)", (500000, 4999852, 128, 8968, 0.055, 0.346, 0.341), dict(keep_clustered=1, keep_staged=0, launch_flags=SPLIT)),
    ("Barabasi-Albert N=32: +0.047 modelled hits is worth the order", (500000, 5999928, 32, 2726, 0.112, 0.159, 0.0), dict(analyse=1, keep_clustered=1)),
    ("very dense LFR mu=0.5 N=32: no cache-blocked path at this width, judged like a sparse graph", (100000, 33022008, 32, 1626, 0.312, 0.453, 0.0),
     dict(analyse=1, dense_try=0, keep_clustered=1)),
    ("very dense LFR mu=0.5 N=128: 0.23 modelled hits against the cache-blocked path", (100000, 33022008, 128, 1626, 0.080, 0.226, 0.0),
     dict(analyse=1, dense_try=1, keep_clustered=0)),
    ("very dense LFR mu=0.2 N=64: 0.68", (100000, 32419866, 64, 1597, 0.157, 0.684, 0.0), dict(analyse=1, keep_clustered=1)),
    ("skewed RMAT: storage order already hits", (524288, 12582912, 128, 181863, 0.558, 0.559, 0.0), dict(analyse=1, keep_clustered=0)),
    ("C3 products-like N=128 (no structure)", (2449029, 123718280, 128, 30000, 0.003, 0.02, 0.0),
     dict(analyse=1, keep_clustered=0, launch_flags=SPLIT)),
    ("C1 cit-hepth N=32 (small: B fits the L2s)", (27770, 352807, 32, 2000, 0.0, 0.0, 0.0), dict(analyse=0, keep_clustered=0, launch_flags=STRICT)),
    ("C4 pubmed N=128", (19717, 108365, 128, 172, 0.1, 0.5, 0.0), dict(analyse=1, keep_clustered=1, shallow_unroll=0, build_staged=0)),
    ("C5 rmat-26 shard N=256", (8388608, 134217728, 256, 400000, 0.0, 0.0, 0.0), dict(launch_flags=SPLIT)),
    # ---- the long-row pass follows the plain call: never below 2^20 entries / mean degree 8 even with a hub row
    ("hub row in a small matrix", (20000, 100000, 128, 9000, 0.0, 0.0, 0.0), dict(launch_flags=STRICT)),
    ("hub row, 2^20 entries, mean degree 10", (500000, 4999852, 128, 8968, 0.055, 0.346, 0.0), dict(launch_flags=SPLIT, keep_clustered=1)),
]


@pytest.mark.parametrize("name,shape,expect", TABLE, ids=[t[0] for t in TABLE])
def test_policy_table(name, shape, expect):
    M, nnz, N, maxdeg, hb, ha, sf = shape
    # (steady state — enough launches for any analysis to pay: this table pins the structure and kernel rules; the cost rule has its own)
    got = _lib.plan_policy(M, M, nnz, N, maxdeg, hb, ha, sf, expected_launches=1000000)
    for k, v in expect.items():
        if k == "launch_flags":
            assert got[k] & (SPLIT | STRICT) == v, (name, k, got)
        else:
            assert got[k] == v, (name, k, got)


# The cost rule (round 5, plan_policy.cpp: estimate_analysis_cost): name, (M, nnz, N, wedge probe, expected launches) -> analysed or skipped.
# Probes as measured by scripts/probe_calibration.py (profiles/r05/probe_calibration.log).
COST_TABLE = [
    ("com-amazon-sbm N=128, 200 launches: 55 us x 200 > 4.3 ms", (334863, 1851744, 128, 0.416, 0), dict(analyse=1, cost_skipped=0)),
    ("com-amazon-like N=128, 200 launches: a structureless graph does not pay (was -16 % in round 4)", (334863, 1851744, 128, 0.0001, 0),
     dict(analyse=0, cost_skipped=1)),
    ("com-amazon-like N=128, 2000 launches: it does", (334863, 1851744, 128, 0.0001, 2000), dict(analyse=1, cost_skipped=0)),
    ("com-amazon-sbm N=32, 200 launches: 14 us x 200 < 4.3 ms", (334863, 1851744, 32, 0.416, 0), dict(analyse=0, cost_skipped=1)),
    ("com-amazon-sbm N=32, 1000 launches", (334863, 1851744, 32, 0.416, 1000), dict(analyse=1, cost_skipped=0)),
    ("pubmed N=128, 200 launches: the reference's GCN (was +52 % per epoch in round 4)", (19717, 108365, 128, 0.114, 0), dict(analyse=0, cost_skipped=1)),
    ("pubmed N=128, rectangular / unknown probe", (19717, 108365, 128, -1.0, 0), dict(analyse=0, cost_skipped=1)),
    ("pubmed N=128, 10 000 launches", (19717, 108365, 128, 0.114, 10000), dict(analyse=1, cost_skipped=0)),
    ("products-sbm N=128, 200 launches", (2449029, 123718280, 128, 0.320, 0), dict(analyse=1, cost_skipped=0)),
    ("products-like N=128, 200 launches: 640 us x 200 > 80 ms — the analysis runs (and finds nothing)", (2449029, 123718280, 128, 0.0004, 0),
     dict(analyse=1, cost_skipped=0)),
    ("geometric N=128", (600000, 7175884, 128, 0.583, 0), dict(analyse=1, cost_skipped=0)),
    ("LFR mu=0.3 N=128", (300000, 4759166, 128, 0.0615, 0), dict(analyse=1, cost_skipped=0)),
    ("Barabasi-Albert N=128, 200 launches: 30 us x 200 < 6.6 ms (measured gain of its plan: 16 us; measured cost 10 ms)", (500000, 5999928, 128, 0.0004, 0),
     dict(analyse=0, cost_skipped=1)),
    ("unknown probe (rectangular), com-Amazon-sized, N=128: the benefit of the doubt", (334863, 1851744, 128, -1.0, 0), dict(analyse=1, cost_skipped=0)),
]


@pytest.mark.parametrize("name,shape,expect", COST_TABLE, ids=[t[0] for t in COST_TABLE])
def test_cost_rule(name, shape, expect):
    M, nnz, N, probe, launches = shape
    got = _lib.plan_policy(M, M, nnz, N, 100, wedge_probe=probe, expected_launches=launches)
    for k, v in expect.items():
        assert got[k] == v, (name, k, got)
    assert got["est_cost_us"] > 1000 and got["est_gain_us"] > 0
    # an explicit order is never second-guessed, and the 0.2 entry point (no probe, default launches) still answers
    assert _lib.plan_policy(M, M, nnz, N, 100, wedge_probe=probe, expected_launches=launches, reorder=_lib.PLAN_REORDER)["analyse"] == 1
    import ctypes

    q = _lib.PlanPolicyQuery(M, M, nnz, N, 0, -1, 100, 0, 0, 0, 0, 0, 0, 0.0, 0.5, 0.0, 12345, 0, 0.9)  # (the 0.2 symbol must not read the tail)
    a = _lib.PlanPolicyAnswer()
    a.cost_skipped = 77
    assert _lib.lib.gespmm_plan_policy(ctypes.byref(q), ctypes.byref(a)) == 0
    assert a.cost_skipped == 77, "the 0.2 symbol writes the 0.2 answer only"


def test_clustering_effort_follows_the_expected_launches():
    """Plans with a short life cluster three levels deep with three sweeps each (profiles/r05/cluster_sweeps.log, plan_life_compare.log:
    the launch is 2-4 % slower, the analysis 3.4 ms shorter); the launches it takes to pay for the depth shrink with the width, so the
    switch is launches x N >= 100 000; deep plans take the clustering's defaults (six / five)."""
    def effort(N, launches, **kw):
        got = _lib.plan_policy(334863, 334863, 1851744, N, 100, wedge_probe=0.42, expected_launches=launches, **kw)
        return got["analyse"], got["cluster_levels"], got["cluster_sweeps"]
    assert effort(128, 0) == (1, 3, 0)  # the default, 200 launches: three levels, all five sweeps (levels_vs_sweeps.log)
    assert effort(128, 781) == (1, 3, 0) and effort(128, 782) == (1, 0, 0)
    assert effort(512, 195) == (1, 3, 0) and effort(512, 200) == (1, 0, 0)
    assert effort(32, 3000, reorder=_lib.PLAN_REORDER) == (1, 3, 0) and effort(32, 3125) == (1, 0, 0)
    assert effort(128, 99, reorder=_lib.PLAN_REORDER) == (1, 3, 3) and effort(128, 100, reorder=_lib.PLAN_REORDER) == (1, 3, 0)
    none = _lib.plan_policy(19717, 19717, 108365, 128, 100, wedge_probe=0.11)
    assert (none["analyse"], none["cluster_levels"], none["cluster_sweeps"]) == (0, 0, 0)


def test_rows_per_staged_block_follow_the_mean_degree():
    """profiles/r05/staged_rows_per_block.log: short rows want more rows per block (com-Amazon-shaped 88.0 us at 112 rows against 91.3 at
    96), long rows fewer (products-shaped at 256 columns: 5.27 ms at 48 rows against 5.50 at 64)."""
    rows = lambda M, nnz, N: _lib.plan_policy(M, M, nnz, N, 100, hits_after=0.7)["staged_rows"]
    assert rows(334863, 1851744, 128) == 112 and rows(334863, 1851744, 256) == 64 and rows(334863, 1851744, 512) == 64
    assert rows(2449029, 123718280, 128) == 96 and rows(2449029, 123718280, 256) == 48
    assert rows(600000, 7175884, 128) == 112  # (mean degree 12: level between 96 and 112)
    assert rows(334863, 1851744, 32) == 512 and rows(334863, 1851744, 64) == 256
    # round 6: every other width has the general kernel's blocks — the rules of its tile class (two floats per lane: 128-column tiles,
    # four: 256-column tiles), 192 rows where a lane holds one float (odd widths, widths up to 64)
    assert rows(334863, 1851744, 100) == 112 and rows(2449029, 123718280, 100) == 96
    assert rows(334863, 1851744, 200) == 64 and rows(2449029, 123718280, 200) == 48 and rows(334863, 1851744, 602) == 112
    assert rows(334863, 1851744, 48) == 192 and rows(334863, 1851744, 41) == 192


def test_auto_considers_the_staged_kernel_at_even_widths_beyond_64():
    """Round 6 (csrc/spmm_staged_gen.hip): AUTO builds staged tables at N = 100 / 200 / 602 under the rules of N = 128 / 256, not at odd
    widths or below 65 columns (those are served on request: kernel = staged)."""
    q = lambda N, **kw: _lib.plan_policy(334863, 334863, 1851744, N, 120, 0.018, 0.651, 0.75, **kw)
    for N in (100, 128, 200, 256, 602):
        assert q(N)["build_staged"] == 1 and q(N)["keep_staged"] == 1, N
    for N in (41, 47, 65, 129, 48, 20):
        assert q(N)["build_staged"] == 0, N
        assert q(N, kernel=_lib.PLAN_KERNEL_STAGED, reorder=_lib.PLAN_REORDER)["build_staged"] == 1, N
    # the share thresholds of the tile class: 0.55 (0.42 on short rows) at two floats per lane, 0.42 at four
    lfr = lambda N, share: _lib.plan_policy(500000, 500000, 8000000, N, 300, 0.02, 0.6, share)["keep_staged"]
    assert lfr(100, 0.50) == 0 and lfr(100, 0.56) == 1 and lfr(200, 0.43) == 1 and lfr(200, 0.40) == 0


def test_policy_respects_the_callers_choices():
    base = (334863, 334863, 1851744, 128, 120, 0.018, 0.651, 0.0)
    assert _lib.plan_policy(*base, reorder=_lib.PLAN_NO_REORDER)["analyse"] == 0
    assert _lib.plan_policy(*base, kernel=_lib.PLAN_KERNEL_SEG_STREAM)["segmented"] == 1
    assert _lib.plan_policy(*base, kernel=_lib.PLAN_KERNEL_STREAM)["segmented"] == 0
    forced = _lib.plan_policy(*base, kernel=_lib.PLAN_KERNEL_STAGED)
    assert forced["build_staged"] == 1 and forced["keep_staged"] == 1  # an explicit choice is not second-guessed
    assert _lib.plan_policy(*base, kernel=_lib.PLAN_KERNEL_STAGED, analysis=_lib.PLAN_ANALYSIS_HOST)["build_staged"] == 0
    assert _lib.plan_policy(*base, task_entries=24)["task_entries"] == 24
    assert _lib.plan_policy(*base, task_entries=24)["group_task_entries"] == 12
    assert _lib.plan_policy(*base, flags=_lib.FLAG_SPLIT_LONG_ROWS)["launch_flags"] & SPLIT
    # a launch at another width than the plan's asks the kernel rule with THAT width
    p = (2449029, 2449029, 123718280, 128, 1446, 0.003, 0.84, 0.2)
    assert _lib.plan_policy(*p)["keep_staged"] == 0 and _lib.plan_policy(*p)["segmented"] == 1
    assert _lib.plan_policy(*p, N_launch=64)["segmented"] == 0


def test_policy_rejects_bad_queries():
    with pytest.raises(_lib.GespmmError):
        _lib.plan_policy(-1, 1, 1, 1, 1)
    with pytest.raises(_lib.GespmmError):
        _lib.plan_policy(10, 10, 10, 8, 1, variant=99)


def test_warm_up_only_where_an_analysis_could_pay():
    """gespmm_plan_wants_warmup: the cost rule with the most structure a probe could report. pubmed- and cora-sized graphs cannot pay for
    an analysis inside 200 launches even when the library is warm — the Python layer then does not spend gespmm_init's ~60 ms on them
    (profiles/r06/gcn_epochs.log: the reference's GCN run on pubmed regressed by 0.5 ms per epoch while it did) — the headline graph and
    everything larger can; a long-lived plan changes the answer."""
    w = lambda M, nnz, N, launches=0: _lib.lib.gespmm_plan_wants_warmup(M, M, nnz, N, launches)
    assert w(19717, 108365, 128) == 0 and w(2708, 10556, 128) == 0 and w(19717, 108365, 128, 10000) == 1
    assert w(334863, 1851744, 128) == 1 and w(334863, 1851744, 32) == 0 and w(232965, 114615892, 32) == 1
    assert _lib.lib.gespmm_plan_wants_warmup(-1, 1, 1, 1, 0) < 0



def test_padded_record_kernel_rules():
    """plan_policy.cpp want_record_tables / keep_record_tables / records_batches_per_task (profiles/r06/records_audit.log): short rows at
    narrow widths in an order that hits L2; any mean degree at N <= 16; not beside staged tables that were kept; kept by slot fill."""
    amazon = dict(M=334863, K=334863, nnz=1851744, max_degree=120, hits_before=0.054, expected_launches=1000000, wedge_probe=0.3)
    q = lambda **kw: _lib.plan_policy(**{**amazon, **kw})
    a = q(N=32, hits_after=0.675)
    assert (a["keep_clustered"], a["build_records"], a["keep_records"], a["records_batches"]) == (1, 1, 1, 4)
    assert q(N=32, hits_after=0.675, record_slot_fill=0.59)["keep_records"] == 1
    assert q(N=32, hits_after=0.675, record_slot_fill=0.46)["keep_records"] == 0  # (LFR with 16-row tasks: behind the streaming kernel)
    assert q(N=16, hits_after=0.686, record_slot_fill=0.42)["keep_records"] == 0 and q(N=16, hits_after=0.686, record_slot_fill=0.53)["keep_records"] == 1
    assert q(N=64, hits_after=0.668)["build_records"] == 1
    assert q(N=47, hits_after=0.67)["build_records"] == 1  # (41 / 47 classes: 4-byte-aligned vectors, records_anywidth.log)
    for kw in (dict(N=128, hits_after=0.66), dict(N=3, hits_after=0.675), dict(N=66, hits_after=0.675), dict(N=32, hits_after=0.222),
               dict(N=32, hits_after=0.675, max_degree=2726), dict(N=32, hits_after=0.675, variant=_lib.VARIANT_CRC)):
        assert q(**kw)["build_records"] == 0, kw
    # rows of 50 entries (products-shaped, a quarter of the size): the lane-group staged kernel keeps N = 32 / 64, records take N = 16
    prod = dict(M=612257, K=612257, nnz=30929570, max_degree=1009, hits_before=0.04, hits_after=0.85, expected_launches=1000000, wedge_probe=0.3)
    p32 = _lib.plan_policy(N=32, staged_fraction=0.70, **prod)
    assert (p32["keep_staged"], p32["build_records"]) == (1, 0)
    p16 = _lib.plan_policy(N=16, **prod)
    assert (p16["build_records"], p16["records_batches"]) == (1, 11)
    # asked for by name: any order, any mean degree (still 4 <= N <= 64, rows <= 1024)
    assert _lib.plan_policy(N=32, kernel=_lib.PLAN_KERNEL_RECORDS, reorder=_lib.PLAN_NO_REORDER, **{**prod, "hits_after": 0.04})["build_records"] == 1
    assert _lib.plan_policy(N=32, kernel=_lib.PLAN_KERNEL_RECORDS, **{**prod, "max_degree": 1500})["build_records"] == 0


def test_column_slab_rule():
    """plan_policy.cpp slab_count_for (profiles/r06/slabs/): dense clustered matrices at N = 128 are cut into round(mean degree / 64) ascending
    column ranges (reddit-shaped communities: 8); rows of 50 entries keep the one-launch staged kernel (two to four ranges lose there); the
    structureless dense graph keeps its storage order (nothing to cut); other widths are not served; asked for by name: any mean degree."""
    reddit = dict(M=232965, K=232965, nnz=114615892, max_degree=7021, hits_before=0.026, expected_launches=1000000, wedge_probe=0.2)
    assert _lib.plan_policy(N=128, hits_after=0.725, **reddit)["slab_ranges"] == 8
    assert _lib.plan_policy(N=128, hits_after=0.033, **reddit)["slab_ranges"] == 0  # (order not kept: the cache-blocked path)
    assert _lib.plan_policy(N=256, hits_after=0.725, **reddit)["slab_ranges"] == 0
    assert _lib.plan_policy(N=128, hits_after=0.725, variant=_lib.VARIANT_CRC, **reddit)["slab_ranges"] == 0
    prod = dict(M=2449029, K=2449029, nnz=123718280, max_degree=1009, hits_before=0.04, hits_after=0.85, expected_launches=1000000, wedge_probe=0.3)
    assert _lib.plan_policy(N=128, **prod)["slab_ranges"] == 0
    assert _lib.plan_policy(N=128, kernel=_lib.PLAN_KERNEL_STAGED_SLABS, **prod)["slab_ranges"] == 2
    assert _lib.plan_policy(N=128, hits_after=0.8, **dict(reddit, nnz=232965 * 96))["slab_ranges"] == 2  # (slab_density.log: 590 -> 464 us)
    assert _lib.plan_policy(N=128, hits_after=0.8, **dict(reddit, nnz=232965 * 64))["slab_ranges"] == 0  # (315 vs 337 us: the one-launch kernel)
    denser = dict(reddit, nnz=232965 * 1100)
    assert _lib.plan_policy(N=128, hits_after=0.725, **denser)["slab_ranges"] == 16  # (the rule stops at 16 ranges)
