"""ctypes binding of libgespmm.so (the C ABI declared in include/gespmm.h).

This is the reference-side binding a maintainer would write (INTEGRATION.md shows
the same stub). It never falls back to a CPU implementation: a missing library is
an ImportError, a failing call is a RuntimeError carrying gespmm_error_string().
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgespmm.so")

VARIANT_AUTO = -1
VARIANT_NAIVE = 0
VARIANT_CRC = 1
VARIANT_CRC_CWM2 = 2
VARIANT_CRC_CWM4 = 3
VARIANT_CRC_CWM8 = 4
VARIANT_PARREDUCE = 5
NUM_VARIANTS = 6

FLAG_NO_XCD_REMAP = 0x1
FLAG_NT_STORE = 0x2
FLAG_SC1_STORE = 0x8000
FLAG_FORCE_IDX64 = 0x4
FLAG_BATCH_STREAM = 0x20
FLAG_SEG_STREAM = 0x80
FLAG_STRICT_ORDER = 0x100
FLAG_SPLIT_LONG_ROWS = 0x200
FLAG_SLAB_BLOCKED = 0x400
FLAG_NO_SLAB_BLOCKED = 0x800
FLAG_ALLOW_REASSOCIATION = 0x1000
FLAG_REUSE_SPLIT = 0x2000
FLAG_SHALLOW_UNROLL = 0x10

# Every symbol include/gespmm.h declares; tests check the library exports all of them.
EXPORTS = [
    "gespmm_version",
    "gespmm_error_string",
    "gespmm_csr_spmm_f32",
    "gespmm_csr_spmm_max_f32",
    "gespmm_select_variant",
    "gespmm_describe_launch",
    "gespmm_dgl_csrmm_sum_f32",
    "gespmm_dgl_csrmm_max_f32",
    "gespmm_dgl_set_readback_rows",
    "gespmm_csr_spmm_f32_cfg",
    "gespmm_csr_spmm_workspace_bytes",
    "gespmm_csr_spmm_f32_ws",
    "gespmm_sddmm_coo_f32",
    "gespmm_sddmm_csr_f32",
    "gespmm_csr2csc_workspace_bytes",
    "gespmm_csr2csc_f32",
    "gespmm_mtx_read",
    "gespmm_mtx_read_cached",
    "gespmm_mtx_free",
    "gespmm_coo_to_csr",
    "gespmm_row_partition",
    "gespmm_baseline_atomic_scatter_f32",
    "gespmm_baseline_copy_f32",
    "gespmm_plan_create",
    "gespmm_plan_spmm_f32",
    "gespmm_plan_spmm_max_f32",
    "gespmm_plan_sddmm_f32",
    "gespmm_plan_set_values",
    "gespmm_plan_get_order",
    "gespmm_plan_describe",
    "gespmm_plan_destroy",
    "gespmm_cluster_rows",
    "gespmm_simulate_l2_hits",
    "gespmm_plan_create_v2",
    "gespmm_plan_policy",
    "gespmm_plan_policy_v2",
    "gespmm_plan_tune",
    "gespmm_set_cached_memory_limit",
    "gespmm_device_cluster_rows",
    "gespmm_device_l2_model",
    "gespmm_plan_debug_tasks",
    "gespmm_release_cached_memory",
    "gespmm_init",
    "gespmm_set_auto_plan",
    "gespmm_auto_plan_clear",
    "gespmm_auto_plan_get_stats",
    "gespmm_cluster_rows_study",
    "gespmm_plan_wants_warmup",
]

PLAN_REORDER_AUTO = 0
PLAN_REORDER = 1
PLAN_NO_REORDER = 2
PLAN_ANALYSIS_DEVICE = 0
PLAN_ANALYSIS_HOST = 1
PLAN_KERNEL_AUTO = 0
PLAN_KERNEL_STREAM = 1
PLAN_KERNEL_SEG_STREAM = 3
PLAN_KERNEL_STAGED = 5
PLAN_KERNEL_RECORDS = 6
PLAN_KERNEL_STAGED_SLABS = 7


class LaunchCfg(Structure):
    _fields_ = [("vec", c_int32), ("strips", c_int32), ("group", c_int32), ("rows_per_wave", c_int32),
                ("slab_rows", c_int32), ("flags", c_int32)]


class PlanOptions(Structure):
    _fields_ = [("reorder", c_int32), ("task_entries", c_int32), ("row_floor", c_int32), ("threads", c_int32),
                ("flags", c_int32), ("kernel", c_int32), ("analysis", c_int32), ("expected_launches", c_int32)]


class PlanPolicyQuery(Structure):
    _fields_ = [("M", c_int64), ("K", c_int64), ("nnz", c_int64), ("N", c_int64), ("N_launch", c_int64), ("variant", c_int32),
                ("max_degree", c_int32), ("reorder", c_int32), ("kernel", c_int32), ("analysis", c_int32), ("flags", c_int32),
                ("task_entries", c_int32), ("row_floor", c_int32), ("hits_before", ctypes.c_double),
                ("hits_after", ctypes.c_double), ("staged_fraction", ctypes.c_double), ("expected_launches", c_int32),
                ("cold_start", c_int32), ("wedge_probe", ctypes.c_double), ("record_slot_fill", ctypes.c_double)]


class PlanPolicyAnswer(Structure):
    _fields_ = [(n, c_int32) for n in ("launch_flags", "analyse", "dense_try", "keep_clustered", "task_entries",
                                       "group_task_entries", "row_floor", "build_staged", "keep_staged", "shallow_unroll",
                                       "segmented", "sddmm_route", "narrow_vec4")] + [("model_window", c_int64), ("model_sample", c_int64),
                                                                                       ("cost_skipped", c_int32), ("cluster_levels", c_int32),
                                                                                       ("est_gain_us", ctypes.c_double),
                                                                                       ("est_cost_us", ctypes.c_double), ("cluster_sweeps", c_int32),
                                                                                       ("staged_rows", c_int32), ("build_records", c_int32),
                                                                                       ("keep_records", c_int32), ("records_batches", c_int32),
                                                                                       ("slab_ranges", c_int32)]


class AutoPlanStats(Structure):
    _fields_ = [(n, c_int64) for n in ("calls_planned", "plans_created", "invalidated", "values_refreshed", "fingerprints", "cached_plans", "calls_async")]


class Coo(Structure):
    _fields_ = [
        ("nrows", c_int32),
        ("ncols", c_int32),
        ("nnz", c_int64),
        ("row", POINTER(c_int32)),
        ("col", POINTER(c_int32)),
        ("val", POINTER(c_float)),
    ]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "gespmm_amd: %s is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C gespmm_amd/csrc`). There is no CPU fallback." % LIB_PATH
        )
    lib = ctypes.CDLL(LIB_PATH)
    p = c_void_p
    lib.gespmm_version.restype = c_char_p
    lib.gespmm_error_string.restype = c_char_p
    lib.gespmm_error_string.argtypes = [c_int]
    lib.gespmm_csr_spmm_f32.restype = c_int
    lib.gespmm_csr_spmm_f32.argtypes = [p, p, p, p, p, c_int64, c_int64, c_int64, c_int64, c_int, p]
    lib.gespmm_csr_spmm_f32_cfg.restype = c_int
    lib.gespmm_csr_spmm_f32_cfg.argtypes = [p, p, p, p, p, c_int64, c_int64, c_int64, c_int64, c_int,
                                            POINTER(LaunchCfg), p]
    lib.gespmm_csr_spmm_workspace_bytes.restype = c_int64
    lib.gespmm_csr_spmm_workspace_bytes.argtypes = [c_int64, c_int64, c_int64, c_int64, c_int, POINTER(LaunchCfg)]
    lib.gespmm_csr_spmm_f32_ws.restype = c_int
    lib.gespmm_csr_spmm_f32_ws.argtypes = [p, p, p, p, p, c_int64, c_int64, c_int64, c_int64, c_int,
                                           POINTER(LaunchCfg), p, c_int64, p]
    lib.gespmm_csr_spmm_max_f32.restype = c_int
    lib.gespmm_csr_spmm_max_f32.argtypes = [p, p, p, p, c_int64, c_int64, c_int64, c_int64, c_float, c_int, p]
    for fn in (lib.gespmm_dgl_csrmm_sum_f32, lib.gespmm_dgl_csrmm_max_f32):
        fn.restype = c_int
        fn.argtypes = [c_int, c_int, p, p, p, p, p]
    lib.gespmm_dgl_set_readback_rows.restype = c_int
    lib.gespmm_dgl_set_readback_rows.argtypes = [c_int64]
    lib.gespmm_describe_launch.restype = c_int
    lib.gespmm_describe_launch.argtypes = [c_int64, c_int64, c_int64, c_int64, c_int, POINTER(LaunchCfg), c_char_p, c_int64]
    lib.gespmm_select_variant.restype = c_int
    lib.gespmm_select_variant.argtypes = [c_int64, c_int64, c_int64]
    lib.gespmm_sddmm_coo_f32.restype = c_int
    lib.gespmm_sddmm_coo_f32.argtypes = [p, p, p, p, p, c_int64, c_int64, p]
    lib.gespmm_sddmm_csr_f32.restype = c_int
    lib.gespmm_sddmm_csr_f32.argtypes = [p, p, p, p, p, c_int64, c_int64, c_int64, p]
    lib.gespmm_csr2csc_workspace_bytes.restype = c_int64
    lib.gespmm_csr2csc_workspace_bytes.argtypes = [c_int64, c_int64, c_int64]
    lib.gespmm_csr2csc_f32.restype = c_int
    lib.gespmm_csr2csc_f32.argtypes = [p, p, p, p, p, p, c_int64, c_int64, c_int64, p, p]
    lib.gespmm_mtx_read.restype = c_int
    lib.gespmm_mtx_read.argtypes = [c_char_p, POINTER(Coo)]
    lib.gespmm_mtx_read_cached.restype = c_int
    lib.gespmm_mtx_read_cached.argtypes = [c_char_p, c_char_p, POINTER(Coo)]
    lib.gespmm_mtx_free.restype = None
    lib.gespmm_mtx_free.argtypes = [POINTER(Coo)]
    lib.gespmm_coo_to_csr.restype = c_int
    lib.gespmm_coo_to_csr.argtypes = [c_int32, c_int32, c_int64, p, p, p, p, p, p]
    lib.gespmm_baseline_atomic_scatter_f32.restype = c_int
    lib.gespmm_baseline_atomic_scatter_f32.argtypes = [p, p, p, p, c_int64, c_int64, c_int64, c_int64, p]
    lib.gespmm_plan_policy_v2.restype = c_int
    lib.gespmm_plan_policy_v2.argtypes = [POINTER(PlanPolicyQuery), c_int64, POINTER(PlanPolicyAnswer), c_int64]
    lib.gespmm_baseline_copy_f32.restype = c_int
    lib.gespmm_baseline_copy_f32.argtypes = [p, p, c_int64, p]
    lib.gespmm_plan_create.restype = c_int
    lib.gespmm_plan_create.argtypes = [POINTER(c_void_p), p, p, p, c_int64, c_int64, c_int64, c_int64, c_int,
                                       POINTER(PlanOptions), p]
    lib.gespmm_plan_create_v2.restype = c_int
    lib.gespmm_plan_create_v2.argtypes = [POINTER(c_void_p), p, p, p, c_int64, c_int64, c_int64, c_int64, c_int,
                                          POINTER(PlanOptions), c_int64, p]
    lib.gespmm_plan_policy.restype = c_int
    lib.gespmm_plan_policy.argtypes = [POINTER(PlanPolicyQuery), POINTER(PlanPolicyAnswer)]
    lib.gespmm_plan_tune.restype = c_int
    lib.gespmm_plan_tune.argtypes = [p, p, p, c_int64, c_int32, p]
    lib.gespmm_set_cached_memory_limit.restype = None
    lib.gespmm_set_cached_memory_limit.argtypes = [c_int64]
    lib.gespmm_plan_spmm_f32.restype = c_int
    lib.gespmm_plan_spmm_f32.argtypes = [p, p, p, c_int64, p]
    lib.gespmm_plan_spmm_max_f32.restype = c_int
    lib.gespmm_plan_spmm_max_f32.argtypes = [p, p, p, c_int64, c_float, p]
    lib.gespmm_plan_sddmm_f32.restype = c_int
    lib.gespmm_plan_sddmm_f32.argtypes = [p, p, p, p, c_int64, p]
    lib.gespmm_plan_set_values.restype = c_int
    lib.gespmm_plan_set_values.argtypes = [p, p, p]
    lib.gespmm_plan_get_order.restype = c_int
    lib.gespmm_plan_get_order.argtypes = [p, p]
    lib.gespmm_plan_describe.restype = c_int
    lib.gespmm_plan_describe.argtypes = [p, c_char_p, c_int64]
    lib.gespmm_plan_destroy.restype = None
    lib.gespmm_plan_destroy.argtypes = [p]
    lib.gespmm_cluster_rows.restype = c_int
    lib.gespmm_cluster_rows.argtypes = [p, p, c_int64, c_int64, c_int32, p, p, p]
    lib.gespmm_device_cluster_rows.restype = c_int
    lib.gespmm_device_cluster_rows.argtypes = [p, p, c_int64, c_int64, c_int64, p, p, p, p]
    lib.gespmm_device_l2_model.restype = ctypes.c_double
    lib.gespmm_device_l2_model.argtypes = [p, p, c_int64, c_int64, c_int64, p, c_int32, c_int64, c_int64, c_int32, p]
    lib.gespmm_release_cached_memory.restype = None
    lib.gespmm_release_cached_memory.argtypes = []
    lib.gespmm_plan_debug_tasks.restype = c_int
    lib.gespmm_plan_debug_tasks.argtypes = [p, c_int32, p, c_int64]
    lib.gespmm_simulate_l2_hits.restype = ctypes.c_double
    lib.gespmm_simulate_l2_hits.argtypes = [p, p, c_int64, c_int64, p, c_int32, c_int64]
    lib.gespmm_row_partition.restype = c_int
    lib.gespmm_row_partition.argtypes = [p, c_int64, c_int32, p]
    lib.gespmm_init.restype = c_int
    lib.gespmm_init.argtypes = [c_int64, c_int64, p]
    lib.gespmm_plan_wants_warmup.restype = c_int
    lib.gespmm_plan_wants_warmup.argtypes = [c_int64, c_int64, c_int64, c_int64, c_int32]
    lib.gespmm_set_auto_plan.restype = c_int
    lib.gespmm_set_auto_plan.argtypes = [c_int32]
    lib.gespmm_auto_plan_clear.restype = None
    lib.gespmm_auto_plan_clear.argtypes = []
    lib.gespmm_auto_plan_get_stats.restype = c_int
    lib.gespmm_auto_plan_get_stats.argtypes = [POINTER(AutoPlanStats)]
    return lib


lib = _load()


class GespmmError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__("%s failed: %s (code %d)" % (where, lib.gespmm_error_string(code).decode(), code))


def plan_policy(M, K, nnz, N, max_degree, hits_before=0.0, hits_after=0.0, staged_fraction=0.0, N_launch=0, variant=VARIANT_AUTO,
                reorder=PLAN_REORDER_AUTO, kernel=PLAN_KERNEL_AUTO, analysis=PLAN_ANALYSIS_DEVICE, flags=0, task_entries=0,
                row_floor=0, expected_launches=0, wedge_probe=-1.0, cold_start=0, record_slot_fill=-1.0):
    """What a plan would decide for a matrix of this shape (gespmm_plan_policy_v2: host only, no device) — a dict.
    `wedge_probe` is the plan's structure probe (share of sampled wedges that close; negative = unknown), `expected_launches` the
    number of products the analysis has to pay for itself in (0 = 200)."""
    q = PlanPolicyQuery(int(M), int(K), int(nnz), int(N), int(N_launch), int(variant), int(max_degree), int(reorder), int(kernel),
                        int(analysis), int(flags), int(task_entries), int(row_floor), float(hits_before), float(hits_after),
                        float(staged_fraction), int(expected_launches), int(cold_start), float(wedge_probe), float(record_slot_fill))
    a = PlanPolicyAnswer()
    check(lib.gespmm_plan_policy_v2(ctypes.byref(q), ctypes.sizeof(q), ctypes.byref(a), ctypes.sizeof(a)), "gespmm_plan_policy_v2")
    return {n: getattr(a, n) for n, _ in PlanPolicyAnswer._fields_ if not n.startswith("reserved")}


def init(rows_hint=0, nnz_hint=0, stream=None):
    """gespmm_init: load the analysis kernels and make the analysis arena NOW (the first plan of a process otherwise pays ~29 ms for
    both), optionally sized for a matrix of (rows_hint, nnz_hint). Needs a HIP device; idempotent per device."""
    check(lib.gespmm_init(int(rows_hint), int(nnz_hint), stream), "gespmm_init")


_initialised = set()


def ensure_init(device_index, stream=None, shape=None, expected_launches=0, forced=False):
    """gespmm_init once per process and device, before the first plan that could use it: the Python layer never builds such a plan cold
    (the ~29 ms of kernel loading and arena allocation belong to start-up, not to the first plan's analysis time — and not to its cost
    rule). `shape` = (M, K, nnz, N) of the plan about to be made: a matrix whose analysis could not pay even when warm (pubmed-sized
    graphs) keeps its storage order either way, so nothing is warmed up for it (the reference's GCN run on pubmed would otherwise pay
    ~0.5 ms per epoch for it: profiles/r06/gcn_epochs.log)."""
    if device_index in _initialised:
        return
    if shape is not None and not forced and lib.gespmm_plan_wants_warmup(int(shape[0]), int(shape[1]), int(shape[2]), int(shape[3]), int(expected_launches)) != 1:
        return
    init(0, 0, stream)
    _initialised.add(device_index)


def set_auto_plan(kth_call):
    """gespmm_set_auto_plan: from the k-th identical stateless call on (gespmm_csr_spmm_f32 / _max / gespmm_dgl_csrmm_*), run through a
    plan the library keeps (0 = off, the default). Every planned call fingerprints the CSR arrays first (one stream synchronisation)."""
    check(lib.gespmm_set_auto_plan(int(kth_call)), "gespmm_set_auto_plan")


def auto_plan_stats():
    st = AutoPlanStats()
    check(lib.gespmm_auto_plan_get_stats(ctypes.byref(st)), "gespmm_auto_plan_get_stats")
    return {n: getattr(st, n) for n, _ in AutoPlanStats._fields_}


def auto_plan_clear():
    lib.gespmm_auto_plan_clear()


def release_cached_memory():
    """Give the analysis stage's cached scratch arenas back to the devices (gespmm_release_cached_memory)."""
    lib.gespmm_release_cached_memory()


def set_cached_memory_limit(nbytes):
    """Largest analysis arena the library keeps between plans, per device (default 1 GiB; larger arenas are freed when the plan
    is made). A caller that builds many large plans in a row raises it (products-sized graphs: ~10 GB) and calls
    release_cached_memory() when done."""
    lib.gespmm_set_cached_memory_limit(int(nbytes))


def check(code, where):
    if code != 0:
        raise GespmmError(code, where)
