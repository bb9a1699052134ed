import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Build artefacts normally exist (driver runs __graft_entry__.build() first; the
    # .so files travel to the GPU box). Build only what is missing.
    lib = os.path.join(ROOT, "gespmm_amd", "lib", "libgespmm.so")
    drv = os.path.join(ROOT, "gespmm_amd", "lib", "spmm_test")
    if not (os.path.exists(lib) and os.path.exists(drv)):
        subprocess.run(["make", "-C", os.path.join(ROOT, "gespmm_amd", "csrc"), "-j8", "all"], check=True)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)


@pytest.fixture(scope="session")
def oracle():
    import oracle_py

    return oracle_py


@pytest.fixture(scope="session")
def pkg():
    import gespmm_amd

    return gespmm_amd


@pytest.fixture(scope="session")
def known_answers():
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def bundled(oracle):
    """CSR (oracle loader + oracle COO->CSR) of the three bundled matrices."""
    out = {}
    for g in ("cora", "citeseer", "pubmed"):
        coo = oracle.read_mtx(os.path.join(GOLDEN, g + ".mtx"))
        indptr, indices, _ = oracle.coo_to_csr(coo["nrows"], coo["row"], coo["col"])
        out[g] = {"M": coo["nrows"], "K": coo["ncols"], "nnz": coo["nnz"], "rowptr": indptr, "colind": indices}
    return out
