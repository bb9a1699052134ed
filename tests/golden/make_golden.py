#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/. Run in the BUILD container
(it reads /root/reference for the spreadsheet facts; the GPU box never runs this).

What is produced and where it comes from:

  known_answers.json
      "xlsx": facts read from the reference's own recorded data,
              /root/reference/matrix_id_info.xlsx sheet1: the (time ms, throughput
              GFLOP/s) cells at N=512 for cora / citeseer / pubmed. The driver computes
              throughput = 2*nnz*N/1e6 / time_ms (spmm_test.cu:728,738), so
              time*tpt*1e6/(2*512) is the nnz the REFERENCE LOADER produced for that
              file (after symmetric expansion and self-loop/duplicate removal) — an
              answer that pins the oracle loader without building the reference.
              Also the N / nnz columns for cit-HepTh and com-Amazon (sizes of the
              synthetic stand-ins).
      "survey": M / nnz / first / last / max-degree of the bundled matrices as obtained
              from the unmodified reference loader during the survey.
  mtx/*.mtx + mtx_expected.json
      hand-made MatrixMarket edge cases with HAND-WRITTEN expected COO (derived from the
      readMtx rules in util/util.hpp:104-333, not from running any code here).
  spmm_checksums.json
      checksums + 64 sampled outputs on the bundled matrices:
        "unweighted_golden", "valued_golden"  OUTPUTS OF THE REFERENCE ITSELF RUN HERE —
            oracle/_ref/libref_host.so = readMtx (util.hpp:57-333), COO->CSR
            (spmm_test.cu:558-581) and the CPU golden loop (spmm_test.cu:596-604) compiled
            from /root/reference by oracle/make_ref.sh; this script only feeds them B / val.
        "valued_fma"  the oracle's restatement of the reference's DEVICE arithmetic (one
            fused multiply-add per non-zero). The reference's kernels themselves, compiled
            for gfx950 (oracle/_ref/libref_kernels.so), are compared with it on the GPU
            (tests/test_gpu_ref_kernels.py) — a CPU cannot produce this column from them.
cora.mtx / citeseer.mtx / pubmed.mtx are the reference's bundled data files
(data/misc/, MIT licence), copied byte for byte as input fixtures.
"""
import json
import os
import re
import sys
import zipfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

N_LIST = [3, 16, 32, 41, 64, 128, 512]


def xlsx_facts(path):
    z = zipfile.ZipFile(path)
    ss = z.read("xl/sharedStrings.xml").decode()
    strs = [re.sub(r"<[^>]+>", "", s) for s in re.findall(r"<si>(.*?)</si>", ss, flags=re.S)]
    sh = z.read("xl/worksheets/sheet1.xml").decode()
    out = {}
    for r in re.findall(r"<row [^>]*>(.*?)</row>", sh, flags=re.S):
        cells = {}
        for m in re.finditer(r'<c r="([A-Z]+)\d+"([^>]*?)(?:/>|>(.*?)</c>)', r, flags=re.S):
            col, attr, body = m.groups()
            if body is None:
                continue
            v = re.search(r"<v>(.*?)</v>", body)
            if not v:
                continue
            v = v.group(1)
            if 't="s"' in attr:
                v = strs[int(v)]
            cells[col] = v
        name = cells.get("A")
        if name in ("cora", "citeseer", "pubmed"):
            t, tpt = float(cells["E"]), float(cells["F"])  # N=512: time (ms), csrmm2 throughput
            out[name] = {"time_ms_n512": t, "gflops_n512": tpt, "implied_nnz": t * tpt * 1e6 / (2 * 512)}
        if name in ("cit-HepTh", "com-Amazon"):
            out[name] = {"M": int(cells["B"]), "nnz": int(cells["C"])}
    return out


SURVEY_FACTS = {  # reference loader run unchanged on the bundled files
    "cora": {"M": 2708, "file_entries": 7986, "nnz": 10556, "first": [0, 8], "last": [2707, 2344], "max_degree": 168},
    "citeseer": {"M": 3327, "file_entries": 4676, "nnz": 9104, "first": [0, 628], "last": [3326, 33],
                 "max_degree": 99},
    "pubmed": {"M": 19717, "file_entries": 44327, "nnz": 88648, "first": [0, 1378], "last": [19716, 16030],
               "max_degree": 171},
}

# Hand-made loader cases: file text -> expected result by the readMtx rules.
MTX_CASES = {
    "pattern_general_dups.mtx": {
        "text": "%%MatrixMarket matrix coordinate pattern general\n% a comment\n%another\n4 5 7\n"
                "3 2\n1 5\n1 1\n3 2\n4 4\n1 2\n2 5\n",
        # general: duplicates (3,2)x2 and the self-loop-like (1,1),(4,4) are KEPT; sorted by (row,col)
        "expect": {"rc": 0, "nrows": 4, "ncols": 5, "row": [0, 0, 0, 1, 2, 2, 3], "col": [0, 1, 4, 4, 1, 1, 3],
                   "val": [1, 1, 1, 1, 1, 1, 1]},
    },
    "integer_symmetric.mtx": {
        "text": "%%MatrixMarket matrix coordinate integer symmetric\n5 5 6\n"
                "1 1 7\n2 1 3\n1 2 3\n4 2 -2\n5 5 9\n5 3 4\n",
        # symmetric: mirror off-diagonals, drop self-loops (1,1),(5,5) and the duplicate pairs that
        # (2,1)+(1,2) create; values are integers cast to float
        "expect": {"rc": 0, "nrows": 5, "ncols": 5, "row": [0, 1, 1, 2, 3, 4], "col": [1, 0, 3, 4, 1, 2],
                   "val": [3, 3, -2, 4, -2, 4]},
    },
    "real_general_rect.mtx": {
        "text": "%%MatrixMarket matrix coordinate real general\n%comment\n\n3 6 4\n"
                "3 6 -1.5e+00\n1 1 2.5\n2 4 1e-3\n1 6 0.125\n",
        # blank line before the size line (mmio.hpp:326-333 falls through to fscanf); rectangular
        "expect": {"rc": 0, "nrows": 3, "ncols": 6, "row": [0, 0, 1, 2], "col": [0, 5, 3, 5],
                   "val": [2.5, 0.125, 0.001, -1.5]},
    },
    "upper_banner.mtx": {
        "text": "%%MatrixMarket MATRIX Coordinate REAL General\n2 2 2\n2 1 1.0\n1 2 4.0\n",
        # tokens after the first are case-insensitive (mmio.hpp:236-239)
        "expect": {"rc": 0, "nrows": 2, "ncols": 2, "row": [0, 1], "col": [1, 0], "val": [4.0, 1.0]},
    },
    "short_entries.mtx": {
        "text": "%%MatrixMarket matrix coordinate pattern general\n3 3 5\n1 1\n2 3\n3 1\n",
        # fewer entries than promised: "Error: not enough rows in mtx file." and keep what was read
        "expect": {"rc": 0, "nrows": 3, "ncols": 3, "row": [0, 1, 2], "col": [0, 2, 0], "val": [1, 1, 1]},
    },
    "skew_not_expanded.mtx": {
        "text": "%%MatrixMarket matrix coordinate integer skew-symmetric\n3 3 2\n2 1 5\n3 2 -1\n",
        # only `symmetric` is expanded (util.hpp:322); skew-symmetric is read as stored
        "expect": {"rc": 0, "nrows": 3, "ncols": 3, "row": [1, 2], "col": [0, 1], "val": [5, -1]},
    },
    "symmetric_pattern_empty_rows.mtx": {
        "text": "%%MatrixMarket matrix coordinate pattern symmetric\n6 6 3\n4 1\n6 4\n2 2\n",
        # rows 1,2,4 (0-based) end up empty; diagonal (2,2) dropped
        "expect": {"rc": 0, "nrows": 6, "ncols": 6, "row": [0, 3, 3, 5], "col": [3, 0, 5, 3], "val": [1, 1, 1, 1]},
    },
    "bad_banner.mtx": {
        "text": "%%NotMatrixMarket matrix coordinate real general\n2 2 1\n1 1 1.0\n",
        "expect": {"rc": "format"},
    },
    "short_banner.mtx": {
        "text": "%%MatrixMarket matrix coordinate real\n2 2 1\n1 1 1.0\n",
        "expect": {"rc": "format"},
    },
    "complex_rejected.mtx": {
        "text": "%%MatrixMarket matrix coordinate complex general\n2 2 1\n1 1 1.0 0.0\n",
        # reference: reads nothing yet reports nnz=1 (util.hpp:315-320) -> UB downstream; we reject
        "expect": {"rc": "format"},
    },
}


def sample_positions(M, N, count=64, seed=12345):
    rng = np.random.RandomState(seed)
    return [(int(rng.randint(0, M)), int(rng.randint(0, N))) for _ in range(count)]


def main():
    import oracle_py as o
    import ref_py as r

    if not r.available():
        r.build()

    ka = {"xlsx": xlsx_facts("/root/reference/matrix_id_info.xlsx"), "survey": SURVEY_FACTS,
          "provenance": "see docstring of tests/golden/make_golden.py"}
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1, sort_keys=True)

    os.makedirs(os.path.join(HERE, "mtx"), exist_ok=True)
    expected = {}
    for name, case in MTX_CASES.items():
        with open(os.path.join(HERE, "mtx", name), "w") as f:
            f.write(case["text"])
        expected[name] = case["expect"]
    with open(os.path.join(HERE, "mtx_expected.json"), "w") as f:
        json.dump(expected, f, indent=1, sort_keys=True)

    sums = {}
    for g in ("cora", "citeseer", "pubmed"):
        coo = r.read_mtx(os.path.join(HERE, g + ".mtx"))  # the reference's loader
        coo["nnz"] = coo["nvals"]
        indptr, indices, ones = r.coo_to_csr(coo["nrows"], coo["ncols"], coo["row"], coo["col"])
        val = o.hash_val(coo["nnz"], seed=7)
        sums[g] = {}
        for N in N_LIST:
            B = o.hash_B(coo["ncols"], N, seed=1)
            pos = sample_positions(coo["nrows"], N)
            entry = {}
            for mode, v in (("unweighted_golden", None), ("valued_golden", val), ("valued_fma", val)):
                if mode == "valued_fma":
                    C = o.spmm(indptr, indices, v, B, mode="fma")
                else:  # the reference's own loop
                    C = r.golden(indptr, indices, ones if v is None else v, B)
                entry[mode] = {
                    "sum": float(C.astype(np.float64).sum()),
                    "sumsq": float((C.astype(np.float64) ** 2).sum()),
                    "xor": int(np.bitwise_xor.reduce(C.view(np.uint32).ravel())),
                    "samples": [[r, c, int(C[r, c:c + 1].view(np.uint32)[0])] for r, c in pos],
                }
            sums[g][str(N)] = entry
    with open(os.path.join(HERE, "spmm_checksums.json"), "w") as f:
        json.dump({"B": "oracle_py.hash_B(K, N, seed=1)", "val": "oracle_py.hash_val(nnz, seed=7)",
                   "provenance": "unweighted_golden / valued_golden: reference lines via oracle/_ref "
                                 "(see make_golden.py); valued_fma: oracle restatement",
                   "graphs": sums}, f, indent=0, sort_keys=True)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
