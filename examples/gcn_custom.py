#!/usr/bin/env python3
"""GCN training benchmark on the custom op — the caller of the hot path that
pytorch-custom/gcn_custom.py (2 convs, edge weights) and gcn_custom_2layer.py
(3 convs, no edge weights) are in the reference.

    python examples/gcn_custom.py --n-hidden 128                  # gcn_custom.py
    python examples/gcn_custom.py --n-hidden 128 --convs 3        # gcn_custom_2layer.py
    python examples/gcn_custom.py --dataset reddit-like --n-hidden 128 --epochs 20
    python examples/gcn_custom.py --graph-capture                 # replay each step from a HIP graph

Differences forced by the environment (no network, no torch_geometric):
  * PubMed's adjacency is the reference's bundled data/misc/pubmed.mtx (tests/golden/);
    the Planetoid node features / labels / masks are not available, so features are
    row-normalised U[0,1) 19717 x 500 (seed 0), labels uniform over 3 classes and the
    masks 60 / 500 / 1000 random nodes. Accuracy is therefore chance
    level by construction; the point of the script is the per-epoch op time.
  * `proc()` builds CSR/CSC (+ self loops) exactly like gcn_custom.py:29-49 does with
    scipy, including the naming trap: the CSC arrays go into rowptr/colind, so the
    forward SpMM aggregates over in-neighbours.
  * The profiler table of the reference (torch.autograd.profiler, gcn_custom.py:134,143)
    is replaced by HIP-event timing of the epoch loop plus an optional torch.profiler run.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import scipy.sparse as scpsp  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import gespmm_amd  # noqa: E402,F401
from gespmm_amd import GCNConv, graphs  # noqa: E402


def load_edges(dataset, device):
    if dataset == "pubmed":
        coo = graphs.read_mtx(os.path.join(ROOT, "tests", "golden", "pubmed.mtx"))
        return np.stack([coo["row"], coo["col"]]).astype(np.int32), coo["nrows"], 500, 3
    if dataset in ("cora", "citeseer"):
        coo = graphs.read_mtx(os.path.join(ROOT, "tests", "golden", dataset + ".mtx"))
        return np.stack([coo["row"], coo["col"]]).astype(np.int32), coo["nrows"], 500, 7
    if dataset in ("reddit-like", "reddit-sbm", "com-amazon-sbm", "com-amazon-like", "products-sbm"):
        # DGL reddit shape: 602 input features, 41 classes; the co-purchase stand-ins: 100 features, 47 classes (ogbn-products)
        g = graphs.synthetic_graph(dataset, seed=42, device=device)
        rp = g["rowptr"].long()
        rows = torch.repeat_interleave(torch.arange(g["M"], device=device), rp[1:] - rp[:-1])
        ei = torch.stack([rows, g["colind"].long()]).cpu().numpy().astype(np.int32)
        return (ei, g["M"], 602, 41) if dataset.startswith("reddit") else (ei, g["M"], 100, 47)
    raise SystemExit("unknown dataset " + dataset)


def proc(edge_index, n_v, device, add_self_loop=True):
    """Both index orders of the (self-looped) graph, under the key names the reference's caller uses
    (gcn_custom.py:29-49: `colptr`/`rowind` hold the source-major order, `rowptr`/`colind` the destination-major one,
    so the forward product aggregates over in-neighbours). Built on the device: one stable sort per order; duplicate
    edges add up like scipy's COO -> CSR conversion does."""
    ei = torch.as_tensor(edge_index, dtype=torch.int64, device=device)
    if add_self_loop:
        ids = torch.arange(n_v, dtype=torch.int64, device=device)
        ei = torch.cat([ei, torch.stack([ids, ids])], dim=1)

    def compressed(major, minor):
        key, inverse = torch.unique(major * n_v + minor, sorted=True, return_inverse=True)
        weight = torch.zeros(key.numel(), dtype=torch.float32, device=device).index_add_(
            0, inverse, torch.ones(inverse.numel(), dtype=torch.float32, device=device))
        ptr = torch.zeros(n_v + 1, dtype=torch.int64, device=device)
        ptr[1:] = torch.cumsum(torch.bincount(key // n_v, minlength=n_v), 0)
        return ptr.to(torch.int32), (key % n_v).to(torch.int32), weight

    g = {}
    g["colptr"], g["rowind"], g["value_csc"] = compressed(ei[0], ei[1])
    g["rowptr"], g["colind"], g["value_csr"] = compressed(ei[1], ei[0])
    return g


class Net(torch.nn.Module):
    def __init__(self, n_in, n_hidden, n_out, convs, weighted, cached=True):
        super().__init__()
        dims = [n_in] + [n_hidden] * (convs - 1) + [n_out]
        self.convs = torch.nn.ModuleList(
            [GCNConv(dims[i], dims[i + 1], cached=cached, normalize=True) for i in range(convs)])
        self.weighted = weighted
        self.reg_params = self.convs[0].parameters()
        self.non_reg_params = [p for c in self.convs[1:] for p in c.parameters()]

    def forward(self, x, g):
        a = [g["rowptr"], g["colind"], g["colptr"], g["rowind"]]
        if self.weighted:
            a += [g["value_csr"], g["value_csc"]]
        for i, conv in enumerate(self.convs):
            x = conv(x, *a)
            if i + 1 < len(self.convs):
                x = F.dropout(F.relu(x), training=self.training)
        return F.log_softmax(x, dim=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-hidden", type=int, default=64, help="number of hidden features")
    ap.add_argument("--dataset", default="pubmed")
    ap.add_argument("--convs", type=int, default=2, help="2 = gcn_custom.py, 3 = gcn_custom_2layer.py")
    ap.add_argument("--no-edge-weight", action="store_true", help="unweighted kernels (gcn_custom_2layer.py)")
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--graph-capture", action="store_true", help="capture one training step in a HIP graph")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--no-plans", action="store_true", help="GCNConv(cached=False): every SpMM is a plain call (no analysis stage)")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs a HIP device (the op has no CPU path)")
    device = torch.device("cuda")

    edge_index, n_v, n_feat, n_cls = load_edges(args.dataset, device)
    g = proc(edge_index, n_v, device)
    gen = torch.Generator().manual_seed(0)
    x = torch.rand(n_v, n_feat, generator=gen)
    x = (x / x.sum(1, keepdim=True)).to(device)  # T.NormalizeFeatures()
    y = torch.randint(0, n_cls, (n_v,), generator=gen).to(device)
    perm = torch.randperm(n_v, generator=gen)
    masks = {}
    for name, (a, b) in (("train", (0, 20 * n_cls)), ("val", (20 * n_cls, 20 * n_cls + 500)),
                         ("test", (20 * n_cls + 500, 20 * n_cls + 1500))):
        m = torch.zeros(n_v, dtype=torch.bool)
        m[perm[a:b]] = True
        masks[name] = m.to(device)

    weighted = not args.no_edge_weight and args.convs == 2
    model = Net(n_feat, args.n_hidden, n_cls, args.convs, weighted, cached=not args.no_plans).to(device)
    optimizer = torch.optim.Adam([dict(params=model.reg_params, weight_decay=5e-4),
                                  dict(params=model.non_reg_params, weight_decay=0)], lr=0.01,
                                 capturable=args.graph_capture)

    train_idx = masks["train"].nonzero().squeeze(1)  # index tensor: boolean masks sync (not capturable)
    y_train = y[train_idx]

    def train_step():
        optimizer.zero_grad(set_to_none=False)
        out = model(x, g)
        loss = F.nll_loss(out.index_select(0, train_idx), y_train)
        loss.backward()
        optimizer.step()
        return loss

    @torch.no_grad()
    def test():
        model.eval()
        logits, accs = model(x, g), []
        for m in masks.values():
            pred = logits[m].max(1)[1]
            accs.append(pred.eq(y[m]).sum().item() / m.sum().item())
        model.train()
        return accs

    model.train()
    graph = None
    if args.graph_capture:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                train_step()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_loss = train_step()
    else:
        for _ in range(3):
            train_step()

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    best_val = test_acc = 0.0
    for epoch in range(1, args.epochs + 1):
        if graph is not None:
            graph.replay()
            loss = static_loss
        else:
            loss = train_step()
        if epoch % 50 == 0 or epoch == args.epochs:
            tr, va, te = test()
            if va > best_val:
                best_val, test_acc = va, te
            print("Epoch: {:03d}, Loss: {:.4f}, Train: {:.4f}, Val: {:.4f}, Test: {:.4f}".format(
                epoch, float(loss), tr, best_val, test_acc))
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print("dataset=%s n=%d nnz(+I)=%d hidden=%d convs=%d weighted=%s graph_capture=%s" %
          (args.dataset, n_v, g["colind"].numel(), args.n_hidden, args.convs, weighted, args.graph_capture))
    print("epochs=%d  gpu %.3f ms/epoch  wall %.3f ms/epoch" %
          (args.epochs, e0.elapsed_time(e1) / args.epochs, wall * 1e3 / args.epochs))

    if args.profile:
        from torch.profiler import ProfilerActivity, profile

        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for _ in range(10):
                train_step()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=15))


if __name__ == "__main__":
    main()
