"""Device-side ingest on the GPU: the device planner against the host planner
(array for array), the native COO->CSR reduction against the oracle, the reference's adjacency
recipe (pygcn/utils.py:360-368) built on the device against the host recipe, and the binary cache
file round trip."""
import numpy as np
import pytest
import torch

import inputs as gin
from conftest import assert_normwise, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from pygcn_amd import _native
    _native.lib()
    return torch.device("cuda:0")


def _graph_from_degrees(deg, dev, idx64=False, **kw):
    from pygcn_amd import CSRGraph
    deg = np.asarray(deg, np.int64)
    rowptr = np.concatenate([[0], np.cumsum(deg)])
    nnz = int(rowptr[-1])
    n = len(deg)
    rng = np.random.default_rng(nnz + n)
    col = rng.integers(0, max(n, 1), size=nnz).astype(np.int32)
    val = rng.random(nnz).astype(np.float32)
    return CSRGraph(torch.from_numpy(rowptr.astype(np.int64 if idx64 else np.int32)).to(dev),
                    torch.from_numpy(col).to(dev), torch.from_numpy(val).to(dev), (n, max(n, 1)), **kw)


def _assert_same_schedule(g):
    host = g.plan_arrays_host_planner()
    g.plan()
    k = g._keep
    for name in ("n_items", "n_chunks", "n_long"):
        assert k[name] == host[name], (name, k[name], host[name])
    assert np.array_equal(k["items"][:2 * k["n_items"]].cpu().numpy(), host["items"])
    assert np.array_equal(k["chunk_row"][:k["n_chunks"]].cpu().numpy(), host["chunk_row"])
    assert np.array_equal(k["chunk_e0"][:k["n_chunks"]].cpu().numpy(), host["chunk_e0"])
    assert np.array_equal(k["long_row"][:k["n_long"]].cpu().numpy(), host["long_row"])
    assert np.array_equal(k["long_chunk0"].cpu().numpy(), host["long_chunk0"])


def test_device_planner_equals_host_planner_on_edge_shapes(dev):
    rng = np.random.default_rng(5)
    cases = {
        "one row": [3], "one long row": [1000], "all empty": [0] * 500,
        "all self-loop": [1] * 1000, "exactly 64-row items": [0] * 640,
        "cost boundary": [63, 0, 62, 1, 1, 64, 63, 63, 1],
        "all long": [300] * 70, "long/short alternating": [300, 1] * 100,
        "runs of long rows": [1] * 10 + [400] * 50 + [2] * 10 + [257] * 3 + [256] * 3,
        "thresholds": [255, 256, 257, 512, 513, 0, 256],
        "poisson": rng.poisson(9, 20000), "heavy tail": (rng.pareto(1.1, 30000) * 3).astype(np.int64) % 5000,
    }
    for name, deg in cases.items():
        _assert_same_schedule(_graph_from_degrees(deg, dev))
    # knobs and 64-bit row pointers
    deg = (rng.pareto(1.2, 8000) * 4).astype(np.int64) % 3000
    for kw in ({"item_cost": 16}, {"item_cost": 300}, {"long_thresh": 32}, {"long_thresh": 1000, "item_cost": 128}):
        _assert_same_schedule(_graph_from_degrees(deg, dev, **kw))
    _assert_same_schedule(_graph_from_degrees(deg, dev, idx64=True))


def test_device_planner_on_rmat_and_no_bulk_host_transfer(dev, monkeypatch):
    """C3-sized R-MAT: same schedule as the host planner; and from the arrays to the first launch
    nothing larger than a few scalars crosses to the host (the row pointer stays in HBM)."""
    from pygcn_amd import CSRGraph, spmm_csr
    from pygcn_amd.utils import rmat_graph
    n = 1_000_000
    rowptr, col, val = rmat_graph(n, 10_000_000, seed=42, perm_seed=43, device=dev)
    moved = []
    real_cpu, real_tolist, real_item = torch.Tensor.cpu, torch.Tensor.tolist, torch.Tensor.item
    monkeypatch.setattr(torch.Tensor, "cpu", lambda t, *a, **k: (moved.append(t.numel()), real_cpu(t, *a, **k))[1])
    monkeypatch.setattr(torch.Tensor, "tolist", lambda t: (moved.append(t.numel()), real_tolist(t))[1])
    monkeypatch.setattr(torch.Tensor, "item", lambda t: (moved.append(t.numel()), real_item(t))[1])
    g = CSRGraph(rowptr, col, val, (n, n))
    B = torch.randn(n, 64, device=dev)
    out = spmm_csr(g, B)
    out_t = spmm_csr(g.t(), B)
    torch.cuda.synchronize()
    monkeypatch.undo()
    assert moved and max(moved) <= 3, f"host transfers of {sorted(set(moved))} elements"
    assert float((spmm_csr(g, torch.ones(n, 8, device=dev)) - 1).abs().max()) <= 1e-5
    _assert_same_schedule(g)
    _assert_same_schedule(g.t())
    st = g.schedule_stats()
    assert st["n_long"] > 100 and st["n_chunks"] > st["n_long"] and st["n_items"] > n // 64
    assert torch.isfinite(out).all() and torch.isfinite(out_t).all()


@pytest.mark.parametrize("reduce", ["sum", "max"])
def test_native_coo_to_csr_matches_oracle(oracle, dev, reduce):
    """Unsorted COO with duplicates (the layout the reference emits, utils.py:407-414): duplicates
    reduced in STORAGE order — bitwise equal to a sequential host reduction."""
    from pygcn_amd import CSRGraph
    rows, cols, vals = gin.random_coo(700, 650, 9000, seed=77, duplicates=2500, hub_row=5, hub_deg=900,
                                      empty_rows=(0, 13, 699))
    g = CSRGraph.from_coo(torch.from_numpy(rows), torch.from_numpy(cols), torch.from_numpy(vals),
                          (700, 650), device=dev, reduce=reduce)
    # sequential reference: stable sort by (row, col), reduce runs left to right in fp32
    key = rows * 650 + cols
    order = np.argsort(key, kind="stable")
    ks, vs = key[order], vals[order]
    heads = np.flatnonzero(np.concatenate([[True], ks[1:] != ks[:-1]]))
    ref_val = np.empty(len(heads), np.float32)
    ends = np.concatenate([heads[1:], [len(ks)]])
    for i, (a, b) in enumerate(zip(heads, ends)):
        acc = np.float32(vs[a])
        for v in vs[a + 1:b]:
            acc = max(acc, np.float32(v)) if reduce == "max" else np.float32(acc + np.float32(v))
        ref_val[i] = acc
    uk = ks[heads]
    assert g.nnz == len(heads) and g.shape == (700, 650)
    assert np.array_equal(g.col.cpu().numpy(), (uk % 650).astype(np.int32))
    assert np.array_equal(g.val.cpu().numpy(), ref_val)                     # bitwise
    counts = np.bincount(uk // 650, minlength=700)
    assert np.array_equal(g.rowptr.cpu().numpy().astype(np.int64), np.concatenate([[0], np.cumsum(counts)]))
    if reduce == "sum":      # and it is the matrix the oracle's COO product multiplies with
        B = gin.dense((650, 16), 78)
        from pygcn_amd import spmm_csr
        assert_normwise(spmm_csr(g, torch.from_numpy(B).to(dev)).cpu(),
                        oracle.spmm_coo(rows, cols, vals, B, 700), 1e-5, "A·B")
    # degenerate inputs
    e = CSRGraph.from_coo(torch.empty(0, dtype=torch.int64), torch.empty(0, dtype=torch.int64),
                          torch.empty(0), (5, 4), device=dev)
    assert e.nnz == 0 and e.rowptr.tolist() == [0] * 6
    with pytest.raises(RuntimeError, match="out of range"):
        CSRGraph.from_coo(torch.tensor([0, 5]), torch.tensor([0, 0]), torch.ones(2), (5, 4), device=dev)


def test_reference_adjacency_recipe_on_the_device(dev):
    """cora.cites -> symmetric adjacency -> normalize(A + I) (pygcn/utils.py:360-368), built by the
    native ingest kernels, equals the adjacency the reference's own helpers built (fixture
    cora_graph.npz: CSR arrays captured through the imported reference)."""
    from pygcn_amd import CSRGraph
    z = load_golden("cora_graph.npz")
    g = CSRGraph.from_edge_list(z["edges"], int(z["n"]), device=dev)
    assert g.nnz == int(z["nnz"]) == 13264
    assert np.array_equal(g.rowptr.cpu().numpy(), z["csr_rowptr"])
    assert np.array_equal(g.col.cpu().numpy(), z["csr_col"])
    np.testing.assert_allclose(g.val.cpu().numpy(), z["csr_val"], rtol=2e-7)   # 1/sum vs sum^-1: 1 ulp
    ones = torch.ones(g.shape[1], 4, device=dev)
    from pygcn_amd import spmm_csr
    assert float((spmm_csr(g, ones) - 1).abs().max()) <= 1e-6


def test_cache_file_round_trip_without_replanning(dev, tmp_path, monkeypatch):
    from pygcn_amd import CSRGraph, spmm_csr
    from pygcn_amd.utils import rmat_graph
    n = 50_000
    rowptr, col, val = rmat_graph(n, 700_000, seed=9, device=dev)
    g = CSRGraph(rowptr, col, val, (n, n), item_cost=48)
    B = torch.randn(n, 256, device=dev)
    want, want_t = spmm_csr(g, B), spmm_csr(g.t(), B)
    path = str(tmp_path / "graph.pygcn")
    size = g.save(path)
    assert size > g.nnz * 16                                   # both matrices, 8 B per entry each
    # re-opening must not plan or transpose again
    def boom(*a, **k):
        raise AssertionError("planner / transpose ran on a cached graph")
    monkeypatch.setattr(CSRGraph, "_plan_arrays_device", boom)
    from pygcn_amd import _native
    real = _native.lib().gcn_csr_transpose_device
    h = CSRGraph.load(path, device=dev)
    assert h.shape == g.shape and h.nnz == g.nnz and h.item_cost == 48
    for a, b in ((h, g), (h.t(), g.t())):
        assert torch.equal(a.rowptr, b.rowptr) and torch.equal(a.col, b.col) and torch.equal(a.val, b.val)
        for k in ("items", "chunk_row", "chunk_e0", "long_row", "long_chunk0"):
            m = {"items": 2 * b._keep["n_items"], "chunk_row": b._keep["n_chunks"],
                 "chunk_e0": b._keep["n_chunks"], "long_row": b._keep["n_long"],
                 "long_chunk0": b._keep["n_long"] + 1}[k]
            assert torch.equal(a._keep[k][:m], b._keep[k][:m]), k
    assert torch.equal(spmm_csr(h, B), want) and torch.equal(spmm_csr(h.t(), B), want_t)   # bitwise
    assert real is _native.lib().gcn_csr_transpose_device


def test_cache_file_with_a_foreign_or_garbled_schedule_is_rejected(dev, tmp_path):
    """the schedule of a cache file goes to the kernels as it is — a file from another
    ABI version, or one whose (CRC-consistent) schedule is not the schedule of its row pointer,
    must be refused instead of producing wrong sums / out-of-range reads."""
    from pygcn_amd import CSRGraph, _native, cache, spmm_csr
    from pygcn_amd.utils import rmat_graph
    n = 20_000
    rowptr, col, val = rmat_graph(n, 400_000, seed=3, device=dev)
    g = CSRGraph(rowptr, col, val, (n, n))
    path = str(tmp_path / "g.pygcn")
    g.save(path)
    B = torch.randn(n, 64, device=dev)
    assert torch.equal(spmm_csr(CSRGraph.load(path, device=dev), B), spmm_csr(g, B))
    meta, arr = cache.read_file(path, mmap=False)
    assert meta["abi_version"] == _native.GCN_ABI_VERSION

    def rewritten(name, meta_edit=None, **edits):
        m = dict(meta, **(meta_edit or {}))
        a = {k: np.array(v) for k, v in arr.items()}
        for k, fn in edits.items():
            a[k.replace("__", ".")] = fn(a[k.replace("__", ".")])
        q = str(tmp_path / name)
        cache.write_file(q, m, a)          # (sections get fresh, VALID checksums)
        return q
    with pytest.raises(cache.CacheFormatError, match="ABI version"):
        CSRGraph.load(rewritten("abi", {"abi_version": _native.GCN_ABI_VERSION - 1}), device=dev)
    with pytest.raises(cache.CacheFormatError, match="ABI version"):
        CSRGraph.load(rewritten("abi2", {"abi_version": None}), device=dev, verify=False)

    def shift_items(a):
        b = a.copy()
        b[3] += 1                           # second item now ends one row late
        return b
    with pytest.raises(cache.CacheFormatError, match="schedule"):
        CSRGraph.load(rewritten("items", a__items=shift_items), device=dev)
    with pytest.raises(cache.CacheFormatError, match="schedule"):
        CSRGraph.load(rewritten("chunk", t__chunk_e0=lambda a: a[::-1].copy()), device=dev)
    with pytest.raises(cache.CacheFormatError, match="schedule"):       # another planner setting
        CSRGraph.load(rewritten("cost", {"item_cost": 32}), device=dev)
    # structural damage is caught even with verify=False (no CRC pass, no planner run)
    with pytest.raises(cache.CacheFormatError, match="schedule"):
        CSRGraph.load(rewritten("range", a__items=lambda a: np.where(np.arange(a.size) == 1, n + 7, a)
                                .astype(a.dtype)), device=dev, verify=False)
    with pytest.raises(cache.CacheFormatError, match="schedule"):
        CSRGraph.load(rewritten("short", a__long_chunk0=lambda a: a[:-1].copy()), device=dev, verify=False)
