"""CPU-side tests of the product's host logic: the C-ABI library loads and exports every symbol
of include/gcn_spmm.h, the native planner and transpose are correct, the module surface matches
the reference's, and the product refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import inputs as gin
from conftest import ROOT, assert_normwise, load_golden
from pygcn_amd import _native


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gcn_spmm.h")).read()
    declared = set(re.findall(r"\b(gcn_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
    L = _native.lib()
    for sym in declared:
        assert getattr(L, sym) is not None
    assert L.gcn_abi_version() == _native.GCN_ABI_VERSION
    for name in ("GCN_ABI_VERSION", "GCN_DEFAULT_ITEM_COST", "GCN_DEFAULT_LONG_THRESH",
                 "GCN_DTYPE_F32", "GCN_DTYPE_BF16"):
        assert int(re.search(rf"#define {name}\s+(\d+)", hdr).group(1)) == getattr(_native, name)


def test_plan_struct_layout_matches_header():
    # 3*8 + 8 + 4 + 4 + 2*8 + (8+8) + (8+8+8) + (8+8+8) bytes, natural alignment, no padding holes
    assert ctypes.sizeof(_native.GcnCsrPlan) == 120
    assert _native.GcnCsrPlan.col.offset == 40 and _native.GcnCsrPlan.n_items.offset == 56


def _plan(rowptr, item_cost=0, long_thresh=0):
    L = _native.lib()
    rp = np.ascontiguousarray(rowptr)
    is64 = int(rp.dtype == np.int64)
    n = len(rp) - 1
    ni, nc, nl = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    rc = L.gcn_plan_count_host(rp.ctypes.data, is64, n, item_cost, long_thresh,
                               ctypes.byref(ni), ctypes.byref(nc), ctypes.byref(nl))
    assert rc == 0
    ni, nc, nl = ni.value, nc.value, nl.value
    items = np.zeros(max(2 * ni, 1), np.int32)
    crow, ce0 = np.zeros(max(nc, 1), np.int32), np.zeros(max(nc, 1), np.int64)
    lrow, lc0 = np.zeros(max(nl, 1), np.int32), np.zeros(nl + 1, np.int32)
    rc = L.gcn_plan_fill_host(rp.ctypes.data, is64, n, item_cost, long_thresh,
                              items.ctypes.data, ni, crow.ctypes.data, ce0.ctypes.data, nc,
                              lrow.ctypes.data, lc0.ctypes.data, nl)
    assert rc == 0
    return items[:2 * ni].reshape(-1, 2), crow[:nc], ce0[:nc], lrow[:nl], lc0


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_planner_covers_every_row_exactly_once(dtype, seed):
    rng = np.random.default_rng(seed)
    n = 5000
    deg = rng.integers(0, 12, size=n)
    deg[rng.integers(0, n, size=20)] = rng.integers(257, 3000, size=20)   # long rows
    deg[rng.integers(0, n, size=200)] = 0                                  # empty rows
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(dtype)
    L = (256, 0, 1000)[seed]                  # explicit, library default, explicit
    items, crow, ce0, lrow, lc0 = _plan(rp, long_thresh=L)
    L = L or _native.GCN_DEFAULT_LONG_THRESH
    covered = np.zeros(n, np.int32)
    for ra, rb in items:
        assert 0 < rb - ra <= 64
        covered[ra:rb] += 1
        cost = (rp[rb] - rp[ra]) + (rb - ra)
        assert cost <= 64 or rb - ra == 1 or cost - (deg[rb - 1] + 1) < 64
        assert (deg[ra:rb] <= L).all()
    covered[lrow] += 1
    assert (covered == 1).all()
    np.testing.assert_array_equal(np.sort(lrow), np.nonzero(deg > L)[0])
    for j, r in enumerate(lrow):
        c0, c1 = lc0[j], lc0[j + 1]
        assert c1 - c0 == -(-deg[r] // L)
        assert (crow[c0:c1] == r).all()
        np.testing.assert_array_equal(ce0[c0:c1], rp[r] + L * np.arange(c1 - c0))


def test_planner_rejects_bad_input():
    L = _native.lib()
    rp = np.array([0, 5, 3], np.int32)   # not monotone
    ni = ctypes.c_int64()
    rc = L.gcn_plan_count_host(rp.ctypes.data, 0, 2, 0, 0, ctypes.byref(ni), None, None)
    assert rc == -1 and b"monotone" in L.gcn_last_error()
    with pytest.raises(RuntimeError, match="monotone"):
        _native.check(rc, "gcn_plan_count_host")


def test_spmm_argument_errors_without_gpu():
    """Argument validation happens before any launch, so it is testable on CPU."""
    L = _native.lib()
    assert ctypes.sizeof(_native.GcnEpilogue) == 120      # (ABI 22: + drop_row_base, c_absmax)
    assert L.gcn_spmm_csr(None, 0, None, 0, None, 0, 4, None, 0, None, 0, None) == -1
    p = _native.GcnCsrPlan()
    p.n_rows, p.n_cols, p.nnz = 4, 4, 0
    assert L.gcn_spmm_csr(ctypes.byref(p), 7, None, 4, None, 4, 4, None, 0, None, 0, None) == -1
    assert b"dtype" in L.gcn_last_error()
    assert L.gcn_spmm_workspace_bytes(ctypes.byref(p), 256) == 0
    p.n_chunks = 3
    assert L.gcn_spmm_workspace_bytes(ctypes.byref(p), 256) == 3 * 256 * 4


def test_transpose_host_matches_oracle(oracle):
    rows, cols, vals = gin.random_coo(60, 90, 700, seed=11, duplicates=50, empty_rows=(3, 4))
    a = oracle.CSR.from_coo(rows, cols, vals, (60, 90))
    L = _native.lib()
    rp32 = a.rowptr.astype(np.int32)
    rpt, ct, vt = np.zeros(91, np.int32), np.zeros(a.nnz, np.int32), np.zeros(a.nnz, np.float32)
    rc = L.gcn_csr_transpose_host(rp32.ctypes.data, 0, a.col.ctypes.data, a.val.ctypes.data,
                                  60, 90, rpt.ctypes.data, ct.ctypes.data, vt.ctypes.data)
    assert rc == 0
    o_rp, o_c, o_v = oracle.csr_transpose(a.rowptr, a.col, a.val, 90)
    np.testing.assert_array_equal(rpt, o_rp)
    np.testing.assert_array_equal(ct, o_c)
    np.testing.assert_array_equal(vt, o_v)


def test_module_surface_matches_reference():
    from pygcn_amd import GCN, GraphConvolution
    g1 = load_golden("g1_init.npz")
    torch.manual_seed(42)
    gc1, gc2 = GraphConvolution(1433, 16), GraphConvolution(16, 7)
    # same RNG stream as the reference's init (layers.py:23-29): bit-identical parameters
    np.testing.assert_array_equal(gc1.weight.detach().numpy(), g1["gc1_weight"])
    np.testing.assert_array_equal(gc1.bias.detach().numpy(), g1["gc1_bias"])
    np.testing.assert_array_equal(gc2.weight.detach().numpy(), g1["gc2_weight"])
    np.testing.assert_array_equal(gc2.bias.detach().numpy(), g1["gc2_bias"])
    assert repr(gc1) == str(g1["repr"])
    assert (gc1.in_features, gc1.out_features) == (1433, 16)
    nb = GraphConvolution(256, 256, bias=False)
    assert nb.bias is None and list(nb.state_dict()) == ["weight"]
    torch.manual_seed(7)
    nb = GraphConvolution(256, 256, bias=False)
    np.testing.assert_array_equal(nb.weight.detach().numpy()[:4], g1["nobias_weight_head"])
    m = GCN(nfeat=1433, nhid=16, nclass=7, dropout=0.5)
    assert list(m.state_dict()) == ["gc1.weight", "gc1.bias", "gc2.weight", "gc2.bias"]


def test_flat_imports_like_the_reference():
    """`from layers import GraphConvolution`, `from models import GCN`,
    `from utils import load_data, accuracy` with cwd = the package dir (models.py:4,
    train.py:15-16)."""
    import subprocess
    import sys
    code = ("from layers import GraphConvolution; from models import GCN; "
            "from utils import load_data, accuracy, normalize, sparse_mx_to_torch_sparse_tensor; "
            "m = GCN(8, 4, 3, 0.5); print(type(m.gc1).__module__, m.gc1)")
    out = subprocess.run([sys.executable, "-c", code], cwd=os.path.join(ROOT, "pygcn_amd"),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "GraphConvolution (8 -> 4)" in out.stdout and out.stdout.startswith("layers ")


def test_load_data_reproduces_reference_adjacency():
    from pygcn_amd.utils import load_data
    g = load_golden("cora_graph.npz")
    adj, feats, labels, itr, iva, ite = load_data()
    assert adj.layout == torch.sparse_coo and adj.shape == (2708, 2708)
    assert adj._indices().dtype == torch.int64 and adj._values().dtype == torch.float32
    np.testing.assert_array_equal(adj._indices()[0].numpy(), g["coo_row"])
    np.testing.assert_array_equal(adj._indices()[1].numpy(), g["coo_col"])
    np.testing.assert_allclose(adj._values().numpy(), g["coo_val"], rtol=1e-7)
    assert feats.shape == (2708, 1433) and labels.shape == (2708,)
    np.testing.assert_allclose(feats.numpy(), gin.cora_features(), rtol=3e-7)
    np.testing.assert_array_equal(labels.numpy(), gin.cora_labels())
    assert (len(itr), len(iva), len(ite)) == (140, 300, 1000)


def test_product_refuses_cpu_tensors():
    """No CPU fallback: the layer and the op raise on non-HIP tensors."""
    from pygcn_amd import GraphConvolution, sparse_mm as spmm
    layer = GraphConvolution(4, 3)
    adj = torch.eye(5).to_sparse()
    with pytest.raises(RuntimeError, match="HIP device"):
        layer(torch.randn(5, 4), adj)
    with pytest.raises(RuntimeError, match="HIP device"):
        spmm(adj, torch.randn(5, 3))


def test_rmat_generator_shapes():
    from pygcn_amd.utils import rmat_graph
    n = 5000
    rowptr, col, val = rmat_graph(n, 40000, seed=42, device="cpu")
    assert rowptr.numel() == n + 1 and rowptr[0] == 0 and rowptr[-1] == col.numel()
    deg = (rowptr[1:] - rowptr[:-1])
    assert int(deg.min()) >= 1                      # self-loops
    rows = torch.repeat_interleave(torch.arange(n), deg.long())
    assert bool(((rows == col).sum() == n))         # exactly one diagonal entry per row
    sums = torch.zeros(n).index_add_(0, rows, val)
    np.testing.assert_allclose(sums.numpy(), 1.0, atol=1e-5)
    key = rows * n + col
    assert bool((key[1:] > key[:-1]).all())         # sorted, no duplicates
    assert int(deg.max()) > 20 * float(deg.float().median())   # skewed


def test_ctypes_structs_follow_the_header_field_by_field():
    """The ctypes mirrors must list the fields of `struct gcn_csr_plan` / `struct gcn_epilogue`
    in the header's order with matching widths (a silent mismatch would shift every later
    field)."""
    hdr = open(os.path.join(ROOT, "include", "gcn_spmm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    width = {"int32_t": 4, "int64_t": 8, "uint64_t": 8, "float": 4}
    for cname, mirror in (("gcn_csr_plan", _native.GcnCsrPlan), ("gcn_epilogue", _native.GcnEpilogue),
                          ("gcn_gemm_epilogue", _native.GcnGemmEpilogue)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(?:const )?([A-Za-z0-9_]+) (\*?)([A-Za-z0-9_]+)$", decl)
            assert m, decl
            fields.append((m.group(3), 8 if m.group(2) else width[m.group(1)]))
        got = [(n, ctypes.sizeof(t)) for n, t in mirror._fields_]
        assert got == fields, (cname, got, fields)


def test_registered_operator_surface():
    """the product is a registered PyTorch operator with an autograd formula, a
    fake (meta) kernel for tracing, and NO CPU kernel that computes anything."""
    import pygcn_amd  # noqa: F401  (registers the operator)
    op = torch.ops.pygcn_amd.spmm_csr
    schema = str(op.default._schema)
    assert "Tensor rowptr" in schema and "Tensor? bias" in schema and "n_cols" in schema
    rp = torch.tensor([0, 1, 3], dtype=torch.int32)
    col = torch.tensor([0, 0, 1], dtype=torch.int32)
    val = torch.ones(3)
    with pytest.raises(RuntimeError, match="HIP device"):
        op(rp, col, val, torch.randn(2, 4), None, 2)
    # shape inference without any kernel running (what torch.compile traces through)
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        out = op(torch.empty(6, dtype=torch.int32), torch.empty(9, dtype=torch.int32),
                 torch.empty(9), torch.empty(7, 16), None, 7)
        assert tuple(out.shape) == (5, 16) and out.dtype == torch.float32
        with pytest.raises(RuntimeError, match="size mismatch"):
            op(torch.empty(6, dtype=torch.int32), torch.empty(9, dtype=torch.int32),
               torch.empty(9), torch.empty(8, 16), None, 7)


def test_package_exports_module_not_shadowed():
    """`pygcn_amd.spmm` is the module (a re-exported function used to shadow it)."""
    import types
    import pygcn_amd
    assert isinstance(pygcn_amd.spmm, types.ModuleType)
    assert callable(pygcn_amd.sparse_mm) and callable(pygcn_amd.spmm.spmm)


def test_cache_file_format_round_trip_and_rejections(tmp_path):
    """pygcn_amd/cache.py (binary CSR cache): versioned, checksummed, pure host."""
    from pygcn_amd import cache
    rng = np.random.default_rng(0)
    arrays = {"a.rowptr": np.arange(101, dtype=np.int32), "a.col": rng.integers(0, 100, 777).astype(np.int32),
              "a.val": rng.random(777).astype(np.float32), "a.chunk_e0": np.arange(5, dtype=np.int64),
              "a.long_row": np.empty(0, np.int32)}
    path = str(tmp_path / "g.bin")
    size = cache.write_file(path, {"n_rows": 100, "note": "x"}, arrays)
    assert size == os.path.getsize(path)
    meta, got = cache.read_file(path)
    assert meta == {"n_rows": 100, "note": "x"} and list(got) == list(arrays)
    for k, v in arrays.items():
        assert got[k].dtype == v.dtype and np.array_equal(got[k], v), k
    raw = bytearray(open(path, "rb").read())

    def rejected(mutated, what):
        p = str(tmp_path / "bad.bin")
        open(p, "wb").write(bytes(mutated))
        with pytest.raises(cache.CacheFormatError, match=what):
            cache.read_file(p)
    rejected(b"NOTACSR!" + raw[8:], "bad magic")
    rejected(raw[:8] + (99).to_bytes(4, "little") + raw[12:], "format version 99")
    rejected(raw[:len(raw) - 40], "outside the file")
    import json
    hlen = int.from_bytes(raw[12:16], "little")
    sec = {x["name"]: x for x in json.loads(raw[16:16 + hlen].decode())["sections"]}
    flipped = bytearray(raw)
    flipped[sec["a.val"]["offset"] + 401] ^= 0x40           # one bit inside a section's payload
    rejected(flipped, "CRC-32")
    cache.read_file(str(tmp_path / "bad.bin"), verify=False)   # (the same file opens unverified)


def test_row_sparse_gradient_carrier_mechanics():
    """pygcn_amd/rowgrad.py on CPU tensors (pure autograd plumbing, no kernels): a RowGrad travels
    between custom autograd nodes, materialises for every other consumer, and `output[idx]` is
    intercepted only for an int64 index tensor on a grad-requiring 2-D output."""
    from pygcn_amd.rowgrad import RowGrad, RowSelectable

    class Layer(torch.autograd.Function):
        seen = []

        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            Layer.seen.append(type(g).__name__)
            if isinstance(g, RowGrad):
                return RowGrad(g.rows, g.values @ w.t(), x.shape[0]), x.index_select(0, g.rows).t() @ g.values
            return g @ w.t(), x.t() @ g

    torch.manual_seed(0)
    x = torch.randn(12, 4, requires_grad=True)
    w1, w2 = torch.randn(4, 4, requires_grad=True), torch.randn(4, 3, requires_grad=True)
    idx = torch.tensor([7, 1, 7, 3])                                  # unsorted, with a duplicate

    def run(select):
        for t in (x, w1, w2):
            t.grad = None
        out = Layer.apply(Layer.apply(x, w1), w2)
        select(out).sum().backward()
        return [t.grad.clone() for t in (x, w1, w2)]
    Layer.seen.clear()
    got = run(lambda o: o.as_subclass(RowSelectable)[idx])
    assert Layer.seen == ["RowGrad", "RowGrad"]
    Layer.seen.clear()
    want = run(lambda o: o[idx])
    assert Layer.seen == ["Tensor", "Tensor"]
    for a, b in zip(got, want):
        assert torch.allclose(a, b, atol=1e-6)
    assert not isinstance(got[0], RowGrad)                            # the leaf received a dense tensor
    # a second consumer: the RowGrad is added to a dense gradient (materialised by dispatch)
    Layer.seen.clear()
    both = run(lambda o: o.as_subclass(RowSelectable)[idx].sum() + 0.5 * o.sum())
    assert Layer.seen == ["Tensor", "Tensor"]
    ref = run(lambda o: o[idx].sum() + 0.5 * o.sum())
    for a, b in zip(both, ref):
        assert torch.allclose(a, b, atol=1e-6)
    # what is NOT intercepted: list / slice indices, no-grad outputs; results are plain tensors
    out = Layer.apply(Layer.apply(x, w1), w2).as_subclass(RowSelectable)
    assert type(out[idx.tolist()]) is torch.Tensor and out[idx.tolist()].grad_fn.name() != "SelectRowsFunctionBackward"
    assert type(out[2:5]) is torch.Tensor and type(out + 1) is torch.Tensor and type(out.detach()) is torch.Tensor
    from pygcn_amd.rowgrad import LossRows
    sel = out[idx]             # (a LossRows alias of the selection: upstream's next call is F.nll_loss)
    assert isinstance(sel, LossRows) and type(sel + 1) is torch.Tensor
    assert "SelectRowsFunctionBackward" in (sel.grad_fn.name(), sel.grad_fn.next_functions[0][0].name())
    with torch.no_grad():
        assert out.detach().as_subclass(RowSelectable)[idx].grad_fn is None
    rg = RowGrad(torch.tensor([2, 0, 2]), torch.ones(3, 2), 4)
    assert tuple(rg.shape) == (4, 2) and torch.equal(rg.dense(), torch.tensor([[1., 1.], [0., 0.], [2., 2.], [0., 0.]]))
    assert torch.equal(rg + 1, rg.dense() + 1)


def test_package_ships_its_own_cora_fixture():
    """The package must not reach into tests/: its default dataset is its own copy of the edge-list
    fixture — byte-identical to the golden one the generating script writes."""
    import pygcn_amd.utils as U
    assert os.path.dirname(U.DEFAULT_CORA).endswith(os.path.join("pygcn_amd", "data"))
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cora_graph.npz")
    assert open(U.DEFAULT_CORA, "rb").read() == open(golden, "rb").read()
    pkg = os.path.dirname(os.path.abspath(U.__file__))
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            src = open(os.path.join(pkg, name)).read()
            assert '"tests"' not in src and "from tests" not in src and "import tests" not in src, name
