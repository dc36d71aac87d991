#!/usr/bin/env python3
"""Generate golden vectors by running the UNMODIFIED reference (Coder-Yu/QRec) in-process.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Needs /root/reference (present only in the build container, never on the GPU box); the
fixtures it writes next to this file are committed and are what the tests read.

How the reference is made importable without editing it:
  * ``sys.modules`` stubs for ``numba`` (jit = identity), ``mkl`` and ``tensorflow``
    (none are installed; only the numpy path and the pure-python sampler / scipy
    adjacency builder are executed);
  * cwd = scratch dir holding a ``dataset -> /root/reference/dataset`` symlink, because
    conf paths, ./log and ./results are cwd-relative;
  * RNGs seeded (the reference never seeds): ``random.seed(s); np.random.seed(s)``.

What is captured, per case: the (u,i,j) index stream of every SGD step, P/Q and the loss
after every epoch, the learning-rate schedule, the python ``random`` state at the end, the
recommendation lists and the measure strings.  Big arrays are stored as .npz; streams that
would be large are stored as a sha256 plus a head/tail sample.
"""
import hashlib
import io
import json
import os
import random
import sys
import tempfile
import types
from contextlib import redirect_stdout

import numpy as np

REF = "/root/reference"
OUT = os.environ.get("QREC_GOLDEN_OUT") or os.path.dirname(os.path.abspath(__file__))      # QREC_GOLDEN_OUT: regenerate into another directory (tests/test_oracle_golden.py compares)


def install_stubs():
    nb = types.ModuleType("numba"); nb.jit = lambda *a, **k: (lambda f: f)
    mkl = types.ModuleType("mkl"); mkl.set_num_threads = lambda n: None; mkl.get_max_threads = lambda: 1
    tf = types.ModuleType("tensorflow")
    sys.modules.update(numba=nb, mkl=mkl, tensorflow=tf)
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def write_conf(path, **kv):
    with open(path, "w") as f:
        for k, v in kv.items():
            f.write(f"{k.replace('__', '.')}={v}\n")


def run_numpy_model(conf_path, seed, model_mod, model_name, hook_attr, social=False):
    """Run QRec(conf) end to end with instrumentation of the per-sample step."""
    from QRec import QRec
    from util.config import ModelConf
    import importlib
    mod = importlib.import_module(model_mod)
    cls = getattr(mod, model_name)
    rec = {"steps": [], "epochs": []}

    if hook_attr == "optimization":       # BPR: record (u,i,j)
        orig = cls.optimization
        def optimization(self, u, i, j):
            rec["steps"].append((u, i, j)); return orig(self, u, i, j)
        cls.optimization = optimization
    orig_conv = cls.isConverged
    def isConverged(self, epoch):
        loss_before = self.loss; lr_used = self.lRate
        r = orig_conv(self, epoch)
        rec["epochs"].append(dict(epoch=epoch, loss=float(loss_before), lr_used=float(lr_used),
                                  lr_next=float(self.lRate), converged=bool(r),
                                  P=self.P.copy(), Q=self.Q.copy(),
                                  **({"Bu": self.Bu.copy(), "Bi": self.Bi.copy()} if hasattr(self, "Bu") else {}),
                                  **({"Y": self.Y.copy()} if hasattr(self, "Y") else {}),
                                  order=[(self.data.user[a], self.data.item[b], c)
                                         for a, b, c in self.data.trainingData]
                                  if hook_attr == "mf" else None))
        return r
    cls.isConverged = isConverged
    orig_init = cls.initModel
    def initModel(self):
        orig_init(self); rec["P0"] = self.P.copy(); rec["Q0"] = self.Q.copy()
        if hasattr(self, "Bu"):
            rec["Bu0"] = self.Bu.copy(); rec["Bi0"] = self.Bi.copy()
        if hasattr(self, "Y"):
            rec["Y0"] = self.Y.copy()
        rec["order0"] = [(self.data.user[a], self.data.item[b], c) for a, b, c in self.data.trainingData]
    cls.initModel = initModel
    orig_eval = cls.evalRanking
    holder = {}
    import base.recommender as br
    orig_rm = br.Measure.rankingMeasure
    def rankingMeasure(origin, res, N):
        holder["recList"] = res; holder["origin"] = origin
        return orig_rm(origin, res, N)
    br.Measure.rankingMeasure = staticmethod(rankingMeasure)

    random.seed(seed); np.random.seed(seed)
    buf = io.StringIO()
    with redirect_stdout(buf):
        q = QRec(ModelConf(conf_path))
        # QRec.execute() instantiates through eval(); do the same thing by hand to keep
        # a handle on the model object
        rec["raw_relation"] = [list(r) for r in q.relation] if social else None     # before the model prunes the list in place
        m = cls(q.config, q.trainingData, q.testData, q.relation) if social else cls(q.config, q.trainingData, q.testData)
        measure = m.execute()
    cls.isConverged = orig_conv; cls.initModel = orig_init
    br.Measure.rankingMeasure = staticmethod(orig_rm)
    if hook_attr == "optimization":
        cls.optimization = orig
    rec["model"] = m; rec["measure"] = measure; rec["holder"] = holder
    rec["py_state"] = random.getstate()
    rec["train_rows"] = q.trainingData; rec["test_rows"] = q.testData
    return rec


def pack_bpr(rec, name, keep_full_stream):
    m = rec["model"]
    steps = np.array(rec["steps"], dtype=np.int32).reshape(-1, 3)
    n_per_epoch = steps.shape[0] // len(rec["epochs"])
    # training data as the reference loaded/split it (ids as assigned by data/rating.py)
    tr = np.array([(m.data.user[a], m.data.item[b]) for a, b, _ in rec["order0"]], dtype=np.int32) \
        if False else None
    train_uid = np.array([m.data.user[r[0]] for r in rec["train_rows"]], dtype=np.int32)
    train_iid = np.array([m.data.item[r[1]] for r in rec["train_rows"]], dtype=np.int32)
    train_r = np.array([r[2] for r in rec["train_rows"]], dtype=np.float64)
    # test rows: names may be unknown to the train id maps -> -1
    test_uid = np.array([m.data.user.get(r[0], -1) for r in rec["test_rows"]], dtype=np.int32)
    test_iid = np.array([m.data.item.get(r[1], -1) for r in rec["test_rows"]], dtype=np.int32)
    test_uname = np.array([r[0] for r in rec["test_rows"]]); test_iname = np.array([r[1] for r in rec["test_rows"]])
    arrays = dict(P0=rec["P0"], Q0=rec["Q0"], train_uid=train_uid, train_iid=train_iid,
                  train_r=train_r, test_uid=test_uid, test_iid=test_iid,
                  test_uname=test_uname, test_iname=test_iname,
                  py_state=np.array(rec["py_state"][1], dtype=np.uint32))
    for k, e in enumerate(rec["epochs"]):
        arrays[f"P{k+1}"] = e["P"]; arrays[f"Q{k+1}"] = e["Q"]
    if keep_full_stream:
        arrays["steps"] = steps
    else:
        arrays["steps_head"] = steps[:4096]; arrays["steps_tail"] = steps[-4096:]
    # recommendation lists (ids + scores) in testSet_u order
    rl = rec["holder"]["recList"]
    users = list(rl.keys())
    N = max(len(v) for v in rl.values())
    ids = np.full((len(users), N), -1, dtype=np.int32); sc = np.zeros((len(users), N))
    for a, un in enumerate(users):
        for b, (iname, s) in enumerate(rl[un]):
            ids[a, b] = m.data.item[iname]; sc[a, b] = s
    arrays["rec_users"] = np.array([m.data.user.get(un, -1) for un in users], dtype=np.int32)
    arrays["rec_user_names"] = np.array(users)
    arrays["rec_ids"] = ids; arrays["rec_scores"] = sc
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    meta = dict(name=name, n_users=len(m.data.user), n_items=len(m.data.item),
                n_train=len(rec["train_rows"]), n_test=len(rec["test_rows"]),
                triplets_per_epoch=int(n_per_epoch), stream_sha256=sha(steps),
                epochs=[{k: v for k, v in e.items() if k not in ("P", "Q", "order")} for e in rec["epochs"]],
                measure=rec["measure"], emb_size=m.emb_size, regU=m.regU, regI=m.regI,
                topN=m.ranking["-topN"])
    return meta


def case_bpr_filmtrust(tmp):
    conf = os.path.join(tmp, "bpr_ft.conf")
    write_conf(conf, ratings="./dataset/FilmTrust/trainset.txt", ratings__setup="-columns 0 1 2",
               model__name="BPR", evaluation__setup="-testSet ./dataset/FilmTrust/testset.txt -b 1",
               item__ranking="on -topN 10,20", num__factors="8", num__max__epoch="3",
               batch_size="1500", learnRate="-init 0.05 -max 1",
               reg__lambda="-u 0.01 -i 0.01 -b 0.2 -s 0.2", output__setup="off -dir ./results/")
    rec = run_numpy_model(conf, 20260923, "model.ranking.BPR", "BPR", "optimization")
    meta = pack_bpr(rec, "bpr_filmtrust", keep_full_stream=True)
    meta["seed"] = 20260923; meta["conf"] = open(conf).read()
    return meta


def case_bpr_lastfm(tmp):
    conf = os.path.join(tmp, "bpr_lfm.conf")
    # stock config/BPR.conf with fewer epochs / factors so that the fixture stays small
    write_conf(conf, ratings="./dataset/lastfm/ratings.txt", ratings__setup="-columns 0 1 2",
               model__name="BPR", evaluation__setup="-ap 0.2 -b 1", item__ranking="on -topN 20",
               num__factors="16", num__max__epoch="2", batch_size="1500",
               learnRate="-init 0.01 -max 1", reg__lambda="-u 0.001 -i 0.001 -b 0.2 -s 0.2",
               output__setup="off -dir ./results/")
    rec = run_numpy_model(conf, 7, "model.ranking.BPR", "BPR", "optimization")
    # keep the fixture small: P0/Q0 are regenerable (np.random.seed(7); rand(U,d)/3;
    # rand(I,d)/3 -- base/iterativeRecommender.py:37-38), so store only their sha256;
    # keep the final P in full and every 4th row of the final Q plus its sha256.
    last = len(rec["epochs"])
    meta = pack_bpr(rec, "bpr_lastfm", keep_full_stream=False)
    z = dict(np.load(os.path.join(OUT, "bpr_lastfm.npz")))
    for k in range(1, last):
        z.pop(f"P{k}"); z.pop(f"Q{k}")
    meta["P0_sha256"] = sha(z.pop("P0")); meta["Q0_sha256"] = sha(z.pop("Q0"))
    meta["Qlast_sha256"] = sha(z[f"Q{last}"])
    z[f"Q{last}_every4"] = z.pop(f"Q{last}")[::4].copy()
    # which raw rows the -ap split sent to the test set (util/dataSplit.py:9-26)
    n_raw = len(rec["train_rows"]) + len(rec["test_rows"])
    mask = np.zeros(n_raw, dtype=bool); a = b = 0
    from util.io import FileIO
    from util.config import ModelConf
    with redirect_stdout(io.StringIO()):
        raw = FileIO.loadDataSet(ModelConf(conf), "./dataset/lastfm/ratings.txt", binarized=True, threshold=1.0)
    assert len(raw) == n_raw
    for k, row in enumerate(raw):
        if b < len(rec["test_rows"]) and rec["test_rows"][b] == row:
            mask[k] = True; b += 1
        else:
            assert rec["train_rows"][a] == row; a += 1
    z["split_is_test"] = mask
    np.savez_compressed(os.path.join(OUT, "bpr_lastfm.npz"), **z)
    meta["seed"] = 7; meta["conf"] = open(conf).read(); meta["kept_epochs"] = [last]
    return meta


def _case_rating_mf(tmp, model, fixture, seed, factors, lr, reg, epochs=3, **extra):
    conf = os.path.join(tmp, fixture + ".conf")
    write_conf(conf, ratings="./dataset/FilmTrust/trainset.txt", ratings__setup="-columns 0 1 2",
               model__name=model, evaluation__setup="-testSet ./dataset/FilmTrust/testset.txt",
               item__ranking="off -topN 10", num__factors=str(factors), num__max__epoch=str(epochs),
               batch_size="1024", learnRate="-init %s -max 1" % lr,
               reg__lambda=reg, output__setup="off -dir ./results/", **extra)
    rec = run_numpy_model(conf, seed, "model.rating." + model, model, "mf")
    m = rec["model"]
    arrays = dict(P0=rec["P0"], Q0=rec["Q0"],
                  order0=np.array([(a, b) for a, b, _ in rec["order0"]], dtype=np.int32),
                  rating0=np.array([c for _, _, c in rec["order0"]], dtype=np.float64),
                  py_state=np.array(rec["py_state"][1], dtype=np.uint32))
    for k, e in enumerate(rec["epochs"]):
        arrays[f"P{k+1}"] = e["P"]; arrays[f"Q{k+1}"] = e["Q"]
        arrays[f"order{k+1}"] = np.array([(a, b) for a, b, _ in e["order"]], dtype=np.int32)
        if "Bu" in e:
            arrays[f"Bu{k+1}"] = e["Bu"]; arrays[f"Bi{k+1}"] = e["Bi"]
        if "Y" in e:
            arrays[f"Y{k+1}"] = e["Y"]
    if "Y0" in rec:
        arrays["Y0"] = rec["Y0"]
    if "Bu0" in rec:
        arrays["Bu0"] = rec["Bu0"]; arrays["Bi0"] = rec["Bi0"]
    # test predictions appended by evalRatings: [user,item,rating,pred]
    arrays["test_pred"] = np.array([r[3] for r in m.data.testData], dtype=np.float64)
    arrays["test_rating"] = np.array([r[2] for r in m.data.testData], dtype=np.float64)
    arrays["test_uid"] = np.array([m.data.user.get(r[0], -1) for r in m.data.testData], dtype=np.int32)
    arrays["test_iid"] = np.array([m.data.item.get(r[1], -1) for r in m.data.testData], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, fixture + ".npz"), **arrays)
    return dict(name=fixture, seed=seed, conf=open(conf).read(),
                n_users=len(m.data.user), n_items=len(m.data.item), n_train=len(rec["train_rows"]),
                epochs=[{k: v for k, v in e.items() if k not in ("P", "Q", "order", "Bu", "Bi", "Y")} for e in rec["epochs"]],
                measure=rec["measure"], globalMean=m.data.globalMean,
                rScale=[float(x) for x in m.data.rScale], regU=m.regU, regI=m.regI, regB=m.regB)


def case_basicmf(tmp):
    # BASELINE.json config #1: BasicMF on FilmTrust, d=10, reference numpy path (no -tf)
    return _case_rating_mf(tmp, "BasicMF", "basicmf_filmtrust", 1, 10, 0.03, "-u 0.05 -i 0.05 -b 0.1 -s 0.1")


def case_pmf(tmp):
    return _case_rating_mf(tmp, "PMF", "pmf_filmtrust", 2, 10, 0.02, "-u 0.01 -i 0.01 -b 0.1 -s 0.1")


def case_svd(tmp):
    return _case_rating_mf(tmp, "SVD", "svd_filmtrust", 3, 10, 0.005, "-u 0.01 -i 0.02 -b 0.02 -s 0.1")


def case_ee(tmp):
    return _case_rating_mf(tmp, "EE", "ee_filmtrust", 4, 10, 0.005, "-u 0.005 -i 0.005 -b 0.005 -s 0.1")


def case_svdpp(tmp):
    out = _case_rating_mf(tmp, "SVDPlusPlus", "svdpp_filmtrust", 5, 10, 0.02, "-u 0.01 -i 0.01 -b 0.1 -s 0.1", epochs=2,
                          SVDPlusPlus="-y 0.01")
    out["regY"] = 0.01
    return out


def case_tbpr_filmtrust(tmp):
    """model/ranking/TBPR.py (numpy path, runs unmodified): tie strengths, strong / weak / joint item sets, the chained
    pairwise updates per positive item and the per-user regularisation terms of the loss, on FilmTrust + trust.txt.
    The order of every user's joint-item list comes from a Python set of strings (hash-seed dependent, so it differs
    from process to process in the reference itself): the lists of THIS run are recorded, and the run is made under
    PYTHONHASHSEED=0 (main() re-executes the interpreter with it) so that it can be repeated."""
    conf = os.path.join(tmp, "tbpr_ft.conf")
    write_conf(conf, ratings="./dataset/FilmTrust/trainset.txt", social="./dataset/FilmTrust/trust.txt",
               ratings__setup="-columns 0 1 2", social__setup="-columns 0 1 2",
               model__name="TBPR", evaluation__setup="-testSet ./dataset/FilmTrust/testset.txt -b 1",
               item__ranking="on -topN 10,20", num__factors="8", num__max__epoch="3", learnRate="-init 0.03 -max 1",
               reg__lambda="-u 0.01 -i 0.01 -b 0.2 -s 0.2", TBPR="-regT 0.01", output__setup="off -dir ./results/")
    rec = run_numpy_model(conf, 77, "model.ranking.TBPR", "TBPR", "optimization", social=True)
    meta = pack_bpr(rec, "tbpr_filmtrust", keep_full_stream=True)
    m = rec["model"]
    z = dict(np.load(os.path.join(OUT, "tbpr_filmtrust.npz")))
    unknown = {}
    code = lambda name: m.data.user[name] if name in m.data.user else -1 - unknown.setdefault(name, len(unknown))
    z["raw_follower"] = np.array([code(r[0]) for r in rec["raw_relation"]], dtype=np.int64)
    z["raw_followee"] = np.array([code(r[1]) for r in rec["raw_relation"]], dtype=np.int64)
    z["raw_weight"] = np.array([r[2] for r in rec["raw_relation"]], dtype=np.float64)
    for tag, book in (("joint", m.jointSet), ("weak", m.weakSet), ("strong", m.strongSet)):
        ptr, items = [0], []
        for user in m.data.user:                                   # id order
            items += [m.data.item[it] for it in (book[user].keys() if user in book else [])]
            ptr.append(len(items))
        z[tag + "_indptr"] = np.array(ptr, dtype=np.int64); z[tag + "_items"] = np.array(items, dtype=np.int32)
    z["tie_weights"] = np.asarray(m.weights, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "tbpr_filmtrust.npz"), **z)
    meta.update(seed=77, conf=open(conf).read(), theta=float(m.theta), t_s=float(m.t_s), t_w=float(m.t_w), g_theta=float(m.g_theta),
                relations_loaded=len(rec["raw_relation"]), relations_kept=len(m.social.relation),
                python_hash_seed=os.environ.get("PYTHONHASHSEED"))           # the joint-item lists' order depends on it (see main())
    return meta


def case_sbpr_filmtrust(tmp):
    """model/ranking/SBPR.py, numpy path (trainModel :31-78), FilmTrust + trust.txt.  The UNMODIFIED file cannot produce this fixture:
    `Suk = self.FPSet[user][kItems]` (SBPR.py:46) indexes a dict with a list and raises TypeError for the first user who has social
    feedback -- recorded below by running it.  The fixture is the run of the same source with that ONE token replaced (`kItems` ->
    `item_k`: the count the friend-consumed item was drawn with, the statement's evident meaning); everything else -- including the
    negative's rejection test `item_j in self.FPSet`, which looks the ITEM name up among the USER names that are keys of FPSet so far
    -- runs as written.  The per-positive (u, i, k, j, Suk) stream is recovered from the order of `choice` and `sigmoid` calls
    (an accepted draw is a `choice` followed directly by a `sigmoid`)."""
    import importlib
    import traceback
    conf = os.path.join(tmp, "sbpr_ft.conf")
    write_conf(conf, ratings="./dataset/FilmTrust/trainset.txt", social="./dataset/FilmTrust/trust.txt",
               ratings__setup="-columns 0 1 2", social__setup="-columns 0 1 2",
               model__name="SBPR", evaluation__setup="-testSet ./dataset/FilmTrust/testset.txt -b 1",
               item__ranking="on -topN 10,20", num__factors="8", num__max__epoch="3", learnRate="-init 0.03 -max 1",
               reg__lambda="-u 0.01 -i 0.01 -b 0.2 -s 0.2", output__setup="off -dir ./results/")
    # 1. the file as it is
    raised = None
    try:
        run_numpy_model(conf, 91, "model.ranking.SBPR", "SBPR", "none", social=True)
    except TypeError as e:
        tb = traceback.extract_tb(e.__traceback__)[-1]
        raised = dict(type="TypeError", message=str(e), file=os.path.relpath(tb.filename, REF), line=tb.lineno, statement=tb.line)
    assert raised is not None and raised["line"] == 46, raised
    # (run_numpy_model restores its hooks only on success)
    importlib.reload(importlib.import_module("model.ranking.SBPR"))
    import base.recommender as br
    from util.measure import Measure
    br.Measure = Measure
    # 2. the same source, one token replaced
    path = os.path.join(REF, "model", "ranking", "SBPR.py")
    src = open(path).read()
    assert src.count("self.FPSet[user][kItems]") == 1
    mod = types.ModuleType("model.ranking.SBPR_line46")
    exec(compile(src.replace("self.FPSet[user][kItems]", "self.FPSet[user][item_k]"), path + " (line 46: kItems -> item_k)", "exec"), mod.__dict__)
    sys.modules["model.ranking.SBPR_line46"] = mod
    events = []
    ref_choice, ref_sigmoid = mod.choice, mod.sigmoid
    mod.choice = lambda seq: (lambda r: (events.append((len(seq), r)), r)[1])(ref_choice(seq))
    mod.sigmoid = lambda x: (events.append(None), ref_sigmoid(x))[1]
    rec = run_numpy_model(conf, 91, "model.ranking.SBPR_line46", "SBPR", "none", social=True)
    m = rec["model"]
    I = len(m.data.item)
    # accepted draws = a choice directly followed by a sigmoid; a draw from a list shorter than the catalogue is the friend-consumed item
    acc = [events[t] for t in range(len(events) - 1) if events[t] is not None and events[t + 1] is None]
    stream, it = [], iter(acc)
    n_epochs = len(rec["epochs"])
    for _ in range(n_epochs):
        for user in m.PositiveSet:
            u = m.data.user[user]
            for item in m.PositiveSet[user]:
                ln, name = next(it)
                if len(m.FPSet[user]) > 0:
                    assert ln == len(m.FPSet[user]) and name in m.FPSet[user]
                    k, w = m.data.item[name], m.FPSet[user][name]
                    ln, name = next(it)
                else:
                    k, w = -1, 0
                assert ln == I
                stream.append((u, m.data.item[item], k, m.data.item[name], w))
    assert next(it, None) is None
    rec["steps"] = [(a, b, d) for a, b, c, d, e in stream]
    meta = pack_bpr(rec, "sbpr_filmtrust", keep_full_stream=False)
    z = dict(np.load(os.path.join(OUT, "sbpr_filmtrust.npz")))
    z.pop("steps_head", None); z.pop("steps_tail", None)
    z["stream"] = np.array(stream, dtype=np.int32)                     # (u, i, k or -1, j, Suk) per positive and epoch
    z["b"] = np.asarray(m.b, dtype=np.float64)
    unknown = {}
    code = lambda name: m.data.user[name] if name in m.data.user else -1 - unknown.setdefault(name, len(unknown))
    z["raw_follower"] = np.array([code(r[0]) for r in rec["raw_relation"]], dtype=np.int64)
    z["raw_followee"] = np.array([code(r[1]) for r in rec["raw_relation"]], dtype=np.int64)
    z["raw_weight"] = np.array([r[2] for r in rec["raw_relation"]], dtype=np.float64)
    ptr, items, cnt = [0], [], []
    for user in m.data.user:                                           # id order
        book = m.FPSet[user] if user in m.FPSet else {}
        items += [m.data.item[x] for x in book]; cnt += list(book.values())
        ptr.append(len(items))
    z["fp_indptr"] = np.array(ptr, dtype=np.int64); z["fp_items"] = np.array(items, dtype=np.int32); z["fp_counts"] = np.array(cnt, dtype=np.int32)
    z["positive_set_users"] = np.array([m.data.user[x] for x in m.PositiveSet], dtype=np.int32)
    z["user_names"] = np.array(list(m.data.user.keys())); z["item_names"] = np.array(list(m.data.item.keys()))      # id order
    np.savez_compressed(os.path.join(OUT, "sbpr_filmtrust.npz"), **z)
    meta.update(seed=91, conf=open(conf).read(), unmodified_reference_raises=raised,
                fixture_source="model/ranking/SBPR.py with line 46 `self.FPSet[user][kItems]` -> `self.FPSet[user][item_k]`; nothing else changed",
                users_with_social_feedback=int(sum(1 for x in m.PositiveSet if len(m.FPSet[x]) > 0)), stream_sha256=sha(z["stream"]),
                relations_loaded=len(rec["raw_relation"]), relations_kept=len(m.social.relation),
                python_hash_seed=os.environ.get("PYTHONHASHSEED"))           # the joint-item lists' order depends on it (see main())
    return meta


def case_pairwise_and_adj(tmp):
    """base/deepRecommender.py:29-52 sampler and base/graphRecommender.py:10-29 adjacency,
    both pure python/scipy -> executable with the tensorflow stub."""
    from QRec import QRec
    from util.config import ModelConf
    from model.ranking.LightGCN import LightGCN
    conf = os.path.join(tmp, "lgcn.conf")
    write_conf(conf, ratings="./dataset/FilmTrust/trainset.txt", ratings__setup="-columns 0 1 2",
               model__name="LightGCN", evaluation__setup="-testSet ./dataset/FilmTrust/testset.txt -b 1",
               item__ranking="on -topN 20", num__factors="8", num__max__epoch="1", batch_size="2000",
               learnRate="-init 0.001 -max 1", LightGCN="-n_layer 2",
               reg__lambda="-u 0.001 -i 0.001 -b 0.2 -s 0.2", output__setup="off -dir ./results/")
    random.seed(3); np.random.seed(3)
    with redirect_stdout(io.StringIO()):
        q = QRec(ModelConf(conf))
        m = LightGCN(q.config, q.trainingData, q.testData)
        m.readConfiguration()
    train_uid = np.array([m.data.user[r[0]] for r in m.data.trainingData], dtype=np.int32)
    train_iid = np.array([m.data.item[r[1]] for r in m.data.trainingData], dtype=np.int32)
    batches = []
    for ep in range(2):
        for b in m.next_batch_pairwise():
            batches.append(np.array(b, dtype=np.int32).T)  # [B,3]
    stream = np.concatenate(batches)
    adj = m.create_joint_sparse_adjaceny().tocsr()
    adj.sort_indices()
    row, col = m.create_joint_sparse_adjaceny().nonzero()
    np.savez_compressed(os.path.join(OUT, "pairwise_adj_filmtrust.npz"),
                        train_uid=train_uid, train_iid=train_iid, stream=stream,
                        batch_sizes=np.array([b.shape[0] for b in batches], dtype=np.int32),
                        adj_indptr=adj.indptr.astype(np.int64), adj_indices=adj.indices.astype(np.int32),
                        adj_data=adj.data.astype(np.float32), nz_row=row.astype(np.int32),
                        nz_col=col.astype(np.int32),
                        py_state=np.array(random.getstate()[1], dtype=np.uint32))
    return dict(name="pairwise_adj_filmtrust", seed=3, conf=open(conf).read(),
                n_users=len(m.data.user), n_items=len(m.data.item), n_train=len(m.data.trainingData),
                batch_size=m.batch_size, epochs_sampled=2, stream_sha256=sha(stream),
                adj_dtype=str(adj.dtype), adj_nnz=int(adj.nnz))


def case_pointwise(tmp):
    """base/deepRecommender.py:54-77, the other sampler of the same base class: no shuffle, per training row the positive
    (label 1) followed by four randint negatives (label 0), redrawn while rated.  Two passes of the generator, the
    interpreter's generator state after them."""
    from QRec import QRec
    from util.config import ModelConf
    from model.ranking.LightGCN import LightGCN
    conf = os.path.join(tmp, "lgcn_pw.conf")
    write_conf(conf, ratings="./dataset/FilmTrust/trainset.txt", ratings__setup="-columns 0 1 2",
               model__name="LightGCN", evaluation__setup="-testSet ./dataset/FilmTrust/testset.txt -b 1",
               item__ranking="on -topN 20", num__factors="8", num__max__epoch="1", batch_size="1500",
               learnRate="-init 0.001 -max 1", LightGCN="-n_layer 2",
               reg__lambda="-u 0.001 -i 0.001 -b 0.2 -s 0.2", output__setup="off -dir ./results/")
    random.seed(11); np.random.seed(11)
    with redirect_stdout(io.StringIO()):
        q = QRec(ModelConf(conf))
        m = LightGCN(q.config, q.trainingData, q.testData)
        m.readConfiguration()
    train_uid = np.array([m.data.user[r[0]] for r in m.data.trainingData], dtype=np.int32)
    train_iid = np.array([m.data.item[r[1]] for r in m.data.trainingData], dtype=np.int32)
    batches = []
    for ep in range(2):
        for b in m.next_batch_pointwise():
            batches.append(np.array(b, dtype=np.int32).T)  # [5B,3]: u, i, y
    stream = np.concatenate(batches)
    np.savez_compressed(os.path.join(OUT, "pointwise_filmtrust.npz"), train_uid=train_uid, train_iid=train_iid, stream=stream,
                        batch_sizes=np.array([b.shape[0] for b in batches], dtype=np.int32),
                        py_state=np.array(random.getstate()[1], dtype=np.uint32))
    return dict(name="pointwise_filmtrust", seed=11, conf=open(conf).read(), n_users=len(m.data.user), n_items=len(m.data.item),
                n_train=len(m.data.trainingData), batch_size=m.batch_size, epochs_sampled=2, stream_sha256=sha(stream))


def case_sgl_subgraph(tmp):
    """model/ranking/SGL.py:113-155 _create_adj_mat(is_subgraph=True): node dropout (aug 0) and edge
    dropout (aug 1) sub-adjacencies, drawn with random.sample from the seeded CPython generator."""
    from QRec import QRec
    from util.config import ModelConf
    from model.ranking.SGL import SGL
    conf = os.path.join(tmp, "sgl.conf")
    write_conf(conf, ratings="./dataset/FilmTrust/trainset.txt", ratings__setup="-columns 0 1 2",
               model__name="SGL", evaluation__setup="-testSet ./dataset/FilmTrust/testset.txt -b 1",
               item__ranking="on -topN 20", num__factors="8", num__max__epoch="1", batch_size="2048",
               learnRate="-init 0.001 -max 1", SGL="-n_layer 2 -lambda 0.1 -droprate 0.1 -augtype 1 -temp 0.2",
               reg__lambda="-u 0.001 -i 0.001 -b 0.2 -s 0.2", output__setup="off -dir ./results/")
    random.seed(11); np.random.seed(11)
    with redirect_stdout(io.StringIO()):
        q = QRec(ModelConf(conf))
        m = SGL(q.config, q.trainingData, q.testData)
        m.readConfiguration()
    arrays = dict(train_uid=np.array([m.data.user[r[0]] for r in m.data.trainingData], dtype=np.int32),
                  train_iid=np.array([m.data.item[r[1]] for r in m.data.trainingData], dtype=np.int32))
    for tag, aug in (("node", 0), ("edge", 1), ("edge2", 1)):
        arrays[f"state_before_{tag}"] = np.array(random.getstate()[1], dtype=np.uint32)
        A = m._create_adj_mat(is_subgraph=True, aug_type=aug).tocsr(); A.sort_indices()
        arrays[f"{tag}_indptr"] = A.indptr.astype(np.int64); arrays[f"{tag}_indices"] = A.indices.astype(np.int32)
        arrays[f"{tag}_data"] = A.data.astype(np.float32)
    arrays["state_after"] = np.array(random.getstate()[1], dtype=np.uint32)
    np.savez_compressed(os.path.join(OUT, "sgl_subgraph_filmtrust.npz"), **arrays)
    return dict(name="sgl_subgraph_filmtrust", seed=11, conf=open(conf).read(), n_users=len(m.data.user),
                n_items=len(m.data.item), n_train=len(m.data.trainingData), drop_rate=m.drop_rate)


def case_sept_graphs(tmp):
    """model/ranking/SEPT.py:32-114 graph builders (pure scipy + random.sample) on FilmTrust + trust.txt, and what
    SocialRecommender.__init__ (base/socialRecommender.py:6-41) keeps of the relation list."""
    from QRec import QRec
    from util.config import ModelConf
    from model.ranking.SEPT import SEPT
    conf = os.path.join(tmp, "sept.conf")
    write_conf(conf, ratings="./dataset/FilmTrust/trainset.txt", social="./dataset/FilmTrust/trust.txt",
               ratings__setup="-columns 0 1 2", social__setup="-columns 0 1 2",
               model__name="SEPT", evaluation__setup="-testSet ./dataset/FilmTrust/testset.txt -b 1",
               item__ranking="on -topN 10", num__factors="8", num__max__epoch="3", batch_size="2000",
               learnRate="-init 0.001 -max 1", SEPT="-n_layer 2 -ss_rate 0.005 -drop_rate 0.3 -ins_cnt 10",
               reg__lambda="-u 0.001 -i 0.01 -b 0.2 -s 0.2", output__setup="off -dir ./results/")
    random.seed(13); np.random.seed(13)
    with redirect_stdout(io.StringIO()):
        q = QRec(ModelConf(conf))
        n_loaded = len(q.relation)
        raw = [list(r) for r in q.relation]                      # before SocialRecommender.__init__ prunes the list in place
        m = SEPT(q.config, q.trainingData, q.testData, q.relation)
        m.readConfiguration()
        m.num_users, m.num_items, m.train_size = m.data.trainingSize()
    def csr(prefix, A, arrays):
        A = A.tocsr(); A.sort_indices()
        arrays[prefix + "_indptr"] = A.indptr.astype(np.int64); arrays[prefix + "_indices"] = A.indices.astype(np.int32)
        arrays[prefix + "_data"] = A.data.astype(np.float64)
    arrays = dict(train_uid=np.array([m.data.user[r[0]] for r in m.data.trainingData], dtype=np.int32),
                  train_iid=np.array([m.data.item[r[1]] for r in m.data.trainingData], dtype=np.int32),
                  follower=np.array([m.data.user[r[0]] for r in m.social.relation], dtype=np.int32),
                  followee=np.array([m.data.user[r[1]] for r in m.social.relation], dtype=np.int32))
    unknown = {}
    code = lambda name: m.data.user[name] if name in m.data.user else -1 - unknown.setdefault(name, len(unknown))
    arrays["raw_follower"] = np.array([code(r[0]) for r in raw], dtype=np.int64)     # >= 0: training-user id; < 0: a user
    arrays["raw_followee"] = np.array([code(r[1]) for r in raw], dtype=np.int64)     # that never occurs in the training data
    arrays["raw_weight"] = np.array([r[2] for r in raw], dtype=np.float64)
    bs = m.get_birectional_social_matrix()
    social, sharing = m.get_social_related_views(bs, m.buildSparseRatingMatrix())
    csr("social", social, arrays); csr("sharing", sharing, arrays)
    csr("full", m.get_adj_mat(), arrays)
    for tag in ("sub1", "sub2"):
        arrays[f"state_before_{tag}"] = np.array(random.getstate()[1], dtype=np.uint32)
        csr(tag, m.get_adj_mat(is_subgraph=True), arrays)
    arrays["state_after"] = np.array(random.getstate()[1], dtype=np.uint32)
    np.savez_compressed(os.path.join(OUT, "sept_graphs_filmtrust.npz"), **arrays)
    return dict(name="sept_graphs_filmtrust", seed=13, conf=open(conf).read(), n_users=len(m.data.user), n_items=len(m.data.item),
                n_train=len(m.data.trainingData), relations_loaded=n_loaded, relations_kept=len(m.social.relation),
                social_users=len(m.social.user), drop_rate=m.drop_rate, regS=m.regS)


def case_mhcn_graphs(tmp):
    """model/ranking/MHCN.py:26-85 (pure scipy): the three motif-induced hypergraph adjacencies H_s, H_j, H_p, and the
    value list of buildJointAdjacency (:46-52; tf.SparseTensor is replaced by a recorder for this one call)."""
    import tensorflow as tf
    from QRec import QRec
    from util.config import ModelConf
    from model.ranking.MHCN import MHCN
    conf = os.path.join(tmp, "mhcn.conf")
    write_conf(conf, ratings="./dataset/FilmTrust/trainset.txt", social="./dataset/FilmTrust/trust.txt",
               ratings__setup="-columns 0 1 2", social__setup="-columns 0 1 2",
               model__name="MHCN", evaluation__setup="-testSet ./dataset/FilmTrust/testset.txt -b 1",
               item__ranking="on -topN 10", num__factors="8", num__max__epoch="2", batch_size="2000",
               learnRate="-init 0.001 -max 1", MHCN="-n_layer 2 -ss_rate 0.01",
               reg__lambda="-u 0.001 -i 0.01 -b 0.2 -s 0.2", output__setup="off -dir ./results/")
    random.seed(17); np.random.seed(17)
    with redirect_stdout(io.StringIO()):
        q = QRec(ModelConf(conf))
        m = MHCN(q.config, q.trainingData, q.testData, q.relation)
        m.readConfiguration()
        m.num_users, m.num_items, m.train_size = m.data.trainingSize()
    arrays = dict(train_uid=np.array([m.data.user[r[0]] for r in m.data.trainingData], dtype=np.int32),
                  train_iid=np.array([m.data.item[r[1]] for r in m.data.trainingData], dtype=np.int32),
                  train_r=np.array([r[2] for r in m.data.trainingData], dtype=np.float64),
                  follower=np.array([m.data.user[r[0]] for r in m.social.relation], dtype=np.int32),
                  followee=np.array([m.data.user[r[1]] for r in m.social.relation], dtype=np.int32))
    with np.errstate(divide="ignore"):
        H = m.buildMotifInducedAdjacencyMatrix()
    for tag, A in zip(("Hs", "Hj", "Hp"), H):
        A = A.tocsr(); A.sort_indices()
        assert A.data.dtype == np.float32 and A.shape[1] < 32768
        arrays[tag + "_indptr"] = A.indptr.astype(np.int64); arrays[tag + "_indices"] = A.indices.astype(np.int16)
        arrays[tag + "_data"] = A.data                                # float32, as the reference holds them
    tf.SparseTensor = lambda indices, values, dense_shape: (indices, values, dense_shape)
    idx, vals, shape = m.buildJointAdjacency()
    del tf.SparseTensor
    arrays["R_indices"] = np.array(idx, dtype=np.int32); arrays["R_values"] = np.array(vals, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "mhcn_graphs_filmtrust.npz"), **arrays)
    return dict(name="mhcn_graphs_filmtrust", seed=17, conf=open(conf).read(), n_users=len(m.data.user), n_items=len(m.data.item),
                n_train=len(m.data.trainingData), relations_kept=len(m.social.relation), R_shape=[int(x) for x in shape],
                nnz=[int(A.nnz) for A in H])


def loader_inputs():
    """the rating files (as text) and load options the loader fixture covers: util/io.py:31-76's splitting rule (every
    single delimiter character splits, so two blanks make an empty field), -columns orders, -header, -delim, -b
    thresholds, train / test flag, universal newlines, float literals, empty file"""
    rng = np.random.default_rng(0)
    lits = ["1", "2.5", "4", "0.5", "5.0", "1e0", ".5", "+3", "3.", "-1"]
    big = "\n".join(f"u{u} i{i} {r}" for u, i, r in zip(rng.integers(0, 50, 400), rng.integers(0, 70, 400), rng.choice(lits, 400))) + "\n"
    return [
        dict(name="plain", text=big, setup="-columns 0 1 2"),
        dict(name="test_flag", text=big, setup="-columns 0 1 2", bTest=True),
        dict(name="binarized_2.5", text=big, setup="-columns 0 1 2", binarized=True, threshold=2.5),
        dict(name="binarized_default_threshold", text=big, setup="-columns 0 1 2", binarized=True),
        dict(name="binarized_test", text=big, setup="-columns 0 1 2", binarized=True, threshold=1.0, bTest=True),
        dict(name="mixed_delims_crlf_cr_no_final_newline_swapped_columns", text="a,b\t3\r\nc d,4.5  \r\n  e\tf 2\rg,h,1", setup="-columns 1 0 2"),
        dict(name="two_blanks_make_an_empty_field", text="u1 i1 3 9\nu2  i2 4\n", setup="-columns 0 1 3"),
        dict(name="header_no_rating_column", text="user item\nu1 i1\nu2 i1\n", setup="-columns 0 1 -header"),
        dict(name="header_with_ratings", text="user,item,rating\nu1,i1,3\nu2,i1,0.5\n", setup="-columns 0 1 2 -header"),
        dict(name="custom_delim", text="u1;i1;3\nu2;i2;4\n", setup="-columns 0 1 2 -delim ;"),
        dict(name="rating_in_column_3_of_5", text="7 u1 i1 3.5 x\n8 u2 i2 1 y\n", setup="-columns 1 2 3"),
        dict(name="tabs_only", text="u1\ti1\t2\nu1\ti2\t5\n", setup="-columns 0 1 2"),
        dict(name="repeated_pairs_keep_every_row", text="u1 i1 1\nu1 i1 4\nu1 i1 1\n", setup="-columns 0 1 2"),
        dict(name="empty_file", text="", setup="-columns 0 1 2"),
    ]


def case_loader(tmp):
    """FileIO.loadDataSet (util/io.py:31-76) of the unmodified reference on the files above: the rows it returns"""
    from util.io import FileIO
    cases = []
    for c in loader_inputs():
        path = os.path.join(tmp, "loader_case.txt")
        with open(path, "w", newline="") as f:
            f.write(c["text"])
        kw = {k: c[k] for k in ("bTest", "binarized", "threshold") if k in c}
        with redirect_stdout(io.StringIO()):
            rows = FileIO.loadDataSet({"ratings.setup": c["setup"]}, path, **kw)
        cases.append(dict(c, rows=rows))
    with open(os.path.join(OUT, "loader_cases.json"), "w") as f:
        json.dump(cases, f, indent=0)
    return dict(name="loader", n_cases=len(cases), n_rows=sum(len(c["rows"]) for c in cases))


def main():
    # TBPR.py:122 orders every user's joint items by iterating a Python SET OF STRINGS: the reference's own run depends on the
    # interpreter's string-hash seed.  The fixtures are recorded under PYTHONHASHSEED=0 (written into golden_meta.json), so that an
    # unchanged reference regenerates byte-identical files.
    if os.environ.get("PYTHONHASHSEED") != "0":
        os.environ["PYTHONHASHSEED"] = "0"
        os.execv(sys.executable, [sys.executable] + sys.argv)
    install_stubs()
    tmp = tempfile.mkdtemp(prefix="qrec_golden_")
    os.symlink(os.path.join(REF, "dataset"), os.path.join(tmp, "dataset"))
    os.chdir(tmp)
    only = sys.argv[1:]
    cases = [case_bpr_filmtrust, case_bpr_lastfm, case_basicmf, case_pmf, case_svd, case_ee, case_svdpp, case_pairwise_and_adj, case_pointwise, case_sgl_subgraph, case_sept_graphs, case_tbpr_filmtrust, case_sbpr_filmtrust, case_mhcn_graphs, case_loader]
    if only:   # regenerate a subset, keep the other entries of golden_meta.json
        cases = [c for c in cases if c.__name__ in only]
        old = json.load(open(os.path.join(OUT, "golden_meta.json"))) if os.path.exists(os.path.join(OUT, "golden_meta.json")) else {}
        metas = [c(tmp) for c in cases]
        old.update({m["name"]: m for m in metas})
        with open(os.path.join(OUT, "golden_meta.json"), "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
        for m in metas:
            print(m["name"], "ok")
        return
    metas = [c(tmp) for c in cases]
    with open(os.path.join(OUT, "golden_meta.json"), "w") as f:
        json.dump({m["name"]: m for m in metas}, f, indent=1, sort_keys=True)
    for m in metas:
        print(m["name"], "ok", {k: m[k] for k in ("n_users", "n_items", "n_train") if k in m})


if __name__ == "__main__":
    main()
