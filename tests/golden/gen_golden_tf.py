#!/usr/bin/env python3
"""Golden vectors for the reference's TensorFlow-path models, produced by running the reference's OWN model classes
(model/ranking/{BPR,LightGCN,NGCF,SimGCL}.py, base/{deepRecommender,graphRecommender}.py, util/loss.py) unmodified, with
``tests/golden/tf1shim.py`` standing in for the ``tensorflow`` module (TF 1.14 is not installable here; see the shim's
header for what it restates -- primitive op semantics -- and what it does not -- the models).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_tf.py

Needs /root/reference (build container only).  Writes tf_<model>_filmtrust.npz + one entry per model in golden_tf.json
next to this file; tests/test_oracle_tf_golden.py checks oracle/tfmodels.py against them.

Per case: a FilmTrust subset (the first 300 users of dataset/FilmTrust/ratings.txt, written to a scratch file, split by the
reference's own ``-ap 0.2``), a handful of training steps at a small embedding size.  Captured: the id-mapped training
pairs, every variable's initial and final value, each step's (u, i, j) batch as the reference's sampler drew it, the losses
the reference printed, the Session.run index of each step (the key to regenerate the step's random draws:
``tf1shim.random_uniform(seed, run_index, op_index, shape)``), the embeddings the reference scores with, and its measures.
"""
import io
import json
import os
import random
import sys
import tempfile
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G          # noqa: E402  (install_stubs, write_conf)
import tf1shim                  # noqa: E402

REF = G.REF
N_SUBSET_USERS = 300


def make_subset(tmp):
    """the rows of the first N_SUBSET_USERS distinct users of FilmTrust's ratings file, order kept"""
    src = os.path.join(REF, "dataset", "FilmTrust", "ratings.txt")
    users, rows = {}, []
    with open(src) as f:
        for line in f:
            u = line.split()[0]
            if u not in users:
                if len(users) == N_SUBSET_USERS:
                    continue
                users[u] = 1
            rows.append(line)
    os.makedirs(os.path.join(tmp, "sub"), exist_ok=True)
    path = os.path.join(tmp, "sub", "ratings.txt")
    with open(path, "w") as f:
        f.writelines(rows)
    return "./sub/ratings.txt", len(rows)


def run_tf_model(conf_path, seed, module, cls_name, after=None, social=False):
    """QRec(conf) -> model.execute() with the shim as tensorflow; every Session.run that carries a train op is logged"""
    import importlib
    from QRec import QRec
    from util.config import ModelConf
    tf1shim.reset(seed)
    mod = importlib.import_module(module)
    cls = getattr(mod, cls_name)
    steps = []
    self_model = []
    seen_ops, first_steps = [], []
    orig_run = tf1shim.Session.run

    def run(self, fetches, feed_dict=None, **kw):
        idx = tf1shim.STATE.run_index
        fl = fetches if isinstance(fetches, (list, tuple)) else [fetches]
        ops = [t for t in fl if isinstance(t, tf1shim._TrainOp)]
        # the first step of every train op (SEPT has two: the recommendation task alone, then the joint objective): the variables the
        # step started from and the gradients its minimize() applied -- the pre-Adam comparison point of the HIP trainers
        first = bool(ops) and id(ops[0]) not in seen_ops
        before = {v.index: v.value.detach().numpy().copy() for v in tf1shim.all_variables()} if first else None
        out = orig_run(self, fetches, feed_dict, **kw)
        if ops:
            if first:
                seen_ops.append(id(ops[0]))
                first_steps.append(dict(step=len(steps), before=before, grads={k: g.copy() for k, g in ops[0].opt.last_grads_by_index.items()}))
            feeds = {getattr(k, "name", None): np.asarray(v) for k, v in (feed_dict or {}).items()}
            labels = {id(t): key for key, t in getattr(self_model[0], "sub_mat", {}).items()} if self_model else {}
            feeds_all = {labels.get(id(k), getattr(k, "name", None)): v for k, v in (feed_dict or {}).items()}
            steps.append(dict(run_index=idx, feeds=feeds, feeds_all=feeds_all, out=[o for o in out if o is not None],
                              random=list(tf1shim.STATE.run_log[-1][1]), signs=tf1shim.STATE.sign_log.pop(idx, {})))
        tf1shim.STATE.sign_log.pop(idx, None)
        return out
    tf1shim.Session.run = run
    rec = {}
    orig_init = cls.initModel

    def initModel(self):
        orig_init(self)
        self_model.append(self)
        rec["order0"] = [(self.data.user[a], self.data.item[b]) for a, b, _ in self.data.trainingData]
        rec["rating0"] = [float(r) for _, _, r in self.data.trainingData]
    cls.initModel = initModel
    random.seed(seed); np.random.seed(seed)
    buf = io.StringIO()
    try:
        with redirect_stdout(buf):
            q = QRec(ModelConf(conf_path))
            m = cls(q.config, q.trainingData, q.testData, q.relation) if social else cls(q.config, q.trainingData, q.testData)
            measure = m.execute()
            extra = after(m) if after else {}
    finally:
        tf1shim.Session.run = orig_run
        cls.initModel = orig_init
    rec.update(model=m, measure=measure, steps=steps, extra=extra, stdout=buf.getvalue(), first_steps=first_steps)
    return rec


def pack(rec, name, var_names, conf_text, seed, params):
    m = rec["model"]
    order0 = np.array(rec["order0"], dtype=np.int32)
    arrays = dict(train_uid=order0[:, 0], train_iid=order0[:, 1])
    vs = {v.name: v for v in tf1shim.all_variables()}
    used = []
    for vn in var_names:
        # a name may have been given to several variables (iterativeRecommender.py:47-48 names three 'U'): the LAST one with the
        # requested name that received gradients is the model's
        cands = [v for v in tf1shim.all_variables() if v.name == vn]
        v = cands[0] if len(cands) == 1 else [c for c in cands if not np.array_equal(c.initial, c.value.detach().numpy())][0]
        arrays[f"init_{vn}"] = v.initial.astype(np.float32)
        arrays[f"final_{vn}"] = v.value.detach().numpy().astype(np.float32)
        used.append(dict(name=vn, index=v.index, init=list(v.init_spec[:1]) + [list(v.init_spec[1]), v.init_spec[2]]))
        # grad<k>_<var>: the gradient the k-th train op's FIRST step applied to this variable (absent: the loss of that op does not
        # reach it); pre<k>_<var>: the variable's value when that step started (k = 0: the initial value, not stored twice)
        for k, fs in enumerate(rec["first_steps"]):
            if v.index in fs["grads"]:
                arrays[f"grad{k}_{vn}"] = fs["grads"][v.index].astype(np.float32)
            if k > 0:
                arrays[f"pre{k}_{vn}"] = fs["before"][v.index].astype(np.float32)
    del vs
    u = [s["feeds"]["u_idx"].astype(np.int32) for s in rec["steps"]]
    arrays["batch_offsets"] = np.concatenate([[0], np.cumsum([x.size for x in u])]).astype(np.int64)
    arrays["batch_u"] = np.concatenate(u)
    arrays["batch_i"] = np.concatenate([s["feeds"]["v_idx"].astype(np.int32) for s in rec["steps"]])
    if "neg_holder" in rec["steps"][0]["feeds"]:
        arrays["batch_j"] = np.concatenate([s["feeds"]["neg_holder"].astype(np.int32) for s in rec["steps"]])
    width = max(len(s["out"]) for s in rec["steps"])
    arrays["losses"] = np.array([[float(x) for x in s["out"]] + [np.nan] * (width - len(s["out"])) for s in rec["steps"]], dtype=np.float64)
    arrays["run_index"] = np.array([s["run_index"] for s in rec["steps"]], dtype=np.int64)
    for k, v in rec["extra"].items():
        arrays[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    return dict(name=name, seed=seed, shim_dtype=str(tf1shim.DT), n_users=len(m.data.user), n_items=len(m.data.item), n_train=int(order0.shape[0]),
                n_steps=len(rec["steps"]), first_steps=[fs["step"] for fs in rec["first_steps"]], emb_size=m.emb_size, lr=m.lRate, regU=m.regU, batch_size=m.batch_size,
                variables=used, random_ops=[[list(map(lambda t: list(t) if isinstance(t, tuple) else t, r)) for r in s["random"]] for s in rec["steps"][:1]],
                measure=rec["measure"], conf=conf_text, **params)


def base_conf(tmp, ratings, **kv):
    conf = os.path.join(tmp, kv["model__name"] + "_tf.conf")
    d = dict(ratings=ratings, ratings__setup="-columns 0 1 2", evaluation__setup="-ap 0.2 -b 1", item__ranking="on -topN 10",
             num__factors="8", num__max__epoch="2", batch_size="1000", learnRate="-init 0.01 -max 1",
             reg__lambda="-u 0.01 -i 0.01 -b 0.2 -s 0.2", output__setup="off -dir ./results/")
    d.update(kv)
    G.write_conf(conf, **d)
    return conf


def case_lightgcn(tmp, ratings):
    conf = base_conf(tmp, ratings, model__name="LightGCN", LightGCN="-n_layer 2")
    rec = run_tf_model(conf, 101, "model.ranking.LightGCN", "LightGCN",
                       after=lambda m: dict(score_U=np.asarray(m.U, np.float32), score_V=np.asarray(m.V, np.float32)))
    return pack(rec, "tf_lightgcn_filmtrust", ["U", "V"], open(conf).read(), 101, dict(n_layers=2))


def case_bpr_tf(tmp, ratings):
    conf = base_conf(tmp, ratings, model__name="BPR", evaluation__setup="-ap 0.2 -b 1 -tf")
    rec = run_tf_model(conf, 102, "model.ranking.BPR", "BPR",
                       after=lambda m: dict(score_U=np.asarray(m.P, np.float32), score_V=np.asarray(m.Q, np.float32)))
    return pack(rec, "tf_bpr_filmtrust", ["U", "V"], open(conf).read(), 102, {})


def case_ngcf(tmp, ratings):
    conf = base_conf(tmp, ratings, model__name="NGCF")

    def after(m):        # the tables the reference scores with: the inference graph (isTraining = 0, NGCF.py:65-69)
        U, V = m.sess.run([m.multi_user_embeddings, m.multi_item_embeddings], feed_dict={m.isTraining: 0})
        return dict(score_U=U.astype(np.float32), score_V=V.astype(np.float32))
    rec = run_tf_model(conf, 103, "model.ranking.NGCF", "NGCF", after=after)
    return pack(rec, "tf_ngcf_filmtrust", ["U", "V", "W_0_1", "W_0_2", "W_1_1", "W_1_2"], open(conf).read(), 103, dict(keep_prob=0.9))


def case_simgcl(tmp, ratings):
    conf = base_conf(tmp, ratings, model__name="SimGCL", SimGCL="-n_layer 2 -lambda 0.5 -eps 0.1")

    def after(m):
        U, V = m.sess.run([m.main_user_embeddings, m.main_item_embeddings])
        return dict(score_U=U.astype(np.float32), score_V=V.astype(np.float32), best_U=np.asarray(m.U, np.float32), best_V=np.asarray(m.V, np.float32))
    rec = run_tf_model(conf, 104, "model.ranking.SimGCL", "SimGCL", after=after)
    # the sign pattern every perturbation of every training step used (SimGCL.py:35 `tf.sign(emb)`), sign ops in creation order =
    # view 1 layer 1, layer 2, view 2 layer 1, layer 2 (the random ops' order): bit planes "negative" and "exactly zero" --
    # sign() is discontinuous, and a trainer that is to follow this run past an entry within rounding of zero needs the pattern
    ops = sorted(rec["steps"][0]["signs"])
    assert len(ops) == 4 and all(sorted(s["signs"]) == ops for s in rec["steps"]), ops
    sg = np.stack([np.stack([s["signs"][o] for o in ops]) for s in rec["steps"]])          # [steps, 4, N, d] int8
    rec["extra"]["sign_neg_bits"] = np.packbits(sg < 0, axis=None)
    rec["extra"]["sign_zero_bits"] = np.packbits(sg == 0, axis=None)
    rec["extra"]["sign_shape"] = np.array(sg.shape, dtype=np.int64)
    # SimGCL.initModel replaces the base class's U / V variables by unnamed Xavier ones (SimGCL.py:42-44): the shim names
    # unnamed variables Variable_<index>; the two that moved are the model's
    moved = [v.name for v in tf1shim.all_variables() if not np.array_equal(v.initial, v.value.detach().numpy())]
    assert len(moved) == 2, moved
    return pack(rec, "tf_simgcl_filmtrust", moved, open(conf).read(), 104, dict(n_layers=2, cl_rate=0.5, eps=0.1, var_roles=dict(zip(moved, ["U", "V"]))))


def _case_sgl(tmp, ratings, aug, name, seed):
    """-augtype 1 (edge dropout): two sub-graphs per epoch; 2 (random walk): two per layer and epoch; 0 (node dropout): two per
    epoch, each from a dropped-user and a dropped-item list -- all drawn with random.sample (SGL.py:118-140).  The drawn lists
    are recorded (the test rebuilds the normalised sub-adjacencies from them and checks them against the hash of what was fed)."""
    conf = base_conf(tmp, ratings, model__name="SGL", SGL=f"-n_layer 2 -lambda 0.1 -droprate 0.1 -augtype {aug} -temp 0.2")
    from model.ranking.SGL import SGL
    kept, orders, restore = record_subgraph_draws(SGL, "_create_adj_mat")
    try:
        def after(m):
            U, V = m.sess.run([m.main_user_embeddings, m.main_item_embeddings])
            return dict(score_U=U.astype(np.float32), score_V=V.astype(np.float32))
        with np.errstate(divide="ignore"):
            rec = run_tf_model(conf, seed, "model.ranking.SGL", "SGL", after=after)
    finally:
        restore()
    pos = {p: k for k, p in enumerate(rec["order0"])}
    for k, o in enumerate(orders):       # the training list is shuffled in place by the sampler: as a permutation of the initial order
        rec["extra"][f"order_{k}"] = np.array([pos[p] for p in o], dtype=np.int32)
    rec["extra"].update({f"keep_{k}": v for k, v in enumerate(kept)})
    keys = [f"adj_{w}_sub{v}" for v in (1, 2) for w in ("indices", "values")] if aug in (0, 1) else \
           [f"adj_{w}_sub{v}{k}" for k in range(2) for v in (1, 2) for w in ("indices", "values")]
    fed = [[G.sha(np.asarray(s["feeds_all"][k])) for k in keys] for s in rec["steps"]]
    meta = pack(rec, name, ["U", "V"], open(conf).read(), seed, dict(n_layers=2, ssl_reg=0.1, temp=0.2, drop_rate=0.1, aug_type=aug))
    meta["n_keep_lists"] = len(kept); meta["n_subgraphs"] = len(orders)
    meta["fed_sha256"] = fed; meta["fed_keys"] = keys
    return meta


def case_sgl(tmp, ratings):
    return _case_sgl(tmp, ratings, 1, "tf_sgl_filmtrust", 105)


def case_sgl_random_walk(tmp, ratings):
    return _case_sgl(tmp, ratings, 2, "tf_sgl_rw_filmtrust", 109)


def case_sgl_node_dropout(tmp, ratings):
    return _case_sgl(tmp, ratings, 0, "tf_sgl_nd_filmtrust", 110)


def record_subgraph_draws(cls, method):
    """wrap random.sample and cls.<method>(is_subgraph=True): the kept-edge lists and the order of the (in-place shuffled) training
    list at every draw; returns (kept, orders, restore)"""
    kept, orders = [], []
    orig_sample, orig_m = random.sample, getattr(cls, method)

    def sample(population, k, **kw):
        r = orig_sample(population, k, **kw)
        kept.append(np.array(r, dtype=np.int32))
        return r

    def wrapped(self, is_subgraph=False, *a, **kw):
        if is_subgraph:
            orders.append([(self.data.user[x], self.data.item[y]) for x, y, _ in self.data.trainingData])
        return orig_m(self, is_subgraph, *a, **kw)
    random.sample = sample
    setattr(cls, method, wrapped)

    def restore():
        random.sample = orig_sample
        setattr(cls, method, orig_m)
    return kept, orders, restore


def case_buir(tmp, ratings):
    conf = base_conf(tmp, ratings, model__name="BUIR", BUIR="-n_layer 2 -tau 0.995 -drop_rate 0.2")
    from model.ranking.BUIR import BUIR
    kept, orders, restore = record_subgraph_draws(BUIR, "get_adj_mat")
    try:
        def after(m):
            return dict(q_user=m.q_user.astype(np.float32), q_item=m.q_item.astype(np.float32), o_user=m.o_user.astype(np.float32), o_item=m.o_item.astype(np.float32))
        rec = run_tf_model(conf, 106, "model.ranking.BUIR", "BUIR", after=after)
    finally:
        restore()
    pos = {p: k for k, p in enumerate(rec["order0"])}
    for k, (kp, o) in enumerate(zip(kept, orders)):
        rec["extra"][f"keep_{k}"] = kp
        rec["extra"][f"order_{k}"] = np.array([pos[p] for p in o], dtype=np.int32)
    fed = [[G.sha(np.asarray(s["feeds_all"][k])) for k in ("adj_indices_sub_o", "adj_values_sub_o", "adj_indices_sub_t", "adj_values_sub_t")] for s in rec["steps"]]
    vs = tf1shim.all_variables()
    byname = lambda n: [v for v in vs if v.name == n and not np.array_equal(v.initial, v.value.detach().numpy())][0]   # noqa: E731
    U, V, tU, tV = byname("U"), byname("V"), byname("t_U"), byname("t_V")
    W = [v for v in vs if v.initial.shape == (8, 8) and v.name.startswith("Variable_")][0]
    b = [v for v in vs if v.initial.shape == (1, 8) and v.name.startswith("Variable_")][0]
    W.name, b.name = "online_mat", "online_bias"
    meta = pack(rec, "tf_buir_filmtrust", ["U", "V", "t_U", "t_V", "online_mat", "online_bias"], open(conf).read(), 106, dict(n_layers=2, tau=0.995, drop_rate=0.2))
    meta["n_keep_lists"] = len(kept)
    meta["fed_sha256"] = fed
    return meta


def social_conf(tmp, ratings, **kv):
    return base_conf(tmp, ratings, social="./dataset/FilmTrust/trust.txt", social__setup="-columns 0 1 2", **kv)


def case_sept(tmp, ratings):
    """three epochs: the first trains the recommendation task alone (v1_op), the other two the joint objective (v2_op) over a
    fresh perturbed graph each (two random.sample draws: rating edges, follow edges; SEPT.py:85-96, 276-292)"""
    conf = social_conf(tmp, ratings, model__name="SEPT", num__max__epoch="3", SEPT="-n_layer 2 -ss_rate 0.005 -drop_rate 0.3 -ins_cnt 10")
    from model.ranking.SEPT import SEPT
    kept, orders, restore = record_subgraph_draws(SEPT, "get_adj_mat")
    try:
        def after(m):
            U, V = m.sess.run([m.rec_user_embeddings, m.rec_item_embeddings])
            return dict(score_U=U.astype(np.float32), score_V=V.astype(np.float32),
                        follower=np.array([m.data.user[r[0]] for r in m.social.relation], dtype=np.int32),
                        followee=np.array([m.data.user[r[1]] for r in m.social.relation], dtype=np.int32))
        rec = run_tf_model(conf, 107, "model.ranking.SEPT", "SEPT", after=after, social=True)
    finally:
        restore()
    assert len(kept) == 2 * len(orders)
    pos = {p: k for k, p in enumerate(rec["order0"])}
    for k, o in enumerate(orders):
        rec["extra"][f"keep_{k}"] = kept[2 * k]; rec["extra"][f"skeep_{k}"] = kept[2 * k + 1]
        rec["extra"][f"order_{k}"] = np.array([pos[p] for p in o], dtype=np.int32)
    fed = [[G.sha(np.asarray(s["feeds_all"][k])) for k in ("adj_indices_sub", "adj_values_sub")] if "adj_indices_sub" in s["feeds_all"] else [] for s in rec["steps"]]
    meta = pack(rec, "tf_sept_filmtrust", ["U", "V"], open(conf).read(), 107, dict(n_layers=2, ss_rate=0.005, drop_rate=0.3, ins_cnt=10))
    meta["n_subgraphs"] = len(orders)
    meta["fed_sha256"] = fed
    return meta


def case_mhcn(tmp, ratings):
    conf = social_conf(tmp, ratings, model__name="MHCN", MHCN="-n_layer 2 -ss_rate 0.01")

    def after(m):
        U, V = m.sess.run([m.final_user_embeddings, m.final_item_embeddings])
        return dict(score_U=U.astype(np.float32), score_V=V.astype(np.float32),
                    follower=np.array([m.data.user[r[0]] for r in m.social.relation], dtype=np.int32),
                    followee=np.array([m.data.user[r[1]] for r in m.social.relation], dtype=np.int32))
    with np.errstate(divide="ignore"):
        rec = run_tf_model(conf, 108, "model.ranking.MHCN", "MHCN", after=after, social=True)
    rec["extra"]["train_r"] = np.array(rec["rating0"], dtype=np.float64)
    names = ["U", "V", "at", "atm"] + [f"{p}_{k}_1" for k in (1, 2, 3, 4) for p in ("g_W", "g_W_b", "sg_W", "sg_W_b")]
    return pack(rec, "tf_mhcn_filmtrust", names, open(conf).read(), 108, dict(n_layers=2, ss_rate=0.01))


CASES = None        # filled below main(): the case functions in generation order


def yardstick_main():
    """``--float64-yardstick``: the SAME runs of the reference's classes with the stand-in's arithmetic in float64 (TF1SHIM_DTYPE=float64;
    same seeds, so the same split, batches, negatives, dropout masks and shuffles -- asserted), SimGCL under the sign pattern its
    committed float32 run recorded (sign() is discontinuous; a float64 replay that re-decided an entry within rounding of zero
    would leave the float32 run by more than any rounding).  Written: tf_f64_yardstick.npz -- per model the trained variables and
    the scoring tables of that run, rounded to float32.  What it is for: |fixture - yardstick| is the distance of the reference's
    OWN float32 run from exact arithmetic, i.e. how far a correct implementation in any other summation order may land from the
    fixture (tests/test_gpu_tf_golden.py holds the HIP trainers to 1e-5 of the fixture where that distance allows it, and reports
    both distances everywhere).  Needs the committed float32 fixtures next to this file (it reads SimGCL's signs from there)."""
    global HERE
    if os.environ.get("TF1SHIM_DTYPE") != "float64":
        os.environ["TF1SHIM_DTYPE"] = "float64"
        os.execv(sys.executable, [sys.executable] + sys.argv)
    committed = HERE
    z = np.load(os.path.join(committed, "tf_simgcl_filmtrust.npz"))
    shape = tuple(int(x) for x in z["sign_shape"]); n = int(np.prod(shape))
    neg = np.unpackbits(z["sign_neg_bits"])[:n].reshape(shape).astype(bool)
    zero = np.unpackbits(z["sign_zero_bits"])[:n].reshape(shape).astype(bool)
    sg = np.where(zero, 0, np.where(neg, -1, 1)).astype(np.int8)                        # [steps, 4 sign ops, N, d]
    G.install_stubs()
    sys.modules["tensorflow"] = tf1shim
    if not hasattr(np, "mat"):
        np.mat = np.asmatrix
    out = {}
    with tempfile.TemporaryDirectory() as tmp, tempfile.TemporaryDirectory() as scratch:
        HERE = scratch                                                                   # pack() writes the float64 runs' files there
        os.symlink(os.path.join(REF, "dataset"), os.path.join(tmp, "dataset"))
        cwd = os.getcwd(); os.chdir(tmp)
        try:
            ratings, _ = make_subset(tmp)
            for case in CASES:
                tf1shim.FORCED_SIGNS.clear()
                if case is case_simgcl:
                    for step, run in enumerate(z["run_index"]):
                        for op in range(shape[1]):
                            tf1shim.FORCED_SIGNS[(int(run), op)] = sg[step, op]
                meta = case(tmp, ratings)
                name = meta["name"]
                a, b = np.load(os.path.join(committed, name + ".npz")), np.load(os.path.join(scratch, name + ".npz"))
                for k in ("train_uid", "train_iid", "batch_u", "batch_i", "batch_j", "run_index"):
                    assert k not in a.files or np.array_equal(a[k], b[k]), (name, k, "the float64 run left the float32 run's inputs")
                for k in b.files:
                    if k.startswith(("final_", "score_")):
                        out[f"{name}/{k}"] = b[k].astype(np.float32)
                print(name, "float32 run vs float64 run:", {k: float(np.linalg.norm(a[k].astype(np.float64) - b[k]) / np.linalg.norm(b[k])) for k in b.files
                                                              if k.startswith("final_") and b[k].size > 64})
        finally:
            os.chdir(cwd); HERE = committed
    np.savez_compressed(os.path.join(committed, "tf_f64_yardstick.npz"), **out)


def main():
    if "--float64-yardstick" in sys.argv:
        return yardstick_main()
    G.install_stubs()
    sys.modules["tensorflow"] = tf1shim
    if not hasattr(np, "mat"):
        np.mat = np.asmatrix          # removed in NumPy 2.0; SGL.py:85,90 still call it (the reference pins numpy 1.x)
    metas = {}
    with tempfile.TemporaryDirectory() as tmp:
        os.symlink(os.path.join(REF, "dataset"), os.path.join(tmp, "dataset"))
        cwd = os.getcwd(); os.chdir(tmp)
        try:
            ratings, n_rows = make_subset(tmp)
            for case in CASES:
                meta = case(tmp, ratings)
                meta["subset"] = dict(source="dataset/FilmTrust/ratings.txt", first_users=N_SUBSET_USERS, rows=n_rows)
                metas[meta["name"]] = meta
                print(meta["name"], "steps", meta["n_steps"], "users", meta["n_users"], "items", meta["n_items"])
        finally:
            os.chdir(cwd)
    with open(os.path.join(HERE, "golden_tf.json"), "w") as f:
        json.dump(metas, f, indent=1, sort_keys=True, default=str)


CASES = (case_lightgcn, case_bpr_tf, case_ngcf, case_simgcl, case_sgl, case_buir, case_sept, case_mhcn, case_sgl_random_walk, case_sgl_node_dropout)

if __name__ == "__main__":
    main()
