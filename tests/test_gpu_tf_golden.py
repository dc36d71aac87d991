"""The HIP trainers against runs of the REFERENCE'S OWN TensorFlow-path classes (tests/golden/tf_*_filmtrust.npz: produced by
tests/golden/gen_golden_tf.py, which executes model/ranking/{BPR-tf, LightGCN, NGCF, SimGCL}.py unmodified through a stand-in for the
tensorflow module -- see tests/test_oracle_tf_golden.py and DESIGN.md s2).  Same initial tables, the same batches the reference's
sampler drew, the same random draws (regenerated from their keys): the losses the reference printed at every step, the variables
it ended with and the tables it scores with.  fp32 on both sides, different summation orders: 1e-5-class agreement on the losses,
Adam-noise-class agreement on the tables (see the note in test_gpu_graph.py::test_simgcl_training_steps_match_restatement)."""
import json
import os

import numpy as np
import pytest

from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.graph import BprTfTrainer, LightGCNTrainer, NGCFTrainer, SimGCLTrainer, joint_norm_adjacency, unique_first_appearance

from helpers import pad_cols, rel_err

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class tf1shim:      # noqa: N801  -- the one function of tests/golden/tf1shim.py needed here, without importing torch into a GPU test process
    @staticmethod
    def random_uniform(seed, run_index, op_index, shape):
        return np.random.default_rng([int(seed), int(run_index), int(op_index)]).random(tuple(int(s) for s in shape), dtype=np.float32)


pytestmark = pytest.mark.gpu
META = json.load(open(os.path.join(HERE, "golden_tf.json")))


@pytest.fixture(scope="module", autouse=True)
def _device():
    capi.init(0)
    yield


def load(name):
    return META[name], np.load(os.path.join(HERE, name + ".npz"))


def batches(z):
    off = z["batch_offsets"]
    for k in range(off.size - 1):
        s = slice(off[k], off[k + 1])
        yield k, z["batch_u"][s].astype(np.int32), z["batch_i"][s].astype(np.int32), z["batch_j"][s].astype(np.int32)


def test_lightgcn_trainer_follows_the_reference_run():
    m, z = load("tf_lightgcn_filmtrust")
    nu, ni = m["n_users"], m["n_items"]
    adj = joint_norm_adjacency(nu, ni, z["train_uid"], z["train_iid"])
    tr = LightGCNTrainer(z["init_U"], z["init_V"], adj, m["n_layers"], lr=m["lr"], reg=m["regU"])
    for k, u, i, j in batches(z):
        tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), u.size)
        assert abs(tr.loss() - z["losses"][k, 0]) / z["losses"][k, 0] < 3e-5, k
    U, V = tr.ego_embeddings()
    assert rel_err(U, z["final_U"]) < 5e-4 and rel_err(V, z["final_V"]) < 5e-4
    Uf, Vf = tr.final_embeddings()
    assert rel_err(Uf, z["score_U"]) < 5e-4 and rel_err(Vf, z["score_V"]) < 5e-4


def test_bpr_tf_trainer_follows_the_reference_run():
    m, z = load("tf_bpr_filmtrust")
    tr = BprTfTrainer(z["init_U"], z["init_V"], m["lr"], m["regU"])
    for k, u, i, j in batches(z):
        tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), u.size)
        assert abs(tr.loss() - z["losses"][k, 0]) / z["losses"][k, 0] < 3e-5, k
    U, V = tr.tables()
    assert rel_err(U, z["final_U"]) < 5e-4 and rel_err(V, z["final_V"]) < 5e-4


def test_ngcf_trainer_follows_the_reference_run():
    m, z = load("tf_ngcf_filmtrust")
    nu, ni, dim = m["n_users"], m["n_items"], m["emb_size"]
    n = nu + ni
    adj = joint_norm_adjacency(nu, ni, z["train_uid"], z["train_iid"])
    W = [[z["init_W_0_1"], z["init_W_0_2"]], [z["init_W_1_1"], z["init_W_1_2"]]]
    tr = NGCFTrainer(z["init_U"], z["init_V"], W, adj, lr=m["lr"], reg=m["regU"])
    ops = sorted(r[0] for r in m["random_ops"][0])
    rate = np.float32(1.0 - m["keep_prob"])
    for k, u, i, j in batches(z):
        masks = [(tf1shim.random_uniform(m["seed"], z["run_index"][k], op, (n, dim)) >= rate).astype(np.float32) for op in ops]
        tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), u.size, masks=[DB.from_numpy(pad_cols(x, tr.ld)) for x in masks])
        assert abs(tr.loss() - z["losses"][k, 0]) / z["losses"][k, 0] < 1e-4, k
    U, V, Wg = tr.parameters()
    assert rel_err(U, z["final_U"]) < 2e-3 and rel_err(V, z["final_V"]) < 2e-3
    for a in range(2):
        for b in range(2):
            assert rel_err(Wg[a][b], z[f"final_W_{a}_{b + 1}"]) < 2e-3
    Ui, Vi = tr.inference_embeddings()
    assert rel_err(Ui, z["score_U"]) < 2e-3 and rel_err(Vi, z["score_V"]) < 2e-3


def test_simgcl_trainer_follows_the_reference_run():
    m, z = load("tf_simgcl_filmtrust")
    nu, ni, dim, L = m["n_users"], m["n_items"], m["emb_size"], m["n_layers"]
    n = nu + ni
    names = {role: name for name, role in m["var_roles"].items()}
    adj = joint_norm_adjacency(nu, ni, z["train_uid"], z["train_iid"])
    tr = SimGCLTrainer(z["init_" + names["U"]], z["init_" + names["V"]], adj, L, lr=m["lr"], reg=m["regU"], cl_rate=m["cl_rate"], eps=m["eps"],
                       max_unique=m["batch_size"])
    ops = sorted(r[0] for r in m["random_ops"][0])
    worst = []
    for k, u, i, j in batches(z):
        noises = [tf1shim.random_uniform(m["seed"], z["run_index"][k], op, (n, dim)) for op in ops]
        uu = unique_first_appearance(u).astype(np.int32); vv = (unique_first_appearance(i) + nu).astype(np.int32)
        tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), u.size, DB.from_numpy(uu), uu.size, DB.from_numpy(vv), vv.size,
                            noises=[DB.from_numpy(pad_cols(x, tr.ld)) for x in noises])      # [N][ld]: the kernels read whole padded rows
        got = np.array(tr.losses())
        err = np.abs(got - z["losses"][k]) / z["losses"][k]
        assert err.max() < 2e-3, (k, err)           # sign(emb) is discontinuous: an entry within rounding of zero may flip (test_oracle_tf_golden.py)
        worst.append(err.max())
    assert np.sum(np.array(worst) > 5e-5) <= 3, worst
    U, V = tr.ego_embeddings()
    E = np.concatenate([z["final_" + names["U"]], z["final_" + names["V"]]])
    assert rel_err(np.concatenate([U, V]), E) < 5e-3
    Um, Vm = tr.main_embeddings()
    assert rel_err(Um, z["score_U"]) < 5e-3 and rel_err(Vm, z["score_V"]) < 5e-3


# ---------------------------------------------------------------------------------------------------------------------
# SGL / BUIR / SEPT / MHCN trainers against their reference runs.  Written when round 2's GPU minutes were used up: the four
# tests above ran green on the MI355X, these have not been run there yet, so they are opt-in (QREC_RUN_PENDING_GOLDEN=1) until
# a GPU run has confirmed them -- a test that has never executed must not be able to turn the suite red or green.
# ---------------------------------------------------------------------------------------------------------------------
pending = pytest.mark.skipif(not os.environ.get("QREC_RUN_PENDING_GOLDEN"), reason="not yet validated on a GPU (set QREC_RUN_PENDING_GOLDEN=1)")


def _csr(a, n):
    import scipy.sparse as sp
    return sp.csr_matrix((a[2], a[1], a[0]), shape=(n, n))


@pending
@pytest.mark.parametrize("name", ["tf_sgl_filmtrust", "tf_sgl_rw_filmtrust", "tf_sgl_nd_filmtrust"])
def test_sgl_trainer_follows_the_reference_run(name):
    from qrec_amd.graph import SGLTrainer
    m, z = load(name)
    nu, ni, L, aug = m["n_users"], m["n_items"], m["n_layers"], m["aug_type"]
    adj = joint_norm_adjacency(nu, ni, z["train_uid"], z["train_iid"])
    tr = SGLTrainer(z["init_U"], z["init_V"], adj, L, lr=m["lr"], reg=m["regU"], ssl_reg=m["ssl_reg"], temp=m["temp"], max_unique=2 * m["batch_size"])
    n_epochs = 2
    steps_per_epoch, per_epoch = m["n_steps"] // n_epochs, m["n_subgraphs"] // n_epochs
    subs = []
    for k in range(m["n_subgraphs"]):
        uid, iid = z["train_uid"][z[f"order_{k}"]], z["train_iid"][z[f"order_{k}"]]
        if aug == 0:
            alive = ~np.isin(uid, z[f"keep_{2 * k}"]) & ~np.isin(iid, z[f"keep_{2 * k + 1}"])
            subs.append(joint_norm_adjacency(nu, ni, uid[alive], iid[alive]))
        else:
            subs.append(joint_norm_adjacency(nu, ni, uid[z[f"keep_{k}"]], iid[z[f"keep_{k}"]]))
    for k, u, i, j in batches(z):
        e = k // steps_per_epoch
        if k % steps_per_epoch == 0:
            mine = subs[per_epoch * e:per_epoch * (e + 1)]
            if aug == 2:
                tr.set_subgraphs([mine[2 * l] for l in range(L)], [mine[2 * l + 1] for l in range(L)])
            else:
                tr.set_subgraphs(mine[0], mine[1])
        rows = np.concatenate([unique_first_appearance(u), unique_first_appearance(i) + nu]).astype(np.int32)
        tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), u.size, DB.from_numpy(rows), rows.size)
        got = np.array(tr.losses())
        assert (np.abs(got - z["losses"][k]) / z["losses"][k]).max() < 5e-5, k
    U, V = tr.ego_embeddings()
    assert rel_err(U, z["final_U"]) < 1e-3 and rel_err(V, z["final_V"]) < 1e-3
    Um, Vm = tr.main_embeddings()
    assert rel_err(Um, z["score_U"]) < 1e-3 and rel_err(Vm, z["score_V"]) < 1e-3


@pending
def test_buir_trainer_follows_the_reference_run():
    from qrec_amd.graph import BUIRTrainer
    m, z = load("tf_buir_filmtrust")
    nu, ni, L = m["n_users"], m["n_items"], m["n_layers"]
    tr = BUIRTrainer(z["init_U"], z["init_V"], z["init_online_mat"], z["init_online_bias"], L, lr=m["lr"], tau=m["tau"])
    steps_per_epoch = m["n_steps"] // 2
    subs = []
    for k in range(m["n_keep_lists"]):
        keep = z[f"order_{k}"][z[f"keep_{k}"]]
        subs.append(joint_norm_adjacency(nu, ni, z["train_uid"][keep], z["train_iid"][keep]))
    off = z["batch_offsets"]
    for k in range(m["n_steps"]):
        e = k // steps_per_epoch
        if k % steps_per_epoch == 0:
            tr.set_subgraphs(subs[2 * e], subs[2 * e + 1])
        u = z["batch_u"][off[k]:off[k + 1]].astype(np.int32); i = z["batch_i"][off[k]:off[k + 1]].astype(np.int32)
        tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), u.size)
        assert abs(tr.loss() - z["losses"][k, 0]) / z["losses"][k, 0] < 1e-4, k
    E = np.concatenate([z["final_U"], z["final_V"]]); Tt = np.concatenate([z["final_t_U"], z["final_t_V"]])
    assert rel_err(tr.online_tables(), E) < 1e-3 and rel_err(tr.target_tables(), Tt) < 1e-3
    Wg, bg = tr.weights()
    assert rel_err(Wg, z["final_online_mat"]) < 1e-3 and rel_err(bg, z["final_online_bias"].ravel()) < 2e-3
    adj = joint_norm_adjacency(nu, ni, z["train_uid"], z["train_iid"])
    for g, key in zip(tr.final_tables(adj), ("q_user", "q_item", "o_user", "o_item")):
        assert rel_err(g, z[key]) < 1e-3, key


@pending
def test_sept_trainer_follows_the_reference_run():
    from oracle import tfmodels as T      # the oracle's scipy graph builders (themselves bit-identical to the reference's, test_oracle_golden.py)
    from qrec_amd.graph import SEPTTrainer
    m, z = load("tf_sept_filmtrust")
    nu, ni, L = m["n_users"], m["n_items"], m["n_layers"]
    uid, iid, fo, fe = z["train_uid"], z["train_iid"], z["follower"], z["followee"]
    adj = T.sept_sub_adjacency(nu, ni, uid, iid, fo, fe)
    social, sharing = T.sept_social_views(nu, ni, uid, iid, fo, fe)
    tr = SEPTTrainer(z["init_U"], z["init_V"], adj.astype(np.float32), social, sharing, L, m["lr"], m["regU"], m["ss_rate"], m["ins_cnt"], max_unique=m["batch_size"])
    n_epochs = 3
    steps_per_epoch = m["n_steps"] // n_epochs
    joint_epochs = [e for e in range(n_epochs) if e > n_epochs / 3]
    for k, u, i, j in batches(z):
        e = k // steps_per_epoch
        joint = e in joint_epochs
        if joint and k % steps_per_epoch == 0:
            s = joint_epochs.index(e)
            order = z[f"order_{s}"]
            tr.set_perturbed_graph(T.sept_sub_adjacency(nu, ni, uid[order], iid[order], fo, fe, z[f"keep_{s}"], z[f"skeep_{s}"]))
        uu = unique_first_appearance(u).astype(np.int32)
        tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), u.size, joint, DB.from_numpy(uu), uu.size)
        got = tr.losses()
        assert got[0] == pytest.approx(z["losses"][k, 0], rel=5e-5), k
        if joint:       # pseudo labels are a top-k of float32 softmax rows: a near-tie may pick another neighbour
            assert got[1] == pytest.approx(m["ss_rate"] * z["losses"][k, 1], rel=2e-3), k
    U, V = tr.variables()
    assert rel_err(np.concatenate([U, V]), np.concatenate([z["final_U"], z["final_V"]])) < 2e-3
    Ur, Vr = tr.rec_embeddings()
    assert rel_err(Ur, z["score_U"]) < 2e-3 and rel_err(Vr, z["score_V"]) < 2e-3


@pending
def test_mhcn_trainer_follows_the_reference_run():
    from oracle import tfmodels as T
    from qrec_amd.graph import MHCNTrainer
    m, z = load("tf_mhcn_filmtrust")
    nu, ni, L, d = m["n_users"], m["n_items"], m["n_layers"], m["emb_size"]
    with np.errstate(divide="ignore", invalid="ignore"):
        H = T.mhcn_motif_adjacencies(nu, ni, z["train_uid"], z["train_iid"], z["follower"], z["followee"])
    Rm = T.mhcn_joint_adjacency(nu, ni, z["train_uid"], z["train_iid"], z["train_r"])
    key = {"attention": "at", "attention_mat": "atm"}
    for c in (1, 2, 3, 4):
        key[f"gating{c}"] = f"g_W_{c}_1"; key[f"gating_bias{c}"] = f"g_W_b_{c}_1"; key[f"sgating{c}"] = f"sg_W_{c}_1"; key[f"sgating_bias{c}"] = f"sg_W_b_{c}_1"
    tr = MHCNTrainer(z["init_U"], z["init_V"], {a: z["init_" + b] for a, b in key.items()}, H, Rm, L, m["lr"], m["regU"], m["ss_rate"])
    ops = sorted(m["random_ops"][0])
    for k, u, i, j in batches(z):
        draws = [np.argsort(tf1shim.random_uniform(m["seed"], z["run_index"][k], r[0], r[2]), kind="stable") for r in ops]
        tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), u.size, perms=[tuple(draws[5 * c:5 * c + 5]) for c in range(3)])
        rec, _ = tr.losses()
        assert rec == pytest.approx(z["losses"][k, 0], rel=1e-4), k
    got = tr.parameters()
    for a, b in key.items():
        assert rel_err(got[a], z["final_" + b]) < 2e-3, a
    assert rel_err(got["U"], z["final_U"]) < 2e-3 and rel_err(got["V"], z["final_V"]) < 2e-3
    Ud, Vd = tr.final_embeddings()
    assert rel_err(Ud, z["score_U"]) < 2e-3 and rel_err(Vd, z["score_V"]) < 2e-3
