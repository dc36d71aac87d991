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

from helpers import check, check_rel, pad_cols, rel_err

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


GRAD_TOL = 1e-5         # north_star: 1e-5 relative on fp32 quantities


def grad_check(got, want, what, bound=GRAD_TOL):
    """a HIP trainer's first-step gradient against the one the reference's minimize() applied (grad<k>_<var> of the fixture):
    same variables on both sides, BEFORE Adam -- the 12-step comparisons further down are the drift check"""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    check(what, rel_err(got, want), bound)


SOLID_FLOOR = 0.1


def solid_check(what, got, ref, grad0, loose, solid_bound=GRAD_TOL):
    """Trained variables after the fixture's 12-18 Adam steps.  Adam's step is lr * m / (sqrt(v) + eps): on a
    coordinate whose gradient is rounding noise the step is ~lr in a direction the noise decides, so two correct fp32 implementations
    end |lr x steps| apart there whatever their parity.  SOLID coordinates -- |first-step gradient| >= SOLID_FLOOR x the variable's
    largest |first-step gradient| (the fixture's grad0) -- are held to north_star's 1e-5; the whole variable to the documented looser
    bound ``loose``.  Measured by floor (QREC_SOLID_PROBE=1, round 5, relative error on the coordinates at or above the floor / share of
    the entries): SimGCL 4.9e-5 on all -> 1.3e-5 at 0.01 (82 %) -> 8.0e-6 at 0.05 (27 %) -> 6.3e-6 at 0.1 (10 %); SGL 1.3e-6 ... 3.3e-6 on all,
    <= 1.1e-6 at 0.1; BUIR 1.2e-7 / 7.7e-8; MHCN item table 5.4e-6 / 7.3e-8.  SEPT is the exception that the rule does NOT describe: 1.3e-5 on
    all coordinates and 2.8e-5 at 0.1 -- its deviation sits on large-gradient coordinates (float-atomic scatter of the self-supervised
    gradient over rows shared by several contrast sets, a discontinuous pseudo-label top-k), so it keeps its documented bound on both."""
    got, ref, g = np.asarray(got, np.float64), np.asarray(ref, np.float64), np.abs(np.asarray(grad0, np.float64))
    assert got.shape == ref.shape == g.shape, (what, got.shape, ref.shape, g.shape)
    if os.environ.get("QREC_SOLID_PROBE"):      # development: the error by floor, into the ledger; nothing asserted
        for fl in (0.0, 1e-2, 3e-2, 5e-2, 1e-1, 2e-1, 3e-1):
            mk = g >= fl * g.max()
            check(f"probe {what} floor {fl:g} share {mk.mean():.3f}", rel_err(got[mk], ref[mk]), 1.0)
        return
    solid = g >= SOLID_FLOOR * g.max()
    check(f"{what}: solid coordinates (|first-step gradient| >= {SOLID_FLOOR:g} of its maximum)", rel_err(got[solid], ref[solid]), solid_bound, ctx=float(solid.mean()))
    check(f"{what}: all coordinates", rel_err(got, ref), loose)


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
        check("abs(tr.loss() - z['losses'][k, 0]) / z['losses'][k, 0]", abs(tr.loss() - z["losses"][k, 0]) / z["losses"][k, 0], 1e-5, ctx=k)
        if k == 0:
            gU, gV = tr.gradients()
            grad_check(gU, z["grad0_U"], "LightGCN dU, step 0"); grad_check(gV, z["grad0_V"], "LightGCN dV, step 0")
    U, V = tr.ego_embeddings()
    check("rel_err(U, z['final_U'])", rel_err(U, z["final_U"]), 1e-5)
    check("rel_err(V, z['final_V'])", rel_err(V, z["final_V"]), 1e-5)
    Uf, Vf = tr.final_embeddings()
    check("rel_err(Uf, z['score_U'])", rel_err(Uf, z["score_U"]), 1e-5)
    check("rel_err(Vf, z['score_V'])", rel_err(Vf, z["score_V"]), 1e-5)


def test_bpr_tf_trainer_follows_the_reference_run():
    m, z = load("tf_bpr_filmtrust")
    tr = BprTfTrainer(z["init_U"], z["init_V"], m["lr"], m["regU"])
    for k, u, i, j in batches(z):
        tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), u.size)
        check("abs(tr.loss() - z['losses'][k, 0]) / z['losses'][k, 0]", abs(tr.loss() - z["losses"][k, 0]) / z["losses"][k, 0], 1e-5, ctx=k)
        if k == 0:
            gU, gV = tr.gradients(np.concatenate([z["init_U"], z["init_V"]]))
            grad_check(gU, z["grad0_U"], "BPR-tf dU, step 0"); grad_check(gV, z["grad0_V"], "BPR-tf dV, step 0")
    U, V = tr.tables()
    check("rel_err(U, z['final_U'])", rel_err(U, z["final_U"]), 1e-5)
    check("rel_err(V, z['final_V'])", rel_err(V, z["final_V"]), 1e-5)


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
        check("abs(tr.loss() - z['losses'][k, 0]) / z['losses'][k, 0]", abs(tr.loss() - z["losses"][k, 0]) / z["losses"][k, 0], 1e-5, ctx=k)
        if k == 0:
            gU, gV, gW = tr.gradients()
            grad_check(gU, z["grad0_U"], "NGCF dU, step 0"); grad_check(gV, z["grad0_V"], "NGCF dV, step 0")
            for a in range(2):
                for b in range(2):
                    grad_check(gW[a][b], z[f"grad0_W_{a}_{b + 1}"], f"NGCF dW_{a}_{b + 1}, step 0")
    U, V, Wg = tr.parameters()
    check("rel_err(U, z['final_U'])", rel_err(U, z["final_U"]), 1e-5)
    check("rel_err(V, z['final_V'])", rel_err(V, z["final_V"]), 1e-5)
    for a in range(2):
        for b in range(2):
            check("rel_err(Wg[a][b], z[f'final_W_{a}_{b + 1}'])", rel_err(Wg[a][b], z[f"final_W_{a}_{b + 1}"]), 1e-5)
    Ui, Vi = tr.inference_embeddings()
    check("rel_err(Ui, z['score_U'])", rel_err(Ui, z["score_U"]), 1e-5)
    check("rel_err(Vi, z['score_V'])", rel_err(Vi, z["score_V"]), 1e-5)


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
        if k == 0:
            gU, gV = tr.gradients()
            grad_check(gU, z["grad0_" + names["U"]], "SimGCL dU, step 0"); grad_check(gV, z["grad0_" + names["V"]], "SimGCL dV, step 0")
        check("SimGCL total / rec / cl loss vs reference run (worst of the three)", err.max(), 0.001, ctx=(k, err))           # sign(emb) is discontinuous: an entry within rounding of zero may flip (test_oracle_tf_golden.py)
        worst.append(err.max())
    assert np.sum(np.array(worst) > 5e-5) <= 3, worst
    U, V = tr.ego_embeddings()
    E = np.concatenate([z["final_" + names["U"]], z["final_" + names["V"]]])
    check("rel_err(np.concatenate([U, V]), E)", rel_err(np.concatenate([U, V]), E), 0.002)
    Um, Vm = tr.main_embeddings()
    check("rel_err(Um, z['score_U'])", rel_err(Um, z["score_U"]), 0.002)
    check("rel_err(Vm, z['score_V'])", rel_err(Vm, z["score_V"]), 0.002)


def test_simgcl_trainer_with_the_recorded_sign_pattern_follows_the_reference_run_throughout():
    """The test above allows three steps out of twelve to leave 5e-5 because `sign(emb)` (SimGCL.py:35) is discontinuous at 0 -- the
    CPU restatement names the flip: step 10, view 2, layer 2, row 805, column 1, an entry of -2.1e-6 in a row of magnitude 0.07
    (tests/test_oracle_tf_golden.py).  Here the excuse is removed: the noise fed to the kernels carries the sign every perturbation of
    the reference's own run used (recorded from its `tf.sign` ops by the generator; include/qrec_hip.h, qrec_perturb_rows), and the
    HIP trainer is held to the run at 1e-5 on the three losses of ALL twelve steps; the trained tables to the 5e-5 of the other
    contrastive models (twelve Adam steps on rounding-noise coordinates)."""
    from helpers import encode_forced_signs, simgcl_recorded_signs
    m, z = load("tf_simgcl_filmtrust")
    nu, ni, dim, L = m["n_users"], m["n_items"], m["emb_size"], m["n_layers"]
    n = nu + ni
    names = {role: name for name, role in m["var_roles"].items()}
    adj = joint_norm_adjacency(nu, ni, z["train_uid"], z["train_iid"])
    tr = SimGCLTrainer(z["init_" + names["U"]], z["init_" + names["V"]], adj, L, lr=m["lr"], reg=m["regU"], cl_rate=m["cl_rate"], eps=m["eps"],
                       max_unique=m["batch_size"])
    ops = sorted(r[0] for r in m["random_ops"][0])
    for k, u, i, j in batches(z):
        signs = simgcl_recorded_signs(z, k)
        noises = [encode_forced_signs(tf1shim.random_uniform(m["seed"], z["run_index"][k], op, (n, dim)), sg) for op, sg in zip(ops, signs)]
        uu = unique_first_appearance(u).astype(np.int32); vv = (unique_first_appearance(i) + nu).astype(np.int32)
        tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), u.size, DB.from_numpy(uu), uu.size, DB.from_numpy(vv), vv.size,
                            noises=[DB.from_numpy(pad_cols(x, tr.ld)) for x in noises])
        err = np.abs(np.array(tr.losses()) - z["losses"][k]) / z["losses"][k]
        check("SimGCL total / rec / cl loss vs reference run under its recorded sign pattern (worst of the three)", err.max(), 1e-5, ctx=(k, err))
    U, V = tr.ego_embeddings()
    E = np.concatenate([z["final_" + names["U"]], z["final_" + names["V"]]])
    # (observed: losses <= 7e-7 on all twelve steps; tables 3e-5, main embeddings 6e-5 -- the CPU restatement under the same signs: 2.7e-5 /
    # 3.5e-5; without the recorded signs: 2.8e-4 ... 4.2e-4.  Twelve Adam steps move a coordinate whose gradient is rounding noise by
    # ~lr per step in a direction the noise decides; the main embeddings are two propagations of those tables)
    solid_check("SimGCL tables after 12 steps under the recorded sign pattern", np.concatenate([U, V]), E,
                np.concatenate([z["grad0_" + names["U"]], z["grad0_" + names["V"]]]), 1e-4)
    Um, Vm = tr.main_embeddings()
    check("SimGCL main user embeddings under the recorded sign pattern", rel_err(Um, z["score_U"]), 1e-4)
    check("SimGCL main item embeddings under the recorded sign pattern", rel_err(Vm, z["score_V"]), 1e-4)


# ---------------------------------------------------------------------------------------------------------------------
# SGL / BUIR / SEPT / MHCN trainers against their reference runs
# ---------------------------------------------------------------------------------------------------------------------


def _csr(a, n):
    import scipy.sparse as sp
    return sp.csr_matrix((a[2], a[1], a[0]), shape=(n, n))


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
        check("SGL total / rec / ssl loss vs reference run (worst of the three)", (np.abs(got - z["losses"][k]) / z["losses"][k]).max(), 1e-5, ctx=k)
        if k == 0:
            gU, gV = tr.gradients()
            grad_check(gU, z["grad0_U"], f"SGL aug {aug} dU, step 0"); grad_check(gV, z["grad0_V"], f"SGL aug {aug} dV, step 0")
    U, V = tr.ego_embeddings()
    solid_check(f"SGL aug {aug} user table after the run", U, z["final_U"], z["grad0_U"], 1e-5)
    solid_check(f"SGL aug {aug} item table after the run", V, z["final_V"], z["grad0_V"], 2e-5)
    Um, Vm = tr.main_embeddings()
    check("rel_err(Um, z['score_U'])", rel_err(Um, z["score_U"]), 1e-5)
    check("rel_err(Vm, z['score_V'])", rel_err(Vm, z["score_V"]), 2e-5)


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
        check("abs(tr.loss() - z['losses'][k, 0]) / z['losses'][k, 0]", abs(tr.loss() - z["losses"][k, 0]) / z["losses"][k, 0], 1e-5, ctx=k)
        if k == 0:
            gE, gW, gb = tr.gradients()
            grad_check(gE[:nu], z["grad0_U"], "BUIR dU, step 0"); grad_check(gE[nu:], z["grad0_V"], "BUIR dV, step 0")
            grad_check(gW, z["grad0_online_mat"], "BUIR dW, step 0"); grad_check(gb[None, :], z["grad0_online_bias"], "BUIR db, step 0")
    E = np.concatenate([z["final_U"], z["final_V"]]); Tt = np.concatenate([z["final_t_U"], z["final_t_V"]])
    solid_check("BUIR online tables after the run", tr.online_tables(), E, np.concatenate([z["grad0_U"], z["grad0_V"]]), 1e-5)
    check("rel_err(tr.target_tables(), Tt)", rel_err(tr.target_tables(), Tt), 1e-5)
    Wg, bg = tr.weights()
    check("rel_err(Wg, z['final_online_mat'])", rel_err(Wg, z["final_online_mat"]), 1e-5)
    check("rel_err(bg, z['final_online_bias'].ravel())", rel_err(bg, z["final_online_bias"].ravel()), 1e-5)
    adj = joint_norm_adjacency(nu, ni, z["train_uid"], z["train_iid"])
    for g, key in zip(tr.final_tables(adj), ("q_user", "q_item", "o_user", "o_item")):
        check("rel_err(g, z[key])", rel_err(g, z[key]), 1e-5, ctx=key)


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
    ssl_err = []
    for k, u, i, j in batches(z):
        e = k // steps_per_epoch
        joint = e in joint_epochs
        if joint and k % steps_per_epoch == 0:
            s = joint_epochs.index(e)
            order = z[f"order_{s}"]
            tr.set_perturbed_graph(T.sept_sub_adjacency(nu, ni, uid[order], iid[order], fo, fe, z[f"keep_{s}"], z[f"skeep_{s}"]))
        uu = unique_first_appearance(u).astype(np.int32)
        if k in m["first_steps"]:       # first step of v1_op / v2_op (SEPT.py:267-270): the gradient from the REFERENCE'S variables at that point
            n_op = m["first_steps"].index(k)
            W_mine = tr.W.numpy()
            before = np.concatenate([z["init_U"], z["init_V"]]) if n_op == 0 else np.concatenate([z[f"pre{n_op}_U"], z[f"pre{n_op}_V"]])
            tr.W.upload(pad_cols(before.astype(np.float32), tr.ld))
            probe = tr.opt[n_op]
            saved = (probe.m.numpy(), probe.v.numpy(), probe.b1p, probe.b2p)
            tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), u.size, joint, DB.from_numpy(uu), uu.size)
            gU, gV = tr.gradients(before)
            grad_check(gU, z[f"grad{n_op}_U"], f"SEPT dU, first step of train op {n_op}"); grad_check(gV, z[f"grad{n_op}_V"], f"SEPT dV, first step of train op {n_op}")
            tr.W.upload(W_mine); probe.m.upload(saved[0]); probe.v.upload(saved[1]); probe.b1p, probe.b2p = saved[2], saved[3]     # undo the probe step
        tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), u.size, joint, DB.from_numpy(uu), uu.size)
        got = tr.losses()
        check_rel("SEPT rec loss vs reference run", got[0], z["losses"][k, 0], 1e-5, ctx=k)
        if joint:
            want = m["ss_rate"] * z["losses"][k, 1]
            ssl_err.append(abs(got[1] - want) / abs(want))
    # The pseudo labels (SEPT.py:190-211) are a top-k over float32 softmax rows -- discontinuous, like SimGCL's sign(): a near-tie may
    # pick another neighbour here than in the reference's run, and the float atomics' summation order differs from launch to launch,
    # so WHICH run meets a near-tie changes (rounds 2-3: every joint step <= 2e-7 in every run; round 4: one run with ONE step -- the
    # last -- at 1.6e-5, the other five <= 2e-7).  So: all joint steps but at most one at the 1e-5 of every other loss, that one
    # inside 1e-4 (a wrong pseudo label on one of ~1,000 contrast rows; an algorithmic difference is >= 1e-2).
    ssl_err = sorted(ssl_err)
    assert len(ssl_err) >= 6
    check("SEPT ssl loss vs reference run (all joint steps but the worst one)", ssl_err[-2], 1e-5, inclusive=True)
    check("SEPT ssl loss vs reference run (the worst joint step: at most one near-tie of the pseudo-label top-k)", ssl_err[-1], 1e-4, inclusive=True)
    U, V = tr.variables()
    # the drift check after 18 Adam steps (12 rec-only + 6 joint): losses and the pre-Adam gradients above are the 1e-5 statement; the
    # trained tables carry what Adam makes of last-bit gradient differences on coordinates whose gradient is ~0 (the step is
    # normalised to lr whatever the gradient's size) and of the float atomics' summation order, which changes from launch to
    # launch -- observed 2.5e-6 ... 1.3e-5 over the runs of this round
    solid_check("SEPT tables after 18 steps", np.concatenate([U, V]), np.concatenate([z["final_U"], z["final_V"]]), np.concatenate([z["grad0_U"], z["grad0_V"]]), 5e-5,
                solid_bound=5e-5)          # (not Adam noise on ~0-gradient coordinates: see solid_check's docstring)
    Ur, Vr = tr.rec_embeddings()
    check("rel_err(Ur, z['score_U'])", rel_err(Ur, z["score_U"]), 5e-5)
    check("rel_err(Vr, z['score_V'])", rel_err(Vr, z["score_V"]), 5e-5)


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
        check_rel("MHCN rec loss vs reference run", rec, z["losses"][k, 0], 1e-5, ctx=k)
        if k == 0:
            before = {a: z["init_" + b] for a, b in key.items()}; before["U"], before["V"] = z["init_U"], z["init_V"]
            g = tr.gradients(before)
            grad_check(g["U"], z["grad0_U"], "MHCN dU, step 0"); grad_check(g["V"], z["grad0_V"], "MHCN dV, step 0")
            for a, b in key.items():
                grad_check(g[a].reshape(z["grad0_" + b].shape), z["grad0_" + b], f"MHCN d{a}, step 0")
    got = tr.parameters()
    for a, b in key.items():
        check("rel_err(got[a], z['final_' + b])", rel_err(got[a], z["final_" + b]), 1e-5, ctx=a)
    solid_check("MHCN user table after the run", got["U"], z["final_U"], z["grad0_U"], 1e-5)
    solid_check("MHCN item table after the run", got["V"], z["final_V"], z["grad0_V"], 5e-5)
    Ud, Vd = tr.final_embeddings()
    check("rel_err(Ud, z['score_U'])", rel_err(Ud, z["score_U"]), 1e-5)
    check("rel_err(Vd, z['score_V'])", rel_err(Vd, z["score_V"]), 5e-5)
