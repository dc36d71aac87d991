"""The HIP trainers against runs of the REFERENCE'S OWN TensorFlow-path classes (tests/golden/tf_*_filmtrust.npz: produced by
tests/golden/gen_golden_tf.py, which executes model/ranking/{BPR-tf, LightGCN, NGCF, SimGCL}.py unmodified through a stand-in for the
tensorflow module -- see tests/test_oracle_tf_golden.py and DESIGN.md s2).  Same initial tables, the same batches the reference's
sampler drew, the same random draws (regenerated from their keys): the losses the reference printed at every step, the variables
it ended with and the tables it scores with.  fp32 on both sides, different summation orders.

Round 6: every trainer here is built under ``ordered_reductions()`` -- the parity mode the drop-in classes run in by default (exact
mode): batch gradients are added row by row in the reference's CPU order (csrc/ordered.hip), MHCN's column sums in a fixed tree, no
float atomics anywhere on the path.  Two consequences, both asserted: (1) every run is executed TWICE and the two results are
bit-identical; (2) trained variables are held to north_star's 1e-5 on ALL coordinates (the solid-coordinate carve-out of round 5 is
gone).  The one quantity that cannot be: SimGCL's item table, where the reference's OWN float32 run sits 3.2e-5 from the same run in
float64 arithmetic (tests/golden/tf_f64_yardstick.npz) -- no implementation in another summation order can be nearer to the fixture
than the fixture is to the truth; that test states its bound as a multiple of that measured distance (kind "floor")."""
import json
import os

import numpy as np
import pytest

from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.graph import BprTfTrainer, LightGCNTrainer, NGCFTrainer, SimGCLTrainer, joint_norm_adjacency, ordered_reductions, unique_first_appearance

from helpers import check, check_rel, pad_cols, rel_err, same_bits

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


YARD = np.load(os.path.join(HERE, "tf_f64_yardstick.npz"))


def table_check(what, got, ref, name, key, bound=GRAD_TOL):
    """A trained variable (or scoring table) after the fixture's 12-18 Adam steps against the reference's run: north_star's 1e-5 on
    the WHOLE variable, every coordinate (round 6: the trainers run with ordered reductions -- qrec_amd.graph.ordered_reductions, the
    exact mode of the drop-in classes -- and the solid-coordinate carve-out of round 5 is gone).
    Beside it, recorded (kind "info"): the distance of the HIP result and of the reference's own float32 run from the SAME run in
    float64 arithmetic (tests/golden/tf_f64_yardstick.npz, gen_golden_tf.py --float64-yardstick) -- the reference sits 2e-6 ... 7e-6 from
    exact arithmetic on these variables, so 1e-5 is a bar a correct fp32 implementation can meet and a wrong one cannot."""
    got, ref = np.asarray(got), np.asarray(ref)
    f64 = YARD[f"{name}/{key}"].reshape(ref.shape)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    check(f"{what}: vs the reference's run", rel_err(got, ref), bound)
    check(f"{what}: HIP vs the float64 run of the reference's classes", rel_err(got, f64), 1.0, kind="info")
    check(f"{what}: the reference's float32 run vs its float64 run", rel_err(ref, f64), 1.0, kind="info")


def batches(z):
    off = z["batch_offsets"]
    for k in range(off.size - 1):
        s = slice(off[k], off[k + 1])
        yield k, z["batch_u"][s].astype(np.int32), z["batch_i"][s].astype(np.int32), z["batch_j"][s].astype(np.int32)


def _parity_twice(what, run):
    """build and run the trainer twice under ordered reductions: the reference run's bounds both times, and the same bits"""
    with ordered_reductions():
        a = run()
        b = run()
    same_bits(what, a, b)


def test_lightgcn_trainer_follows_the_reference_run():
    _parity_twice("LightGCN", _run_lightgcn)


def _run_lightgcn():
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
    table_check("LightGCN user table after the run", U, z["final_U"], "tf_lightgcn_filmtrust", "final_U")
    table_check("LightGCN item table after the run", V, z["final_V"], "tf_lightgcn_filmtrust", "final_V")
    Uf, Vf = tr.final_embeddings()
    table_check("LightGCN scoring users", Uf, z["score_U"], "tf_lightgcn_filmtrust", "score_U")
    table_check("LightGCN scoring items", Vf, z["score_V"], "tf_lightgcn_filmtrust", "score_V")
    return dict(U=U, V=V, Uf=Uf, Vf=Vf)


def test_bpr_tf_trainer_follows_the_reference_run():
    _parity_twice("BPR-tf", _run_bpr_tf)


def _run_bpr_tf():
    m, z = load("tf_bpr_filmtrust")
    tr = BprTfTrainer(z["init_U"], z["init_V"], m["lr"], m["regU"])
    for k, u, i, j in batches(z):
        tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), u.size)
        check("abs(tr.loss() - z['losses'][k, 0]) / z['losses'][k, 0]", abs(tr.loss() - z["losses"][k, 0]) / z["losses"][k, 0], 1e-5, ctx=k)
        if k == 0:
            gU, gV = tr.gradients(np.concatenate([z["init_U"], z["init_V"]]))
            grad_check(gU, z["grad0_U"], "BPR-tf dU, step 0"); grad_check(gV, z["grad0_V"], "BPR-tf dV, step 0")
    U, V = tr.tables()
    table_check("BPR-tf user table after the run", U, z["final_U"], "tf_bpr_filmtrust", "final_U")
    table_check("BPR-tf item table after the run", V, z["final_V"], "tf_bpr_filmtrust", "final_V")
    return dict(U=U, V=V)


def test_ngcf_trainer_follows_the_reference_run():
    _parity_twice("NGCF", _run_ngcf)


def _run_ngcf():
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
    out = dict(U=U, V=V)
    table_check("NGCF user table after the run", U, z["final_U"], "tf_ngcf_filmtrust", "final_U")
    table_check("NGCF item table after the run", V, z["final_V"], "tf_ngcf_filmtrust", "final_V")
    for a in range(2):
        for b in range(2):
            table_check(f"NGCF W_{a}_{b + 1} after the run", Wg[a][b], z[f"final_W_{a}_{b + 1}"], "tf_ngcf_filmtrust", f"final_W_{a}_{b + 1}")
            out[f"W_{a}_{b + 1}"] = Wg[a][b]
    Ui, Vi = tr.inference_embeddings()
    table_check("NGCF scoring users", Ui, z["score_U"], "tf_ngcf_filmtrust", "score_U")
    table_check("NGCF scoring items", Vi, z["score_V"], "tf_ngcf_filmtrust", "score_V")
    out.update(Ui=Ui, Vi=Vi)
    return out


def test_simgcl_trainer_follows_the_reference_run():
    """the product's own sign() path (no recorded pattern): `sign(emb)` (SimGCL.py:35) is discontinuous at 0, an entry within rounding of
    zero may take the other sign than in the recorded run, and from that step on the run is another (equally valid) run -- the rows
    below are kind "discontinuity"; the recorded-pattern test after this one is the parity statement on the same data"""
    _parity_twice("SimGCL", _run_simgcl)


def _run_simgcl():
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
        check("SimGCL total / rec / cl loss vs reference run (worst of the three)", err.max(), 0.001, ctx=(k, err), kind="discontinuity")           # sign(emb) is discontinuous: an entry within rounding of zero may flip (test_oracle_tf_golden.py)
        worst.append(err.max())
    assert np.sum(np.array(worst) > 5e-5) <= 3, worst
    U, V = tr.ego_embeddings()
    E = np.concatenate([z["final_" + names["U"]], z["final_" + names["V"]]])
    check("SimGCL tables, own sign pattern", rel_err(np.concatenate([U, V]), E), 0.002, kind="discontinuity")
    Um, Vm = tr.main_embeddings()
    check("SimGCL main user embeddings, own sign pattern", rel_err(Um, z["score_U"]), 0.002, kind="discontinuity")
    check("SimGCL main item embeddings, own sign pattern", rel_err(Vm, z["score_V"]), 0.002, kind="discontinuity")
    return dict(U=U, V=V, Um=Um, Vm=Vm)


def test_simgcl_trainer_with_the_recorded_sign_pattern_follows_the_reference_run_throughout():
    """The test above may leave the recorded run because `sign(emb)` (SimGCL.py:35) is discontinuous at 0 -- the CPU restatement names
    the flip: step 10, view 2, layer 2, row 805, column 1, an entry of -2.1e-6 in a row of magnitude 0.07
    (tests/test_oracle_tf_golden.py).  Here the excuse is removed: the noise fed to the kernels carries the sign every perturbation of
    the reference's own run used (recorded from its `tf.sign` ops by the generator; include/qrec_hip.h, qrec_perturb_rows), and the
    HIP trainer is held to the run at 1e-5 on the three losses of ALL twelve steps and on the first-step gradients.
    The trained tables: the reference's own float32 run sits 7.7e-6 (users) / 3.2e-5 (items) / 3.3e-5, 2.9e-5 (main embeddings) from
    the SAME run, same signs, in float64 (tf_f64_yardstick.npz) -- InfoNCE's gradient is a difference of two softmax-weighted sums of
    ~900 rows, its small coordinates are cancellation noise in ANY float32 summation order, and Adam turns each into a step of ~lr.
    Every table is held to FLOOR_FACTOR x the reference's own distance from exact arithmetic (kind "floor": two independent float32
    roundings of the same computation are sqrt(2) apart in expectation), and the HIP result must be as close to exact arithmetic as
    the reference's run is (same factor).  Measured (round 6, bit-reproducible): user table 1.5e-5 = 1.9 floors."""
    _parity_twice("SimGCL under the recorded sign pattern", _run_simgcl_recorded)


FLOOR_FACTOR = 2.5


def floor_check(what, got, ref, name, key):
    got, ref = np.asarray(got), np.asarray(ref)
    f64 = YARD[f"{name}/{key}"].reshape(ref.shape)
    floor = rel_err(ref, f64)
    check(f"{what}: the reference's float32 run vs its float64 run (the floor)", floor, 1.0, kind="info")
    check(f"{what}: vs the reference's run, in units of the floor", rel_err(got, ref) / floor, FLOOR_FACTOR, kind="floor")
    check(f"{what}: HIP vs the float64 run, in units of the floor", rel_err(got, f64) / floor, FLOOR_FACTOR, kind="floor")
    check(f"{what}: vs the reference's run (absolute, recorded)", rel_err(got, ref), 1.0, kind="info")


def _run_simgcl_recorded():
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
        if k == 0:
            gU, gV = tr.gradients()
            grad_check(gU, z["grad0_" + names["U"]], "SimGCL dU, step 0, recorded signs"); grad_check(gV, z["grad0_" + names["V"]], "SimGCL dV, step 0, recorded signs")
    U, V = tr.ego_embeddings()
    name = "tf_simgcl_filmtrust"
    floor_check("SimGCL user table after 12 steps under the recorded sign pattern", U, z["final_" + names["U"]], name, "final_" + names["U"])
    floor_check("SimGCL item table after 12 steps under the recorded sign pattern", V, z["final_" + names["V"]], name, "final_" + names["V"])
    Um, Vm = tr.main_embeddings()
    floor_check("SimGCL main user embeddings under the recorded sign pattern", Um, z["score_U"], name, "score_U")
    floor_check("SimGCL main item embeddings under the recorded sign pattern", Vm, z["score_V"], name, "score_V")
    return dict(U=U, V=V, Um=Um, Vm=Vm)


# ---------------------------------------------------------------------------------------------------------------------
# SGL / BUIR / SEPT / MHCN trainers against their reference runs
# ---------------------------------------------------------------------------------------------------------------------


def _csr(a, n):
    import scipy.sparse as sp
    return sp.csr_matrix((a[2], a[1], a[0]), shape=(n, n))


@pytest.mark.parametrize("name", ["tf_sgl_filmtrust", "tf_sgl_rw_filmtrust", "tf_sgl_nd_filmtrust"])
def test_sgl_trainer_follows_the_reference_run(name):
    _parity_twice(name, lambda: _run_sgl(name))


def _run_sgl(name):
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
    table_check(f"SGL aug {aug} user table after the run", U, z["final_U"], name, "final_U")
    table_check(f"SGL aug {aug} item table after the run", V, z["final_V"], name, "final_V")
    Um, Vm = tr.main_embeddings()
    table_check(f"SGL aug {aug} scoring users", Um, z["score_U"], name, "score_U")
    table_check(f"SGL aug {aug} scoring items", Vm, z["score_V"], name, "score_V")
    return dict(U=U, V=V, Um=Um, Vm=Vm)


def test_buir_trainer_follows_the_reference_run():
    _parity_twice("BUIR", _run_buir)


def _run_buir():
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
    name = "tf_buir_filmtrust"
    On, Tg = tr.online_tables(), tr.target_tables()
    table_check("BUIR online user table after the run", On[:nu], z["final_U"], name, "final_U")
    table_check("BUIR online item table after the run", On[nu:], z["final_V"], name, "final_V")
    table_check("BUIR target user table after the run", Tg[:nu], z["final_t_U"], name, "final_t_U")
    table_check("BUIR target item table after the run", Tg[nu:], z["final_t_V"], name, "final_t_V")
    Wg, bg = tr.weights()
    table_check("BUIR online_mat after the run", Wg, z["final_online_mat"], name, "final_online_mat")
    table_check("BUIR online_bias after the run", bg, z["final_online_bias"].ravel(), name, "final_online_bias")
    adj = joint_norm_adjacency(nu, ni, z["train_uid"], z["train_iid"])
    out = dict(On=On, Tg=Tg, W=Wg, b=bg)
    for g, key in zip(tr.final_tables(adj), ("q_user", "q_item", "o_user", "o_item")):
        check("rel_err(g, z[key])", rel_err(g, z[key]), 1e-5, ctx=key)
        out[key] = g
    return out


def test_sept_trainer_follows_the_reference_run():
    _parity_twice("SEPT", _run_sept)


def _run_sept():
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
    # pick another neighbour here than in the reference's run.  Rounds 2-5 (float atomics): WHICH run met a near-tie changed from launch to
    # launch (every joint step <= 2e-7 in most runs, one run with one step at 1.6e-5).  With ordered reductions the run is the same run
    # every time (asserted by the caller, bit for bit), so the statement is plain: every joint step at 1e-5.
    assert len(ssl_err) >= 6
    check("SEPT ssl loss vs reference run (worst joint step)", max(ssl_err), 1e-5, inclusive=True)
    U, V = tr.variables()
    table_check("SEPT user table after 18 steps", U, z["final_U"], "tf_sept_filmtrust", "final_U")
    table_check("SEPT item table after 18 steps", V, z["final_V"], "tf_sept_filmtrust", "final_V")
    Ur, Vr = tr.rec_embeddings()
    table_check("SEPT scoring users", Ur, z["score_U"], "tf_sept_filmtrust", "score_U")
    table_check("SEPT scoring items", Vr, z["score_V"], "tf_sept_filmtrust", "score_V")
    return dict(U=U, V=V, Ur=Ur, Vr=Vr)


def test_mhcn_trainer_follows_the_reference_run():
    _parity_twice("MHCN", _run_mhcn)


def _run_mhcn():
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
    name = "tf_mhcn_filmtrust"
    for a, b in key.items():
        table_check(f"MHCN {a} after the run", np.asarray(got[a]).reshape(z["final_" + b].shape), z["final_" + b], name, "final_" + b)
    table_check("MHCN user table after the run", got["U"], z["final_U"], name, "final_U")
    table_check("MHCN item table after the run", got["V"], z["final_V"], name, "final_V")
    Ud, Vd = tr.final_embeddings()
    table_check("MHCN scoring users", Ud, z["score_U"], name, "score_U")
    table_check("MHCN scoring items", Vd, z["score_V"], name, "score_V")
    out = {a: np.asarray(got[a]) for a in key}
    out.update(U=got["U"], V=got["V"], Ud=Ud, Vd=Vd)
    return out
