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
