"""SURVEY s8 f-2 on the device (csrc/augment.hip, qrec_amd.graph.SubgraphSampler): the per-epoch sub-graphs of SGL and BUIR
(model/ranking/SGL.py:113-155, BUIR.py:41-65) as value arrays over the full graph's CSR.  The checker is the oracle's restatement
(oracle/tfmodels.py: the Philox permutation's first K entries as the random.sample subset, the reference's scipy normalisation of
the kept rows -- bit-identical to the reference's own, tests/test_oracle_golden.py -- scattered onto the full structure); the
device result is held to it BIT FOR BIT, stream and values."""
import io
import random
from contextlib import redirect_stdout

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import c as O
from oracle import tfmodels as T
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.graph import SpmmPlan, SubgraphSampler, joint_norm_adjacency
from qrec_amd.synth import make_dataset

from helpers import check, conf_from_text, load_golden, pad_cols, rel_err, rows_from_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _device():
    capi.init(0)
    yield


def _words_differ(a, b):
    return int(np.count_nonzero(np.ascontiguousarray(a).view(np.uint32) != np.ascontiguousarray(b).view(np.uint32)))


def _filmtrust():
    z = load_golden("bpr_filmtrust")[1]
    meta = load_golden("bpr_filmtrust")[0]
    return meta["n_users"], meta["n_items"], z["train_uid"].astype(np.int32), z["train_iid"].astype(np.int32)


def _synthetic_with_duplicates():
    rng = np.random.default_rng(9)
    nu, ni, E = 211, 97, 3000                                              # 3000 rows over 20 k cells: many duplicated (u, i) rows
    return nu, ni, rng.integers(0, nu, E).astype(np.int32), rng.integers(0, ni, E).astype(np.int32)


@pytest.mark.parametrize("data", ["filmtrust", "duplicates", "small"])
@pytest.mark.parametrize("aug,rate", [(0, 0.1), (1, 0.1), (2, 0.1), (1, 0.5), (0, 0.5), (1, 0.999), (1, 0.0)])
def test_device_subgraph_values_equal_the_oracle_bit_for_bit(data, aug, rate):
    if data == "filmtrust":
        nu, ni, uid, iid = _filmtrust()
    elif data == "duplicates":
        nu, ni, uid, iid = _synthetic_with_duplicates()
    else:
        d = make_dataset("small"); nu, ni, uid, iid = d["n_users"], d["n_items"], d["train_u"].astype(np.int32), d["train_i"].astype(np.int32)
    adj = joint_norm_adjacency(nu, ni, uid, iid)
    smp = SubgraphSampler(nu, ni, uid, iid, adj)
    for seed, stream_id in ((7, (1 << 32) + 4), (2 ** 40 + 3, 11)):
        got = smp.draw(aug, rate, seed, stream_id).numpy()[:adj[1].size]
        kept = T.philox_subgraph_rows(nu, ni, uid, iid, aug, rate, seed, stream_id)
        want = T.subgraph_values_on_full_structure(nu, ni, uid, iid, kept)
        if aug != 0 and rate > 0:
            assert kept.size == int(uid.size * (1 - rate))                  # random.sample's exact subset size (SGL.py:128)
        check(f"device sub-graph values vs the oracle (aug {aug}, rate {rate}, {data}): 32-bit words that differ", _words_differ(got, want), 0, inclusive=True)
    if rate == 0.0:
        check("rate 0: the full graph's own values, bit for bit", _words_differ(got, adj[2]), 0, inclusive=True)


def test_device_permutation_stream_equals_the_oracle():
    for n, seed, sid in ((1, 0, 0), (1000, 5, 3), (35497, 2 ** 63 + 11, (1 << 32) + 7), (1 << 18, 9, 2 ** 40)):
        ws = DB(capi.random_permutations_scratch_bytes(n, 1), np.uint8); p = DB(n, np.int32)
        capi.random_permutations(n, 1, seed, sid, ws, p, None)
        assert np.array_equal(p.numpy(), O.philox_permutation(n, seed, sid)), (n, seed, sid)


@pytest.mark.parametrize("ld,dim", [(64, 64), (32, 20)])
def test_spmm_over_a_value_array_equals_spmm_over_the_compacted_subgraph(ld, dim):
    """the full graph's plan with a sub-graph's value array (dropped entries 0) against the host-built plan of the compacted
    sub-graph: rows that are not split into segments add the same terms in the same order (+0 terms change nothing): bit-identical;
    split rows (seg_len 7 forces many) agree to fp32 rounding"""
    d = make_dataset("small"); nu, ni = d["n_users"], d["n_items"]
    uid, iid = d["train_u"].astype(np.int32), d["train_i"].astype(np.int32)
    adj = joint_norm_adjacency(nu, ni, uid, iid)
    n = nu + ni
    smp = SubgraphSampler(nu, ni, uid, iid, adj)
    vals = smp.draw(1, 0.3, 5, 77)
    kept = T.philox_subgraph_rows(nu, ni, uid, iid, 1, 0.3, 5, 77)
    sub = joint_norm_adjacency(nu, ni, uid[kept], iid[kept])
    X = np.random.default_rng(1).standard_normal((n, dim)).astype(np.float32)
    dX = DB.from_numpy(pad_cols(X, ld))
    for seg_len in (128, 7):
        full_plan = SpmmPlan(adj[0], adj[1], adj[2], ld, seg_len=seg_len, split_row=nu)
        sub_plan = SpmmPlan(sub[0], sub[1], sub[2], ld, seg_len=seg_len, split_row=nu)
        Y1, Y2 = DB.zeros((n, ld), np.float32), DB.zeros((n, ld), np.float32)
        capi.spmm_csr(full_plan.with_values(vals), dX, Y1, ld)
        capi.spmm_csr(sub_plan, dX, Y2, ld)
        a, b = Y1.numpy()[:, :dim], Y2.numpy()[:, :dim]
        check(f"SpMM over the value array vs over the compacted sub-graph (seg_len {seg_len})", rel_err(a, b), 1e-6)
        short = np.diff(adj[0]) <= seg_len                                  # rows neither plan splits
        check(f"... rows that are not split: words that differ (seg_len {seg_len})", _words_differ(a[short], b[short]), 0, inclusive=True)
        ref = sp.csr_matrix((sub[2], sub[1], sub[0]), shape=(n, n)).dot(X)
        check(f"... vs scipy on the compacted sub-graph (seg_len {seg_len})", rel_err(a, ref), 1e-6)


def _class_conf(name, extra):
    meta, z = load_golden("bpr_filmtrust")
    train, test = rows_from_golden(z)
    conf = conf_from_text(meta["conf"])
    conf["model.name"] = name
    conf["num.factors"] = "16"; conf["num.max.epoch"] = "3"; conf["batch_size"] = "2048"; conf["learnRate"] = "-init 0.001 -max 1"
    conf["reg.lambda"] = "-u 0.001 -i 0.001 -b 0.2 -s 0.2"; conf["item.ranking"] = "on -topN 20"
    for k, v in extra.items():
        conf[k] = v
    return conf, train, test


@pytest.mark.parametrize("aug", [0, 1, 2])
def test_sgl_class_in_throughput_mode_trains_without_touching_the_host_generators(aug, monkeypatch):
    from qrec_amd.model.ranking.SGL import SGL
    conf, train, test = _class_conf("SGL", {"SGL": f"-n_layer 2 -lambda 0.1 -droprate 0.1 -augtype {aug} -temp 0.2"})
    monkeypatch.setenv("QREC_MODE", "throughput"); monkeypatch.setenv("QREC_SEED", "4")
    random.seed(6); np.random.seed(6)
    buf = io.StringIO()
    with redirect_stdout(buf):
        m = SGL(conf, train, test)
        m.readConfiguration(); m.initModel()
        state = random.getstate()
        m.trainModel()
    assert m.sampler is not None and m.trainer.ows is None                    # device augmentation, float atomics: the throughput mode
    assert random.getstate() == state                                          # no CPython draws
    out = buf.getvalue()
    rec = [float(l.split("rec_loss:")[1].split()[0]) for l in out.splitlines() if "rec_loss:" in l]
    ssl = [float(l.split("ssl_loss")[1].split()[0]) for l in out.splitlines() if "ssl_loss" in l]
    assert len(rec) == 3 * 16 and np.isfinite(rec).all() and np.isfinite(ssl).all() and rec[-1] < rec[0]
    assert out.count("Quick Ranking Performance") == 3
    # the sub-graphs the trainer multiplies with are the oracle's, for the stream ids the class documents
    n_draws = 2 if aug in (0, 1) else 4
    uid, iid, _ = m.data.training_arrays()
    base = SGL.SUBGRAPH_STREAM0 + 2 * n_draws * 2                               # the last (third) epoch's draws are still in the buffers
    for k in range(n_draws):
        kept = T.philox_subgraph_rows(m.num_users, m.num_items, uid, iid, aug, 0.1, 4, base + 2 * k)
        want = T.subgraph_values_on_full_structure(m.num_users, m.num_items, uid, iid, kept)
        check(f"SGL class, throughput mode, aug {aug}: sub-graph {k} of the last epoch vs the oracle, words that differ",
              _words_differ(m._sub_vals[k].numpy()[:want.size], want), 0, inclusive=True)


def test_sgl_class_throughput_mode_trains_like_the_exact_mode(monkeypatch):
    """Same conf, same initial tables; the exact mode replays CPython's stream (sub-graphs rebuilt on the host with the reference's scipy
    arithmetic), the throughput mode draws sub-graphs, batches and unique rows on the device.  Different random streams, the same
    distributions -- so the TRAINING CURVES must agree in the mean: the recommendation loss and the contrastive loss averaged over the last
    epoch's batches, mean over 16 streams per mode, inside 1 %.  (Recall@20 on this 1,500-user set spreads 0.07 (sd) from stream to stream
    in either mode -- 24 streams per mode left a standard error of 0.02 on the difference, round 6 -- so it is recorded, not asserted: a
    wrong augmentation (a different keep rate, un-normalised values) moves the contrastive loss by far more than 1 %.)"""
    from qrec_amd.model.ranking.SGL import SGL
    conf, train, test = _class_conf("SGL", {"SGL": "-n_layer 2 -lambda 0.1 -droprate 0.1 -augtype 1 -temp 0.2", "num.max.epoch": "10",
                                            "learnRate": "-init 0.01 -max 1"})
    n_batches = -(-len(train) // 2048)

    def run(mode, seed):
        monkeypatch.setenv("QREC_MODE", mode); monkeypatch.setenv("QREC_SEED", str(seed))
        random.seed(seed); np.random.seed(3)
        buf = io.StringIO()
        with redirect_stdout(buf):
            m = SGL(conf, train, test)
            measure = m.execute()
        out = buf.getvalue()
        rec = [float(l.split("rec_loss:")[1].split()[0]) for l in out.splitlines() if "rec_loss:" in l][-n_batches:]
        ssl = [float(l.split("ssl_loss")[1].split()[0]) for l in out.splitlines() if "ssl_loss" in l][-n_batches:]
        return np.mean(rec), np.mean(ssl), [float(x.split(":")[1]) for x in measure if ":" in x][1]
    S = 16
    exact = np.array([run("exact", 3 + k) for k in range(S)]); thr = np.array([run("throughput", 1003 + k) for k in range(S)])
    se = np.sqrt(exact.var(0, ddof=1) / S + thr.var(0, ddof=1) / S)
    rel = np.abs(exact.mean(0) - thr.mean(0)) / np.abs(exact.mean(0))
    print("SGL last-epoch rec / ssl loss, Recall@20: exact", exact.mean(0), "throughput", thr.mean(0), "relative gap", rel, "se", se, "sd exact", exact.std(0, ddof=1))
    assert exact[:, 2].mean() > 0.02 and thr[:, 2].mean() > 0.02                      # both modes learned something (random ranking: 0.01)
    check("SGL throughput mode (device augmentation) vs exact mode: last-epoch recommendation loss, relative difference of the means over 16 streams", rel[0], 0.01,
          kind="statistical", ctx=se[0] / exact[:, 0].mean())
    check("SGL throughput mode (device augmentation) vs exact mode: last-epoch contrastive loss, relative difference of the means over 16 streams", rel[1], 0.01,
          kind="statistical", ctx=se[1] / exact[:, 1].mean())
    check("SGL Recall@20: |difference of the means| (recorded; standard error in ctx)", abs(exact[:, 2].mean() - thr[:, 2].mean()), 1.0, kind="info", ctx=se[2])


def test_buir_class_in_throughput_mode_trains_on_device_drawn_subgraphs(monkeypatch):
    from qrec_amd.model.ranking.BUIR import BUIR
    conf, train, test = _class_conf("BUIR", {"BUIR": "-n_layer 2 -tau 0.995 -drop_rate 0.5", "num.factors": "32", "item.ranking": "on -topN 10"})
    monkeypatch.setenv("QREC_MODE", "throughput"); monkeypatch.setenv("QREC_SEED", "9")
    random.seed(5); np.random.seed(5)
    buf = io.StringIO()
    with redirect_stdout(buf):
        m = BUIR(conf, train, test)
        state = random.getstate()
        measure = m.execute()
    assert m.sampler is not None and random.getstate() == state              # nothing drawn from CPython's generator
    losses = [float(l.rsplit("loss:", 1)[1]) for l in buf.getvalue().splitlines() if "training:" in l]
    assert len(losses) == 3 * -(-len(train) // 2048) and losses[-1] < losses[0] and all(np.isfinite(losses))
    assert any(x.startswith("Recall:") for x in measure)
    uid, iid, _ = m.data.training_arrays()
    kept = T.philox_subgraph_rows(m.num_users, m.num_items, uid, iid, 1, 0.5, 9, (1 << 32) + 2 * 2 + 1)      # sub-graph T of the last epoch
    want = T.subgraph_values_on_full_structure(m.num_users, m.num_items, uid, iid, kept)
    check("BUIR class, throughput mode: the target encoder's sub-graph of the last epoch vs the oracle, words that differ",
          _words_differ(m.trainer.plan_t.values.numpy()[:want.size], want), 0, inclusive=True)


def test_subgraph_values_refuses_bad_arguments_and_handles_degenerate_draws():
    """include/qrec_hip.h, qrec_subgraph_values: null / contradictory arguments are refused with a message and nothing is written;
    an empty keep list and "every node dropped" give the all-zero value array (the reference's empty csr_matrix: SGL.py:131-139 with
    no kept row), a node draw that drops nobody gives the full graph's own values."""
    nu, ni, uid, iid = _synthetic_with_duplicates()
    adj = joint_norm_adjacency(nu, ni, uid, iid)
    smp = SubgraphSampler(nu, ni, uid, iid, adj)
    lib = capi.load()
    out = DB.from_numpy(np.full(smp.nnz, 7.0, np.float32))
    common = lambda: (smp.d_u.ptr, smp.d_i.ptr, smp.d_pos_ui.ptr, smp.d_pos_iu.ptr, smp.n_edges, smp.nu, smp.ni)
    tail = lambda: (smp.d_row_of.ptr, smp.d_indices.ptr, smp.nnz, smp.d_dinv.ptr, smp.max_deg, smp.d_cnt.ptr, smp.d_deg.ptr)
    keep = DB.from_numpy(np.arange(smp.n_edges, dtype=np.int32)); ids = DB.from_numpy(np.arange(max(nu, ni), dtype=np.int32))

    def call(keep_ptr, n_keep, du, ndu, di, ndi, flags, values=out.ptr, u_ptr=None):
        c = list(common())
        if u_ptr is not None:
            c[0] = u_ptr
        return lib.qrec_subgraph_values(*c, keep_ptr, n_keep, du, ndu, di, ndi, *tail(), flags, values, None)

    assert call(None, 0, None, 0, None, 0, None, values=None) < 0 and "null" in lib.qrec_last_error().decode()
    assert call(keep.ptr, smp.n_edges + 1, None, 0, None, 0, None) < 0 and "keep" in lib.qrec_last_error().decode()
    assert call(keep.ptr, -1, None, 0, None, 0, None) < 0
    assert call(keep.ptr, 5, ids.ptr, 3, ids.ptr, 3, smp.d_flags.ptr) < 0 and "not both" in lib.qrec_last_error().decode()
    assert call(None, 0, ids.ptr, 3, ids.ptr, 3, None) < 0 and "flag" in lib.qrec_last_error().decode()           # node dropout without its scratch
    assert call(None, 0, ids.ptr, nu + 1, ids.ptr, 0, smp.d_flags.ptr) < 0                                         # more users dropped than there are
    assert call(None, 0, ids.ptr, 3, None, 2, smp.d_flags.ptr) < 0                                                 # a count without its list
    check("refused calls wrote nothing: entries of the value array that changed", int(np.count_nonzero(out.numpy() != 7.0)), 0, inclusive=True)
    # degenerate draws
    assert call(keep.ptr, 0, None, 0, None, 0, None) == 0
    check("empty keep list: non-zero values", int(np.count_nonzero(out.numpy())), 0, inclusive=True)
    out.upload(np.full(smp.nnz, 7.0, np.float32))
    assert call(None, 0, ids.ptr, nu, ids.ptr, ni, smp.d_flags.ptr) == 0
    check("every node dropped: non-zero values", int(np.count_nonzero(out.numpy())), 0, inclusive=True)
    assert call(None, 0, ids.ptr, 0, ids.ptr, 0, smp.d_flags.ptr) == 0
    check("node draw that drops nobody: words that differ from the full graph's values", _words_differ(out.numpy()[:adj[2].size], adj[2]), 0, inclusive=True)
    assert call(keep.ptr, smp.n_edges, None, 0, None, 0, None) == 0
    check("every row kept: words that differ from the full graph's values", _words_differ(out.numpy()[:adj[2].size], adj[2]), 0, inclusive=True)
