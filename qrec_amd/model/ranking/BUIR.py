"""BUIR behind the reference's class name and hooks (model/ranking/BUIR.py:13-177): a LightGCN online encoder and a
momentum target encoder over two edge-dropped sub-graphs re-drawn every epoch from the CPython ``random`` stream,
trained without negatives (the batch's negative draws are still made -- ``next_batch_pairwise`` is the reference's
batch source -- and ignored)."""
from __future__ import annotations

import os
import random

import numpy as np

from ... import capi
from ...base.deepRecommender import DeepRecommender
from ...capi import DeviceBuffer
from ...graph import BUIRTrainer, SubgraphSampler, joint_norm_adjacency, sample_subgraph_edges
from ...util import config


def _xavier(shape, rng):
    """tf.contrib.layers.xavier_initializer(): U(-l, l), l = sqrt(6 / (fan_in + fan_out)) (TF's own stream is not
    reproducible; parity tests inject the initial values)"""
    lim = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-lim, lim, shape).astype(np.float32)


class BUIR(DeepRecommender):
    def __init__(self, conf, trainingSet=None, testSet=None, fold="[1]"):
        super().__init__(conf, trainingSet, testSet, fold)

    def readConfiguration(self):
        super().readConfiguration()
        args = config.OptionConf(self.config["BUIR"])
        self.n_layers = int(args["-n_layer"])
        self.tau = float(args["-tau"])
        self.drop_rate = float(args["-drop_rate"])

    def initModel(self):
        super().initModel()
        rng = np.random.default_rng(np.random.randint(0, 2 ** 31 - 1))     # seeded from the global numpy stream
        d = self.emb_size
        self.online_mat, self.online_bias = _xavier((d, d), rng), _xavier((1, d), rng)          # BUIR.py:81-82
        U0, V0 = _xavier((self.num_users, d), rng), _xavier((self.num_items, d), rng)            # BUIR.py:83-84
        self.trainer = self.build_trainer(BUIRTrainer, U0, V0, self.online_mat, self.online_bias, self.n_layers, self.lRate, self.tau)
        self.sampler = None
        if self.throughput_mode():
            # the epoch's two edge-dropped sub-graphs are drawn on the device as value arrays over the FULL graph's plan
            # (qrec_amd.graph.SubgraphSampler, csrc/augment.hip); the batch stream too (base/deepRecommender.py)
            uid, iid, _ = self.data.training_arrays()
            adj = self.get_adj_mat()
            self.trainer.set_full_graph(adj)
            self.sampler = SubgraphSampler(self.num_users, self.num_items, uid, iid, adj)

    def get_adj_mat(self, is_subgraph=False):
        """CSR triple of the normalized (sub-)graph adjacency (BUIR.py:41-65); a sub-graph keeps
        int(n (1 - drop_rate)) interactions drawn with random.sample from the CPython stream."""
        uid, iid, _ = self.data.training_arrays()
        if is_subgraph and self.drop_rate > 0:
            state = random.getstate()
            words = capi.state_from_python(state)
            uid, iid = sample_subgraph_edges(words, uid, iid, self.num_users, self.num_items, 1, self.drop_rate)
            random.setstate(capi.state_to_python(words, state[2]))
        return joint_norm_adjacency(self.num_users, self.num_items, uid, iid)

    def _draw_epoch(self):
        """one epoch's host-side randomness in the reference's order (BUIR.py:139-147): sub-graph O, sub-graph T,
        then shuffle + one negative per row"""
        subs = (self.get_adj_mat(True), self.get_adj_mat(True))
        return subs, self.sample_epoch_pairwise()

    def _train_throughput(self):
        quiet = os.environ.get("QREC_QUIET") == "1"
        tr = self.trainer
        seed = int(os.environ.get("QREC_SEED", "0"))
        vals = [None, None]
        for epoch, (d_u, d_i, _) in enumerate(self.iter_epoch_samples_device(self.maxEpoch)):
            for k in range(2):          # sub-graph O, then T (BUIR.py:139-147); stream ids apart from the batch stream's 2 * epoch (+ 1)
                vals[k] = self.sampler.draw(1, self.drop_rate, seed, (1 << 32) + 2 * epoch + k, out=vals[k])
            tr.set_subgraph_values(vals[0], vals[1])
            dp = tr.dp = self.data_parallel()
            step = self.batch_size * (dp.world if dp else 1)
            n_rows = d_u.shape[0]
            for n, s in enumerate(range(0, n_rows, step)):
                lo, B = self.step_share(dp, min(step, n_rows - s))
                tr.train_step_async(d_u.ptr + 4 * (s + lo), d_i.ptr + 4 * (s + lo), B)
                if not quiet:
                    print(self.foldInfo, "training:", epoch + 1, "batch", n, "loss:", tr.loss())
        self._final_py_state = random.getstate()      # untouched: no CPython draws in this mode
        self.q_user, self.q_item, self.o_user, self.o_item = tr.final_tables(self.get_adj_mat())

    def trainModel(self):
        if self.sampler is not None:
            return self._train_throughput()
        quiet = os.environ.get("QREC_QUIET") == "1"
        tr = self.trainer
        for epoch, (subs, (u, i, _)) in enumerate(self.iter_epoch_samples(self.maxEpoch, self._draw_epoch)):
            tr.set_subgraphs(*subs)
            d_u, d_i = DeviceBuffer.from_numpy(u), DeviceBuffer.from_numpy(i)
            dp = tr.dp = self.data_parallel()              # one process per GPU: a step = batch_size x world pairs, this rank's share
            step = self.batch_size * (dp.world if dp else 1)
            for n, s in enumerate(range(0, u.size, step)):
                lo, B = self.step_share(dp, min(step, u.size - s))
                tr.train_step_async(d_u.ptr + 4 * (s + lo), d_i.ptr + 4 * (s + lo), B)
                if not quiet:
                    print(self.foldInfo, "training:", epoch + 1, "batch", n, "loss:", tr.loss())
        self._final_py_state = random.getstate()      # where the reference's generator stands after training
        self.q_user, self.q_item, self.o_user, self.o_item = tr.final_tables(self.get_adj_mat())

    def predictForRanking(self, u):
        if self.data.containsUser(u):
            u = self.data.getUserId(u)
            return self.q_item.dot(self.o_user[u]) + self.o_item.dot(self.q_user[u])
        return [self.data.globalMean] * self.num_items

    def ranking_tables(self):
        """score(u, .) = q_item . o_user[u] + o_item . q_user[u] (BUIR.py:172) = [q_item | o_item] . [o_user[u] | q_user[u]]"""
        return (np.ascontiguousarray(np.concatenate([self.o_user, self.q_user], axis=1)),
                np.ascontiguousarray(np.concatenate([self.q_item, self.o_item], axis=1)))
