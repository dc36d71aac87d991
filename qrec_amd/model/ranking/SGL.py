"""SGL behind the reference's class name and hooks (model/ranking/SGL.py:10-293): LightGCN plus two views
over augmented sub-graphs (node dropout / edge dropout / random walk, re-drawn every epoch from the CPython
``random`` stream) contrasted with InfoNCE over the batch's merged unique users and items; evaluated
every epoch, best epoch kept."""
from __future__ import annotations

import os
import random

import numpy as np

from ... import capi
from ...base.graphRecommender import GraphRecommender
from ...capi import DeviceBuffer
from ...graph import SGLTrainer, joint_norm_adjacency, sample_subgraph_edges, unique_first_appearance
from ...util import config


class SGL(GraphRecommender):
    def __init__(self, conf, trainingSet=None, testSet=None, fold="[1]"):
        super().__init__(conf, trainingSet, testSet, fold)

    def readConfiguration(self):
        super().readConfiguration()
        args = config.OptionConf(self.config["SGL"])
        self.ssl_reg = float(args["-lambda"])
        self.drop_rate = float(args["-droprate"])
        self.aug_type = int(args["-augtype"])
        self.ssl_temp = float(args["-temp"])
        self.n_layers = int(args["-n_layer"])

    def initModel(self):
        super().initModel()
        self.trainer = self.build_trainer(SGLTrainer, self.user_embeddings, self.item_embeddings, self.create_joint_sparse_adjaceny(),
                                  self.n_layers, self.lRate, self.regU, self.ssl_reg, self.ssl_temp,
                                  max_unique=max(2 * self._step_rows(), 64))

    def _step_rows(self) -> int:
        """rows of the batch stream one training step covers: batch_size, times the world size in a multi-GPU run"""
        dp = self.data_parallel()
        return self.batch_size * (dp.world if dp else 1)

    def _create_adj_mat(self, is_subgraph=False, aug_type=0):
        """CSR triple of the (sub-)graph's normalized adjacency (SGL.py:113-155); sub-graphs consume the
        CPython generator exactly as the reference's random.sample calls do."""
        uid, iid, _ = self.data.training_arrays()
        if is_subgraph and aug_type in (0, 1, 2) and self.drop_rate > 0:
            state = random.getstate()
            words = capi.state_from_python(state)
            uid, iid = sample_subgraph_edges(words, uid, iid, self.num_users, self.num_items, aug_type, self.drop_rate)
            random.setstate(capi.state_to_python(words, state[2]))
        return joint_norm_adjacency(self.num_users, self.num_items, uid, iid)

    def _draw_epoch(self):
        """one epoch's host-side randomness in the reference's order (SGL.py:233-251): the sub-graphs
        (view 1 then view 2; per layer for random walk), then shuffle + negatives."""
        if self.aug_type in (0, 1):
            subs = (self._create_adj_mat(True, self.aug_type), self._create_adj_mat(True, self.aug_type))
        else:
            s1, s2 = [], []
            for _ in range(self.n_layers):
                s1.append(self._create_adj_mat(True, self.aug_type)); s2.append(self._create_adj_mat(True, self.aug_type))
            subs = (s1, s2)
        u, i, j = self.sample_epoch_pairwise()
        # tf.unique of every batch (merged user + item rows, SGL.py calc_ssl_loss_v3), also on the sampler thread
        nu = self.num_users
        step = self._step_rows()
        starts = list(range(0, u.size, step))
        rows = [np.concatenate([unique_first_appearance(u[s:s + step]),
                                unique_first_appearance(i[s:s + step]) + nu]).astype(np.int32) for s in starts]
        off = np.concatenate([[0], np.cumsum([r.size for r in rows])])
        return subs, (u, i, j, starts, np.concatenate(rows), off)

    def saveModel(self):
        self.bestU, self.bestV = self.U, self.V

    def trainModel(self):
        quiet = os.environ.get("QREC_QUIET") == "1"
        tr, nu = self.trainer, self.num_users
        dp = tr.dp = self.data_parallel()
        step = self._step_rows()
        for epoch, (subs, (u, i, j, starts, rows, off)) in enumerate(self.iter_epoch_samples(self.maxEpoch, self._draw_epoch)):
            tr.set_subgraphs(*subs)
            d_u, d_i, d_j = DeviceBuffer.from_numpy(u), DeviceBuffer.from_numpy(i), DeviceBuffer.from_numpy(j)
            d_rows = DeviceBuffer.from_numpy(rows)
            for n, s in enumerate(starts):
                B = min(step, u.size - s)
                tr.train_step_async(d_u.ptr + 4 * s, d_i.ptr + 4 * s, d_j.ptr + 4 * s, B, d_rows.ptr + 4 * int(off[n]), int(off[n + 1] - off[n]),
                                    share=self.step_share(dp, B) if dp else None)
                if not quiet:
                    _, rec_l, ssl_l = tr.losses()
                    print("training:", epoch + 1, "batch", n, "rec_loss:", rec_l, "ssl_loss", ssl_l)
            self.U, self.V = tr.main_embeddings()
            self.ranking_performance(epoch)
        self.U, self.V = self.bestU, self.bestV

    def predictForRanking(self, u):
        if self.data.containsUser(u):
            return self.V.dot(self.U[self.data.getUserId(u)])
        return [self.data.globalMean] * self.num_items
