"""SGL behind the reference's class name and hooks (model/ranking/SGL.py:10-293): LightGCN plus two views
over augmented sub-graphs (node dropout / edge dropout / random walk, re-drawn every epoch) contrasted with
InfoNCE over the batch's merged unique users and items; evaluated every epoch, best epoch kept.

Exact mode (default): the sub-graphs come from the CPython ``random`` stream, replayed word for word, and are
rebuilt as CSR + launch plan on the host (the reference's scipy arithmetic, bit-identical).  Throughput mode
(``qrec.mode=throughput``): nothing of an epoch runs on the host -- the sub-graphs are drawn on the device as
value arrays over the full graph's plan (qrec_amd.graph.SubgraphSampler, csrc/augment.hip), the batch stream
and each batch's unique rows likewise; same distributions, the throughput mode's Philox stream."""
from __future__ import annotations

import os
import random

import numpy as np

from ... import capi
from ...base.graphRecommender import GraphRecommender
from ...capi import DeviceBuffer
from ...graph import SGLTrainer, SubgraphSampler, joint_norm_adjacency, sample_subgraph_edges, unique_first_appearance
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
        adj = self.create_joint_sparse_adjaceny()
        self.trainer = self.build_trainer(SGLTrainer, self.user_embeddings, self.item_embeddings, adj,
                                  self.n_layers, self.lRate, self.regU, self.ssl_reg, self.ssl_temp,
                                  max_unique=max(2 * self._step_rows(), 64))
        self.sampler = None
        if self.throughput_mode():
            uid, iid, _ = self.data.training_arrays()
            self.sampler = SubgraphSampler(self.num_users, self.num_items, uid, iid, adj)

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

    # ---- throughput mode: the whole epoch on the device ------------------------------------------------------------------
    SUBGRAPH_STREAM0 = 1 << 32          # Philox stream ids of the sub-graph draws; the batch stream uses 2 * epoch and 2 * epoch + 1

    def _draw_subgraphs_device(self, epoch: int):
        """the epoch's sub-graphs as value arrays over the full graph's plan, in the reference's draw order (SGL.py:233-251): view 1 then
        view 2 (aug 0 / 1), or per layer view 1, view 2 (random walk); node dropout uses two stream ids per draw (users, items)"""
        seed = int(os.environ.get("QREC_SEED", "0"))
        n_draws = 2 if self.aug_type in (0, 1) else 2 * self.n_layers
        if getattr(self, "_sub_vals", None) is None:
            self._sub_vals = [None] * n_draws
        base = self.SUBGRAPH_STREAM0 + 2 * n_draws * epoch
        for k in range(n_draws):
            self._sub_vals[k] = self.sampler.draw(self.aug_type, self.drop_rate, seed, base + 2 * k, out=self._sub_vals[k])
        v = self._sub_vals
        return (v[0], v[1]) if self.aug_type in (0, 1) else (v[0::2], v[1::2])

    def _device_batch_rows(self, d_u, d_i):
        """every batch's merged unique rows (users, then items + n_users; ascending ids) on the device: qrec_unique_per_batch per side,
        the two lists of a batch copied next to each other; the host reads back the counts only"""
        n, step, nu = d_u.shape[0], self._step_rows(), self.num_users
        n_batches = -(-n // step)
        sides = []
        for d_ids, id_range, offset in ((d_u, nu, 0), (d_i, self.num_items, nu)):
            d_rows, d_cnt = DeviceBuffer(max(n, 1), np.int32), DeviceBuffer(max(n_batches, 1), np.int32)
            capi.unique_per_batch(d_ids, n, step, id_range, offset, d_rows, d_cnt)
            sides.append((d_rows, d_cnt.numpy().astype(np.int64)))
        (ru, cu), (ri, ci) = sides
        off = np.concatenate([[0], np.cumsum(cu + ci)])
        d_rows = DeviceBuffer(max(int(off[-1]), 1), np.int32)
        for b in range(n_batches):
            capi.memcpy_d2d(d_rows.ptr + 4 * int(off[b]), ru.ptr + 4 * b * step, 4 * int(cu[b]))
            capi.memcpy_d2d(d_rows.ptr + 4 * int(off[b] + cu[b]), ri.ptr + 4 * b * step, 4 * int(ci[b]))
        return list(range(0, n, step)), d_rows, off, (ru, ri)

    def _train_throughput(self):
        quiet = os.environ.get("QREC_QUIET") == "1"
        tr = self.trainer
        dp = tr.dp = self.data_parallel()
        step = self._step_rows()
        for epoch, (d_u, d_i, d_j) in enumerate(self.iter_epoch_samples_device(self.maxEpoch)):
            tr.set_subgraph_values(*self._draw_subgraphs_device(epoch))
            starts, d_rows, off, keep_alive = self._device_batch_rows(d_u, d_i)
            n_rows = d_u.shape[0]
            for n, s in enumerate(starts):
                B = min(step, n_rows - s)
                tr.train_step_async(d_u.ptr + 4 * s, d_i.ptr + 4 * s, d_j.ptr + 4 * s, B, d_rows.ptr + 4 * int(off[n]), int(off[n + 1] - off[n]),
                                    share=self.step_share(dp, B) if dp else None)
                if not quiet:
                    _, rec_l, ssl_l = tr.losses()
                    print("training:", epoch + 1, "batch", n, "rec_loss:", rec_l, "ssl_loss", ssl_l)
            self.U, self.V = tr.main_embeddings()
            self.ranking_performance(epoch)
            del keep_alive
        self.U, self.V = self.bestU, self.bestV

    def trainModel(self):
        if self.sampler is not None:
            return self._train_throughput()
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
