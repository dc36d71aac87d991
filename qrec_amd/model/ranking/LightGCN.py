"""LightGCN behind the reference's class name and hooks (model/ranking/LightGCN.py:5-49):
``-n_layer`` propagation layers over the joint adjacency, mean of the layer outputs, batch
BPR loss + batch L2, Adam -- every batch re-propagates the whole graph forward and backward,
so the SpMM kernel is the hot loop."""
from __future__ import annotations

import os

import numpy as np

from ... import capi
from ...base.graphRecommender import GraphRecommender
from ...capi import DeviceBuffer
from ...graph import LightGCNTrainer
from ...util.config import OptionConf


class LightGCN(GraphRecommender):
    def __init__(self, conf, trainingSet=None, testSet=None, fold="[1]"):
        super().__init__(conf, trainingSet, testSet, fold)
        self.n_layers = int(OptionConf(self.config["LightGCN"])["-n_layer"])

    def initModel(self):
        super().initModel()
        dp = self.data_parallel()
        self.row_partitioned = dp is not None and os.environ.get("QREC_GRAPH_DIST", "batch") == "rows"
        if self.row_partitioned:
            # one process per GPU, QREC_GRAPH_DIST=rows: the reference's own batch size, the propagation row-partitioned
            # over the ranks (qrec_amd/graph.py); default is the batch-sharded scheme (dist.BatchParallel)
            from ...graph import RowPartitionedLightGCNTrainer
            self.trainer = self.build_trainer(RowPartitionedLightGCNTrainer, dp.comm, self.user_embeddings, self.item_embeddings,
                                                         self.create_joint_sparse_adjaceny(), self.n_layers, self.lRate, self.regU)
            return
        self.trainer = self.build_trainer(LightGCNTrainer, self.user_embeddings, self.item_embeddings,
                                       self.create_joint_sparse_adjaceny(), self.n_layers, self.lRate, self.regU)

    def trainModel(self):
        quiet = os.environ.get("QREC_QUIET") == "1"
        tr = self.trainer
        dp = self.data_parallel()
        if self.row_partitioned:
            dp = None                                  # every rank takes the whole step; the SpMM rows are what is split
        else:
            tr.dp = dp                                 # one process per GPU: a step = batch_size x world rows, this rank's share
        step_rows = self.batch_size * (dp.world if dp else 1)
        for epoch, (n_rows, d_u, d_i, d_j) in enumerate(self.iter_epoch_device_samples(self.maxEpoch)):     # base/deepRecommender.py:29-52
            for n, s in enumerate(range(0, n_rows, step_rows)):
                lo, B = self.step_share(dp, min(step_rows, n_rows - s))
                tr.train_step_async(d_u.ptr + 4 * (s + lo), d_i.ptr + 4 * (s + lo), d_j.ptr + 4 * (s + lo), B)
                if not quiet:                                        # the reference prints every batch
                    print(self.foldInfo, "training:", epoch + 1, "batch", n, "loss:", tr.loss())
        self.U, self.V = tr.final_embeddings()

    def predictForRanking(self, u):
        if self.data.containsUser(u):
            return self.V.dot(self.U[self.data.getUserId(u)])
        return [self.data.globalMean] * self.num_items
