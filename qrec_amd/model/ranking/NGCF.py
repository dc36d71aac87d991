"""NGCF behind the reference's class name and hooks (model/ranking/NGCF.py:4-71): two
neighbourhood-aggregation layers with d x d weights, LeakyReLU(0.2), message dropout 0.1 while
training, L2-normalised layer outputs concatenated with the ego embeddings (3d wide), batch BPR
loss + batch L2, Adam.  Test-time scores come from the inference graph (no dropout)."""
from __future__ import annotations

import os

import numpy as np

from ...base.graphRecommender import GraphRecommender
from ...capi import DeviceBuffer
from ...graph import NGCFTrainer
from .SimGCL import xavier_uniform


class NGCF(GraphRecommender):
    def __init__(self, conf, trainingSet=None, testSet=None, fold="[1]"):
        super().__init__(conf, trainingSet, testSet, fold)

    def initModel(self):
        super().initModel()
        d = self.emb_size
        self.n_layers = 2                                                   # NGCF.py:19
        self.weights = [[xavier_uniform((d, d)), xavier_uniform((d, d))] for _ in range(self.n_layers)]   # W_k_1, W_k_2
        dp = self.data_parallel()
        self.row_partitioned = dp is not None and os.environ.get("QREC_GRAPH_DIST", "batch") == "rows"
        if self.row_partitioned:
            # one process per GPU, QREC_GRAPH_DIST=rows: the reference's own batch size, every node table row-partitioned over
            # the ranks, the d x d weights replicated and their gradients all-reduced (qrec_amd/graph.py); the default is the
            # batch-sharded scheme (dist.BatchParallel)
            from ...graph import RowPartitionedNGCFTrainer
            self.trainer = self.build_trainer(RowPartitionedNGCFTrainer, dp.comm, self.user_embeddings, self.item_embeddings, self.weights,
                                                     self.create_joint_sparse_adjaceny(), self.lRate, self.regU,
                                                     seed=int(os.environ.get("QREC_SEED", "0")))
            return
        self.trainer = self.build_trainer(NGCFTrainer, self.user_embeddings, self.item_embeddings, self.weights,
                                   self.create_joint_sparse_adjaceny(), self.lRate, self.regU,
                                   seed=int(os.environ.get("QREC_SEED", "0")))

    def trainModel(self):
        quiet = os.environ.get("QREC_QUIET") == "1"
        tr = self.trainer
        if self.row_partitioned:
            dp = None                                  # every rank takes the whole step; the node tables are what is split
        else:
            dp = tr.dp = self.data_parallel()          # one process per GPU: a step = batch_size x world rows, this rank's share
        step_rows = self.batch_size * (dp.world if dp else 1)
        for epoch, (n_rows, d_u, d_i, d_j) in enumerate(self.iter_epoch_device_samples(self.maxEpoch)):     # base/deepRecommender.py:29-52
            for n, s in enumerate(range(0, n_rows, step_rows)):
                lo, B = self.step_share(dp, min(step_rows, n_rows - s))
                tr.train_step_async(d_u.ptr + 4 * (s + lo), d_i.ptr + 4 * (s + lo), d_j.ptr + 4 * (s + lo), B)
                if not quiet:
                    print("training:", epoch + 1, "batch", n, "loss:", tr.loss())
        # the reference scores with sess.run(self.test, isTraining=0) per user (NGCF.py:65-69);
        # here the inference-graph tables are materialised once and ranked in one batch
        self.U, self.V = tr.inference_embeddings()

    def predictForRanking(self, u):
        if self.data.containsUser(u):
            return self.V.dot(self.U[self.data.getUserId(u)])
        return [self.data.globalMean] * self.num_items
