"""MHCN behind the reference's class name and hooks (model/ranking/MHCN.py:15-240): multi-channel hypergraph convolution
over three motif-induced user-user graphs (social, joint, purchase) and the user-item graph, self-gated channel inputs,
channel attention, and a hierarchical mutual-information loss per channel whose negatives are row / column shuffles.
Needs the ``social`` file (``social.setup``); evaluated every epoch, best epoch kept."""
from __future__ import annotations

import os

import numpy as np

from ...base.graphRecommender import GraphRecommender
from ...base.socialRecommender import SocialRecommender
from ...capi import DeviceBuffer
from ...graph import MHCNTrainer, mhcn_channel_graphs
from ...util import config


def _xavier(shape) -> np.ndarray:
    """tf.contrib.layers.xavier_initializer(): U(+-sqrt(6 / (fan_in + fan_out))), from numpy's global generator"""
    lim = np.sqrt(6.0 / (shape[0] + shape[1]))
    return np.random.uniform(-lim, lim, shape).astype(np.float32)


class MHCN(SocialRecommender, GraphRecommender):
    def __init__(self, conf, trainingSet=None, testSet=None, relation=None, fold="[1]"):
        SocialRecommender.__init__(self, conf, trainingSet, testSet, relation if relation is not None else [], fold)

    def readConfiguration(self):
        super().readConfiguration()
        args = config.OptionConf(self.config["MHCN"])
        self.n_layers = int(args["-n_layer"])
        self.ss_rate = float(args["-ss_rate"])

    def initModel(self):
        super().initModel()
        uid, iid, r = self.data.training_arrays()
        fo, fe = self.relation_ids()
        H, R = mhcn_channel_graphs(self.num_users, self.num_items, uid, iid, r, fo, fe)
        d = self.emb_size
        self.n_channel = 4
        weights = {}
        for k in range(1, self.n_channel + 1):                       # creation order of MHCN.py:101-107
            weights[f"gating{k}"] = _xavier((d, d)); weights[f"gating_bias{k}"] = _xavier((1, d))
            weights[f"sgating{k}"] = _xavier((d, d)); weights[f"sgating_bias{k}"] = _xavier((1, d))
        weights["attention"] = _xavier((1, d)); weights["attention_mat"] = _xavier((d, d))
        self.trainer = self.build_trainer(MHCNTrainer, self.user_embeddings, self.item_embeddings, weights, H, R, self.n_layers, self.lRate, self.regU,
                                   self.ss_rate, seed=int(os.environ.get("QREC_SEED", "0")))

    def saveModel(self):
        self.bestU, self.bestV = self.U, self.V

    def trainModel(self):
        quiet = os.environ.get("QREC_QUIET") == "1"
        tr = self.trainer
        dp = tr.dp = self.data_parallel()              # one process per GPU: a step = batch_size x world rows, this rank's share
        step = self.batch_size * (dp.world if dp else 1)
        for epoch, (u, i, j) in enumerate(self.iter_epoch_samples(self.maxEpoch)):                  # base/deepRecommender.py:29-52
            d_u, d_i, d_j = DeviceBuffer.from_numpy(u), DeviceBuffer.from_numpy(i), DeviceBuffer.from_numpy(j)
            for n, s in enumerate(range(0, u.size, step)):
                lo, B = self.step_share(dp, min(step, u.size - s))
                tr.train_step_async(d_u.ptr + 4 * (s + lo), d_i.ptr + 4 * (s + lo), d_j.ptr + 4 * (s + lo), B)
                if not quiet:
                    print(self.foldInfo, "training:", epoch + 1, "batch", n, "rec loss:", tr.losses()[0])
            self.U, self.V = tr.final_embeddings()
            self.ranking_performance(epoch)
        self.U, self.V = self.bestU, self.bestV

    def predictForRanking(self, u):
        if self.data.containsUser(u):
            return self.V.dot(self.U[self.data.getUserId(u)])
        return [self.data.globalMean] * self.num_items
