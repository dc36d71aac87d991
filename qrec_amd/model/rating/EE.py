"""EE (Euclidean embedding) behind the reference's class name (model/rating/EE.py:4-34,80-95): rating = global mean +
biases - |P[u] - Q[i]|^2, per-rating SGD in ``trainingData`` order through the order-exact kernel (fp64); like SVD,
the reference ignores the convergence test and always runs ``num.max.epoch`` epochs (EE.py:34)."""
from __future__ import annotations

import numpy as np

from ... import capi
from ...base.iterativeRecommender import IterativeRecommender
from ...engine import DeviceTables, MfSgd


class EE(IterativeRecommender):
    def __init__(self, conf, trainingSet=None, testSet=None, fold="[1]"):
        super().__init__(conf, trainingSet, testSet, fold)

    def initModel(self):
        super().initModel()
        self.Bu = np.random.rand(self.data.trainingSize()[0]) / 10      # EE.py:10-11
        self.Bi = np.random.rand(self.data.trainingSize()[1]) / 10

    def trainModel(self):
        tables = DeviceTables(self.P, self.Q, np.float64)
        sgd = MfSgd(tables, self.data.elemCount(), capi.MF_EE, self.Bu, self.Bi)
        epoch = 0
        while epoch < self.maxEpoch:
            u, i, r = self.data.training_arrays()
            self.loss = sgd.epoch(u, i, r, self.lRate, self.regU, self.regI, self.regB, self.data.globalMean)
            _, _, sbu, sbi = sgd.sumsq_terms()
            self.loss += self.regB * sbu + self.regB * sbi                  # EE.py:32
            epoch += 1
            self.P, self.Q = tables.download(np.float64)
            self.Bu, self.Bi = sgd.biases()
            self.isConverged(epoch)                                        # result ignored, as in the reference

    def predictForRating(self, u, i):
        if self.data.containsUser(u) and self.data.containsItem(i):
            u, i = self.data.user[u], self.data.item[i]
            diff = self.P[u] - self.Q[i]
            return self.data.globalMean + self.Bi[i] + self.Bu[u] - diff.dot(diff)
        return self.data.globalMean

    def predictForRanking(self, u):
        if self.data.containsUser(u):
            u = self.data.user[u]
            return ((self.Q - self.P[u]) * (self.Q - self.P[u])).sum(axis=1) + self.Bi + self.Bu[u] + self.data.globalMean
        return [self.data.globalMean] * self.num_items

    def rank_all_test_users(self, N):
        from ...base.recommender import Recommender
        return Recommender.rank_all_test_users(self, N)      # scores are not an inner product: the generic host loop

    def rank_measure_all_test_users(self, top, N):
        return None
