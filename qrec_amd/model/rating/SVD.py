"""SVD (biased MF) behind the reference's class name (model/rating/SVD.py:4-35,76-90): PMF plus user/item
biases and the global mean; the reference ignores the convergence test here (SVD.py:35) and always runs
``num.max.epoch`` epochs."""
from __future__ import annotations

import numpy as np

from ... import capi
from ...base.iterativeRecommender import IterativeRecommender
from ...engine import DeviceTables, MfSgd


class SVD(IterativeRecommender):
    def __init__(self, conf, trainingSet=None, testSet=None, fold="[1]"):
        super().__init__(conf, trainingSet, testSet, fold)

    def initModel(self):
        super().initModel()
        self.Bu = np.random.rand(self.data.trainingSize()[0]) / 5      # SVD.py:10-11
        self.Bi = np.random.rand(self.data.trainingSize()[1]) / 5

    def trainModel(self):
        tables = DeviceTables(self.P, self.Q, np.float64)
        sgd = MfSgd(tables, self.data.elemCount(), capi.MF_SVD, self.Bu, self.Bi)
        epoch = 0
        while epoch < self.maxEpoch:
            u, i, r = self.data.training_arrays()
            self.loss = sgd.epoch(u, i, r, self.lRate, self.regU, self.regI, self.regB, self.data.globalMean)
            sp, sq, sbu, sbi = sgd.sumsq_terms()
            self.loss += self.regU * sp + self.regI * sq + self.regB * (sbu + sbi)     # SVD.py:32-33
            epoch += 1
            self.P, self.Q = tables.download(np.float64)
            self.Bu, self.Bi = sgd.biases()
            self.isConverged(epoch)                                 # result ignored, as in the reference

    def predictForRating(self, u, i):
        if self.data.containsUser(u) and self.data.containsItem(i):
            u, i = self.data.user[u], self.data.item[i]
            return self.P[u].dot(self.Q[i]) + self.data.globalMean + self.Bi[i] + self.Bu[u]
        return self.data.globalMean

    def predictForRanking(self, u):
        if self.data.containsUser(u):
            u = self.data.getUserId(u)
            return self.Q.dot(self.P[u]) + self.data.globalMean + self.Bi + self.Bu[u]
        return [self.data.globalMean] * self.num_items

    def ranking_tables(self):
        raise NotImplementedError   # scores carry bias terms: the generic host loop ranks this model

    def rank_all_test_users(self, N):
        from ...base.recommender import Recommender
        return Recommender.rank_all_test_users(self, N)

    def rank_measure_all_test_users(self, top, N):
        return None
