"""SVD++ behind the reference's class name (model/rating/SVDPlusPlus.py:5-110): biased MF plus the implicit-feedback
table Y; per-rating SGD in ``trainingData`` order through the order-exact kernel (fp64).  The reference ignores the
convergence test (SVDPlusPlus.py:67) and always runs ``num.max.epoch`` epochs."""
from __future__ import annotations

import numpy as np

from ...base.iterativeRecommender import IterativeRecommender
from ...engine import DeviceTables, SvdppSgd
from ...util import config


class SVDPlusPlus(IterativeRecommender):
    def __init__(self, conf, trainingSet=None, testSet=None, fold="[1]"):
        super().__init__(conf, trainingSet, testSet, fold)

    def readConfiguration(self):
        super().readConfiguration()
        self.regY = float(config.OptionConf(self.config["SVDPlusPlus"])["-y"])

    def printAlgorConfig(self):
        super().printAlgorConfig()
        print("Specified Arguments of", self.config["model.name"] + ":")
        print("regY: %.3f" % self.regY)
        print("=" * 80)

    def initModel(self):
        super().initModel()
        self.Bu = np.random.rand(self.data.trainingSize()[0])            # SVDPlusPlus.py:21-23
        self.Bi = np.random.rand(self.data.trainingSize()[1])
        self.Y = np.random.rand(self.data.trainingSize()[1], self.emb_size)

    def trainModel(self):
        tables = DeviceTables(self.P, self.Q, np.float64)
        sgd = SvdppSgd(tables, self.Y, self.Bu, self.Bi, self.data.rated_csr(), self.data.elemCount())
        epoch = 0
        while epoch < self.maxEpoch:
            u, i, r = self.data.training_arrays()
            self.loss = sgd.epoch(u, i, r, self.lRate, self.regU, self.regI, self.regB, self.regY, self.data.globalMean)
            sp, sq, sy, sbu, sbi = sgd.sumsq_terms()
            self.loss += self.regU * sp + self.regI * sq + self.regY * sy + self.regB * (sbu + sbi)   # SVDPlusPlus.py:64-65
            epoch += 1
            self.P, self.Q = tables.download(np.float64)
            self.Y, self.Bu, self.Bi = sgd.download()
            self.isConverged(epoch)                                        # result ignored, as in the reference

    def _implicit(self, u):
        """sum_j Y[j] / w over the user's rated items (SVDPlusPlus.py:72-81)"""
        items, _ = self.data.userRated(u)
        if not items:
            return None
        total = 0
        for j in items:
            total = total + self.Y[self.data.item[j]]
        return total / len(items)

    def predictForRating(self, u, i):
        pred = 0
        if self.data.containsUser(u) and self.data.containsItem(i):
            imp = self._implicit(u)
            uid, iid = self.data.user[u], self.data.item[i]
            if imp is not None:
                pred += imp.dot(self.Q[iid])
            pred += self.P[uid].dot(self.Q[iid]) + self.data.globalMean + self.Bi[iid] + self.Bu[uid]
        else:
            pred = self.data.globalMean
        return pred

    def predictForRanking(self, u):
        pred = 0
        if self.data.containsUser(u):
            imp = self._implicit(u)
            uid = self.data.user[u]
            if imp is not None:
                pred += self.Q.dot(imp)
            pred += self.Q.dot(self.P[uid]) + self.data.globalMean + self.Bi + self.Bu[uid]
        else:
            pred = [self.data.globalMean] * len(self.data.item)
        return pred

    def rank_all_test_users(self, N):
        from ...base.recommender import Recommender
        return Recommender.rank_all_test_users(self, N)

    def rank_measure_all_test_users(self, top, N):
        return None
