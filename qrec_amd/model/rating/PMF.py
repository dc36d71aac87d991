"""PMF behind the reference's class name (model/rating/PMF.py:5-28): per-rating SGD with L2 terms, visited
in ``trainingData`` order (reshuffled every epoch by isConverged), order-exact device kernel in fp64."""
from __future__ import annotations

import numpy as np

from ... import capi
from ...base.iterativeRecommender import IterativeRecommender
from ...engine import DeviceTables, MfSgd


class PMF(IterativeRecommender):
    def __init__(self, conf, trainingSet=None, testSet=None, fold="[1]"):
        super().__init__(conf, trainingSet, testSet, fold)

    def trainModel(self):
        tables = DeviceTables(self.P, self.Q, np.float64)
        sgd = MfSgd(tables, self.data.elemCount(), capi.MF_PMF)
        epoch = 0
        while epoch < self.maxEpoch:
            u, i, r = self.data.training_arrays()
            self.loss = sgd.epoch(u, i, r, self.lRate, self.regU, self.regI)
            sp, sq, _, _ = sgd.sumsq_terms()
            self.loss += self.regU * sp + self.regI * sq           # PMF.py:25
            epoch += 1
            self.P, self.Q = tables.download(np.float64)           # isConverged() prints rating_performance()
            if self.isConverged(epoch):
                break
