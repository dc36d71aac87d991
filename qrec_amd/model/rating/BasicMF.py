"""BasicMF behind the reference's class name (model/rating/BasicMF.py:5-26): plain
per-rating SGD, visited in ``trainingData`` order (reshuffled every epoch by
isConverged), run by the order-exact device kernel in fp64."""
from __future__ import annotations

import numpy as np

from ...base.iterativeRecommender import IterativeRecommender
from ...engine import DeviceTables, MfSgd


class BasicMF(IterativeRecommender):
    def __init__(self, conf, trainingSet=None, testSet=None, fold="[1]"):
        super().__init__(conf, trainingSet, testSet, fold)

    def trainModel(self):
        tables = DeviceTables(self.P, self.Q, np.float64)
        sgd = MfSgd(tables, self.data.elemCount())
        epoch = 0
        while epoch < self.maxEpoch:
            u, i, r = self.data.training_arrays()
            self.loss = sgd.epoch(u, i, r, self.lRate)
            epoch += 1
            # isConverged() prints rating_performance() from self.P/self.Q
            self.P, self.Q = tables.download(np.float64)
            if self.isConverged(epoch):
                break
