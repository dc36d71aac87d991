"""Iterative (SGD) recommenders: conf parsing, fp64 ``P``/``Q`` tables, the bold-driver
learning-rate schedule, the convergence test and the in-training ranking evaluation of the
reference's ``IterativeRecommender`` (base/iterativeRecommender.py:7-185)."""
from __future__ import annotations

import random
import sys
from math import isnan

import numpy as np

from .. import capi
from ..util import config
from ..util.measure import Measure
from .recommender import Recommender


class IterativeRecommender(Recommender):
    def __init__(self, conf, trainingSet, testSet, fold="[1]"):
        super().__init__(conf, trainingSet, testSet, fold)
        self.bestPerformance = []
        self.earlyStop = 0

    def readConfiguration(self):
        super().readConfiguration()
        self.emb_size = int(self.config["num.factors"])
        self.maxEpoch = int(self.config["num.max.epoch"])
        rate = config.OptionConf(self.config["learnRate"])
        self.lRate, self.maxLRate = float(rate["-init"]), float(rate["-max"])
        if self.evalSettings.contains("-tf"):
            self.batch_size = int(self.config["batch_size"])
        reg = config.OptionConf(self.config["reg.lambda"])
        self.regU, self.regI, self.regB = float(reg["-u"]), float(reg["-i"]), float(reg["-b"])

    def printAlgorConfig(self):
        super().printAlgorConfig()
        print("Embedding Dimension:", self.emb_size)
        print("Maximum Epoch:", self.maxEpoch)
        print("Regularization parameter: regU %.3f, regI %.3f, regB %.3f" % (self.regU, self.regI, self.regB))
        print("=" * 80)

    def initModel(self):
        # global numpy RNG, uniform[0,1)/3, float64 (base/iterativeRecommender.py:37-38)
        self.P = np.random.rand(len(self.data.user), self.emb_size) / 3
        self.Q = np.random.rand(len(self.data.item), self.emb_size) / 3
        self.loss, self.lastLoss = 0, 0

    def updateLearningRate(self, epoch):
        if epoch > 1:
            self.lRate *= 1.05 if abs(self.lastLoss) > abs(self.loss) else 0.5
        if self.lRate > self.maxLRate > 0:
            self.lRate = self.maxLRate

    def predictForRating(self, u, i):
        has_u, has_i = self.data.containsUser(u), self.data.containsItem(i)
        if has_u and has_i:
            return self.P[self.data.user[u]].dot(self.Q[self.data.item[i]])
        if has_u:
            return self.data.userMeans[u]
        if has_i:
            return self.data.itemMeans[i]
        return self.data.globalMean

    def predictForRanking(self, u):
        if self.data.containsUser(u):
            return self.Q.dot(self.P[self.data.user[u]])
        return [self.data.globalMean] * self.num_items

    def ranking_tables(self):
        """(user table, item table) that ``predictForRanking`` multiplies, as host arrays of
        one float dtype -- numpy-path models score ``Q.dot(P[u])`` in fp64
        (base/iterativeRecommender.py:75-80), the TF-path ones ``V.dot(U[u])`` in fp32."""
        if hasattr(self, "U") and hasattr(self, "V"):
            return self.U, self.V
        return self.P, self.Q

    def rank_all_test_users(self, N):
        """The evalRanking inner loop (base/recommender.py:143-150) for all test users at
        once on the device: MFMA scoring, rated items -> 0, reference heap top-N (ties
        included).  Users unknown to the training set get the reference's constant-score
        list (base/iterativeRecommender.py:79-80) on the host."""
        from ..ranking import DeviceRanker
        from ..util.qmath import find_k_largest
        if min(N, self.num_items) > 100:
            # only the per-epoch ranking_performance can ask for this (it takes max(-topN) unclamped, as the reference
            # does; the final evalRanking clamps to 10): the device ranker serves N <= 100, the reference's own loop the rest
            return super().rank_all_test_users(N)
        U, V = self.ranking_tables()
        users = list(self.data.testSet_u)
        warm = [u for u in users if self.data.containsUser(u)]
        recList = {}
        if warm:
            U, V = np.ascontiguousarray(U), np.ascontiguousarray(V)
            ranker = self._device_ranker(U, V)
            uid = np.fromiter((self.data.user[u] for u in warm), dtype=np.int32, count=len(warm))
            ids, scores = ranker.topk(uid, min(N, self.num_items))
            id2item = self.data.id2item
            for u, row_i, row_s in zip(warm, ids.tolist(), scores.tolist()):
                recList[u] = [(id2item[i], s) for i, s in zip(row_i, row_s)]
        for u in users:
            if u not in recList:
                k_ids, k_sc = find_k_largest(N, [self.data.globalMean] * self.num_items)
                recList[u] = [(self.data.id2item[i], s) for i, s in zip(k_ids, k_sc)]
        return {u: recList[u] for u in users}   # testSet_u order, as the reference builds it

    def _device_ranker(self, U, V):
        ranker = getattr(self, "_ranker", None)
        if ranker is not None and (ranker.n_users, ranker.n_items, ranker.d, ranker.dtype) == (U.shape[0], V.shape[0], U.shape[1], U.dtype):
            ranker.update_tables(U, V)             # per-epoch evaluation: keep buffers, re-upload tables
        else:
            from ..ranking import DeviceRanker
            ranker = self._ranker = DeviceRanker(U, V, self.data.rated_csr())
        return ranker

    def data_parallel(self):
        """dist.BatchParallel when the run was started one process per GPU (qrec_amd.main under torch.distributed.run),
        else None.  Looked up at training time, never in a constructor (the device is bound per process)."""
        if not hasattr(self, "_dp"):
            from ..dist import BatchParallel
            self._dp = BatchParallel.from_env()
        return self._dp

    @staticmethod
    def step_share(dp, n_rows: int):
        """(offset, count): this rank's share of a step's rows"""
        return dp.share(n_rows) if dp else (0, n_rows)

    def rank_measure_all_test_users(self, top, N):
        """Measure.rankingMeasure(testSet_u, recList, top) (base/recommender.py:167, util/measure.py:24-49) without
        the recList: per-user hit counts and DCG sums are taken from the top-N lists on the device
        (qrec_rank_hits), cold users (constant-score lists, base/iterativeRecommender.py:79-80) on the host."""
        from ..interactions import user_item_csr
        from ..ranking import ranking_measure_strings
        from ..util.qmath import find_k_largest
        import math
        if min(N, self.num_items) > 100:
            return None                      # host path (see rank_all_test_users)
        U, V = self.ranking_tables()
        U, V = np.ascontiguousarray(U), np.ascontiguousarray(V)
        users = list(self.data.testSet_u)
        cache = getattr(self, "_test_cache", None)
        if cache is None:
            warm_pos = [k for k, u in enumerate(users) if self.data.containsUser(u)]
            uid, iid = [], []
            for k in warm_pos:
                row = self.data.user[users[k]]
                for item in self.data.testSet_u[users[k]]:
                    if item in self.data.item:
                        uid.append(row); iid.append(self.data.item[item])
            test = user_item_csr(np.array(uid, np.int32), np.array(iid, np.int32), np.ones(len(uid)), U.shape[0], V.shape[0])
            cache = self._test_cache = dict(
                warm_pos=np.array(warm_pos, np.int64), test=test, lens=[len(self.data.testSet_u[u]) for u in users],
                warm_uid=np.fromiter((self.data.user[users[k]] for k in warm_pos), dtype=np.int32, count=len(warm_pos)))
        cuts = sorted({min(n, N, self.num_items) for n in top})
        per_n = {n: (np.zeros(len(users), np.int64), np.zeros(len(users), np.float64)) for n in top}
        dp = self.data_parallel()
        if cache["warm_uid"].size:
            ranker = self._device_ranker(U, V)
            if ranker.test is None:
                ranker.set_test(cache["test"])
            # multi-GPU run: the test users are split over the ranks (the tables are replicated); the per-user hit
            # counts / DCG sums of the shares are disjoint, so their sum over the ranks is exact
            lo, cnt = dp.share(cache["warm_uid"].size) if dp else (0, cache["warm_uid"].size)
            mine = slice(lo, lo + cnt)
            per_cut = ranker.topk(cache["warm_uid"][mine], min(N, self.num_items), cuts=cuts, want_lists=False)[2] if cnt else {}
            for n in top:
                hits = np.zeros(cache["warm_uid"].size, np.int64); dcg = np.zeros(cache["warm_uid"].size, np.float64)
                if cnt:
                    hits[mine], dcg[mine] = per_cut[min(n, N, self.num_items)]
                if dp:
                    hits, dcg = dp.all_reduce_host(hits), dp.all_reduce_host(dcg)
                per_n[n][0][cache["warm_pos"]] = hits; per_n[n][1][cache["warm_pos"]] = dcg
        if cache["warm_pos"].size < len(users):
            warm = set(cache["warm_pos"].tolist())
            ids, _ = find_k_largest(N, [self.data.globalMean] * self.num_items)      # cold users: one constant-score list
            for k, u in enumerate(users):
                if k in warm:
                    continue
                truth = self.data.testSet_u[u]
                for n in top:
                    h, x = 0, 0.0
                    for pos, iid in enumerate(ids[:n]):
                        if self.data.id2item[iid] in truth:
                            h += 1; x += 1.0 / math.log(pos + 2)
                    per_n[n][0][k] = h; per_n[n][1][k] = x
        return ranking_measure_strings(cache["lens"], per_n, top)

    def shuffle_training_data(self):
        """``shuffle(self.data.trainingData)`` (base/iterativeRecommender.py:101) with the
        exact CPython draw sequence, done natively: a 1.2 M-row list takes ~1 s in
        ``random.shuffle`` and ~10 ms here.  The Python generator state is advanced in
        lock-step so later ``random`` calls agree with the reference."""
        n = self.data.elemCount()
        state = random.getstate()
        words = capi.state_from_python(state)
        perm = np.arange(n, dtype=np.int64)
        capi.mt_shuffle(words, n, perm)
        random.setstate(capi.state_to_python(words, state[2]))
        self.data.permute_training_data(perm)      # the Python list is rebuilt only if someone reads it
        return perm

    def isConverged(self, epoch):
        if isnan(self.loss):
            print("Loss = NaN or Infinity: current settings does not fit the recommender! Change the settings and try again!")
            sys.exit(-1)
        deltaLoss = self.lastLoss - self.loss
        if self.ranking.isMainOn():
            print("%s %s epoch %d: loss = %.4f, delta_loss = %.5f learning_Rate = %.5f"
                  % (self.modelName, self.foldInfo, epoch, self.loss, deltaLoss, self.lRate))
        else:
            measure = self.rating_performance()
            print("%s %s epoch %d: loss = %.4f, delta_loss = %.5f learning_Rate = %.5f %5s %5s"
                  % (self.modelName, self.foldInfo, epoch, self.loss, deltaLoss, self.lRate,
                     measure[0].strip()[:11], measure[1].strip()[:12]))
        converged = abs(deltaLoss) < 1e-3
        if not converged:
            self.updateLearningRate(epoch)
        self.lastLoss = self.loss
        self.shuffle_training_data()
        return converged

    def rating_performance(self):
        res = [[user, item, rating, self.checkRatingBoundary(self.predictForRating(user, item))]
               for user, item, rating in self.data.testData]
        self.measure = Measure.ratingMeasure(res)
        return self.measure

    def ranking_performance(self, epoch):
        """Evaluation during training (base/iterativeRecommender.py:115-185): top-max(N)
        only, remembers the best epoch and snapshots the model through saveModel()."""
        N = max(int(x) for x in self.ranking["-topN"].split(","))
        print("Evaluating...")
        measure = self.rank_measure_all_test_users([N], N)
        if measure is None:
            measure = Measure.rankingMeasure(self.data.testSet_u, self.rank_all_test_users(N), [N])
        performance = {}
        for m in measure[1:]:
            k, v = m.strip().split(":")
            performance[k] = float(v)
        if self.bestPerformance:
            worse = sum(1 if self.bestPerformance[1][k] > performance[k] else -1 for k in self.bestPerformance[1])
            if worse < 0:
                self.bestPerformance = [epoch + 1, performance]
                self.saveModel()
        else:
            self.bestPerformance = [epoch + 1, performance]
            self.saveModel()
        print("-" * 120)
        print("Quick Ranking Performance " + self.foldInfo + " (Top-" + str(N) + "Item Recommendation)")
        measure = [m.strip() for m in measure[1:]]
        print("*Current Performance*")
        print("Epoch:", str(epoch + 1) + ",", " | ".join(measure))
        best = self.bestPerformance[1]
        print("*Best Performance* ")
        print("Epoch:", str(self.bestPerformance[0]) + ",",
              "Precision:" + str(best["Precision"]) + " | Recall:" + str(best["Recall"]) +
              " | F1:" + str(best["F1"]) + " | MDCG:" + str(best["NDCG"]))
        print("-" * 120)
        return measure
