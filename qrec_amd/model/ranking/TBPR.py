"""TBPR behind the reference's class name and hooks (model/ranking/TBPR.py:8-187, numpy path): BPR whose every
positive item is followed by a chain of socially exposed items -- joint, weak-tie and strong-tie feedback, ties split at
the median Jaccard strength of the followee sets -- and closed by a random negative; consecutive chain members are
ranked pairwise.  Negatives and chain members come from the CPython ``random`` stream (replayed natively), the chained
updates run strictly in order on the device (fp64), so the reference's run is reproduced: same triplets, same tables,
same loss (including its per-user regularisation terms), same learning-rate schedule."""
from __future__ import annotations

import random

import numpy as np

from ... import capi
from ...base.socialRecommender import SocialRecommender
from ...capi import DeviceBuffer
from ...engine import DeviceTables
from ...util import config


class TBPR(SocialRecommender):
    def __init__(self, conf, trainingSet=None, testSet=None, relation=None, fold="[1]"):
        super().__init__(conf, trainingSet, testSet, relation if relation is not None else [], fold)

    def readConfiguration(self):
        super().readConfiguration()
        self.regT = float(config.OptionConf(self.config["TBPR"])["-regT"])

    def initModel(self):
        """tie strength = Jaccard index of the two users' followee sets; theta = its median (TBPR.py:17-42)"""
        super().initModel()
        followees = self.social.getFollowees
        self.strength, weights = {}, []
        for u1 in self.social.user:
            mine = set(followees(u1))
            for u2 in followees(u1):
                if u1 == u2:
                    continue
                theirs = set(followees(u2))
                s = len(mine & theirs) / float(len(mine | theirs))
                self.strength.setdefault(u1, {})[u2] = s
                weights.append(s)
        self.weights = np.array(sorted(weights))
        self.theta = np.median(self.weights)
        self._split_ties()
        half = len(self.weights) // 2
        self.t_s = self.weights[half + 1:].sum() / float(len(self.weights[half + 1:]))
        self.t_w = self.weights[:half].sum() / float(len(self.weights[:half]))

    def _split_ties(self):
        self.strongTies, self.weakTies = {}, {}
        for u1, row in self.strength.items():
            for u2, s in row.items():
                (self.strongTies if s > self.theta else self.weakTies).setdefault(u1, {})[u2] = s

    def _item_sets(self):
        """Per user the three candidate lists of TBPR.py:96-121 as CSR over user ids (joint, weak, strong): what the
        user's strong / weak ties consumed and the user did not; items on both sides form the joint list and leave the
        other two.  Weak and strong lists keep the order in which the reference's dicts meet the items; the joint list
        has the order of the Python set the reference builds from the item names (it depends on the process' string
        hashing there as here)."""
        pos = self.data.positive_csr()
        id2item = self.data.id2item
        ptr = {k: [0] for k in ("joint", "weak", "strong")}
        items = {k: [] for k in ("joint", "weak", "strong")}
        social_users = self.social.user

        def exposed(user, ties, own):
            seen, out = set(), []
            for friend in ties.get(user, ()):
                row = self.data.user[friend]
                for it in pos.indices[pos.indptr[row]:pos.indptr[row + 1]].tolist():
                    if it not in own and it not in seen:
                        seen.add(it); out.append(it)
            return out
        for user, row in self.data.user.items():                        # id order
            if user in social_users:
                own = set(pos.indices[pos.indptr[row]:pos.indptr[row + 1]].tolist())
                strong, weak = exposed(user, self.strongTies, own), exposed(user, self.weakTies, own)
                both = set(id2item[i] for i in strong).intersection(set(id2item[i] for i in weak))
                joint = [self.data.item[name] for name in dict.fromkeys(both, 1)]
                drop = set(joint)
                items["joint"] += joint
                items["weak"] += [i for i in weak if i not in drop]
                items["strong"] += [i for i in strong if i not in drop]
            for k in ptr:
                ptr[k].append(len(items[k]))
        return tuple((np.asarray(ptr[k], np.int64), np.asarray(items[k], np.int32)) for k in ("joint", "weak", "strong"))

    def trainModel(self):
        pos = self.data.positive_csr()                                   # positiveSet, TBPR.py:66-70
        n_items = len(self.data.item)
        print("Training...")
        tables = DeviceTables(self.P, self.Q, np.float64)
        cap = 4 * max(pos.nnz, 1)
        d_u, d_a, d_b = (DeviceBuffer(cap, np.int32) for _ in range(3))
        d_sums, d_loss = DeviceBuffer.zeros(2, np.float64), DeviceBuffer.zeros(2, np.float64)
        sets = None
        epoch = 0
        while epoch < self.maxEpoch:
            self.theta_derivative, self.theta_count = 0, 0               # never fed: optimization_theta has no caller
            if self.theta > self.weights.max():
                self.theta = self.weights.max() - 0.01
            if self.theta < self.weights.min():
                self.theta = self.weights.min() + 0.01
            try:
                above = [w for w in self.weights if w >= self.theta]; below = [w for w in self.weights if w <= self.theta]
                self.t_s = sum(above) / len(above)
                self.t_w = sum(below) / len(below)
            except ZeroDivisionError:
                self.t_w = 0.01
                self.theta = 0.02
            self.g_theta = (self.t_s - self.theta) * (self.theta - self.t_w)
            print("Theta:", self.theta)
            print("g_theta:", self.g_theta)
            print("Preparing item sets...")
            if sets is None:                                             # theta never moves, so neither do the sets
                sets = self._item_sets()
            print("Computing...")
            state = random.getstate()
            words = capi.state_from_python(state)
            u, a, b = capi.mt_tbpr_sample_epoch(words, pos.indptr, pos.indices, n_items, *sets)
            random.setstate(capi.state_to_python(words, state[2]))
            n = int(u.size)
            d_u.upload_head(u); d_a.upload_head(a); d_b.upload_head(b)
            capi.sumsq(tables.P, tables.code, tables.n_users, tables.d, tables.ld, d_sums.ptr)
            capi.sumsq(tables.Q, tables.code, tables.n_items, tables.d, tables.ld, d_sums.ptr + 8)
            capi.tbpr_sgd_ordered(tables.P, tables.Q, tables.code, tables.d, tables.ld, d_u, d_a, d_b, n, self.lRate,
                                  self.regU, self.regI, d_sums, d_loss)
            nll, reg = d_loss.numpy()
            self.loss = float(nll) + float(reg)
            epoch += 1
            if not self.ranking.isMainOn():      # isConverged then prints MAE/RMSE from the host tables (reference: live values)
                self.P, self.Q = tables.download(np.float64)
            if self.isConverged(epoch):
                break
        self.P, self.Q = tables.download(np.float64)

    def predictForRanking(self, u):
        if self.data.containsUser(u):
            return self.Q.dot(self.P[self.data.getUserId(u)])
        return [self.data.globalMean] * self.num_items
