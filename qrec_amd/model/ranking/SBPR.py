"""SBPR behind the reference's class name and hooks (model/ranking/SBPR.py:8-148, numpy path): BPR with social feedback -- for a user
whose followees consumed items the user did not, every positive item is ranked above one of those (scaled by how many friends consumed
it) and that one above a random negative.

What the reference's file does, stated plainly: ``trainModel`` raises ``TypeError: unhashable type: 'list'`` at SBPR.py:46
(``Suk = self.FPSet[user][kItems]`` indexes a dict with the list it was drawn from) as soon as the loop reaches a user who HAS social
feedback (recorded by running it: tests/golden/golden_meta.json ``sbpr_filmtrust.unmodified_reference_raises``); only data without any
social feedback trains.  This class runs the loop with that subscript read as ``item_k`` -- the count the drawn item carries, the
statement's evident meaning -- and everything else as written, including the parts that look unintended: the negative's rejection test
``item_j in self.FPSet`` (:52) looks the ITEM's name up among the USER names that are keys of the defaultdict so far; the item biases
enter the scores but no statement updates them; the decays and the loss of :56-58 exist only on the social branch; the table terms of
the loss (:74) are added once per user.  Draws come from the CPython ``random`` stream (replayed natively), the updates run strictly in
order on the device (fp64): the run of the reference's source with that one token replaced is reproduced -- same rows, same tables,
same loss, same learning-rate schedule (tests/test_gpu_bpr.py::test_sbpr_model_reproduces_the_reference_run)."""
from __future__ import annotations

import random

import numpy as np

from ... import capi
from ...base.socialRecommender import SocialRecommender
from ...capi import DeviceBuffer
from ...engine import DeviceTables


class SBPR(SocialRecommender):
    def __init__(self, conf, trainingSet=None, testSet=None, relation=None, fold="[1]"):
        super().__init__(conf, trainingSet, testSet, relation if relation is not None else [], fold)

    def initModel(self):
        """PositiveSet and FPSet (SBPR.py:12-29) as arrays: FPSet[user] = items the user's followees have in their training rows and
        the user is not positive on, in the order the reference's dict meets them, with the number of such followees."""
        super().initModel()
        print("Preparing item sets...")
        pos, rated = self.data.positive_csr(), self.data.rated_csr()
        n_users = len(self.data.user)
        ptr, items, counts, ps_users = [0], [], [], []
        social_users, followees = self.social.user, self.social.getFollowees
        for user, row in self.data.user.items():                                         # id order
            own = set(pos.indices[pos.indptr[row]:pos.indptr[row + 1]].tolist())
            in_positive_set = bool(own)
            book = {}
            if user in social_users:
                for friend in followees(user):
                    if friend in self.data.user:
                        f = self.data.user[friend]
                        theirs = rated.indices[rated.indptr[f]:rated.indptr[f + 1]].tolist()
                        in_positive_set = in_positive_set or bool(theirs)               # `item not in self.PositiveSet[user]` (:25) makes the key
                        for it in theirs:
                            if it not in own:
                                book[it] = book.get(it, 0) + 1
            if in_positive_set:
                ps_users.append(row)
            items += list(book.keys()); counts += list(book.values())
            ptr.append(len(items))
        self._fp = (np.asarray(ptr, np.int64), np.asarray(items, np.int32), np.asarray(counts, np.int32))
        self._ps_users = np.asarray(ps_users, np.int32)
        # `item_j in self.FPSet`: item names against user names
        self._item_key_user = np.fromiter((self.data.user.get(name, -1) for name in self.data.item), dtype=np.int32, count=len(self.data.item))
        self._is_key = (np.diff(self._fp[0]) > 0).astype(np.uint8)                         # keys after initModel: users WITH feedback (:26-27)
        assert n_users == self._is_key.size

    def trainModel(self):
        self.b = np.random.random(self.num_items)                                        # SBPR.py:32 (never updated afterwards)
        print("Training...")
        pos = self.data.positive_csr()
        n_items = len(self.data.item)
        tables = DeviceTables(self.P, self.Q, np.float64)
        d_bias = DeviceBuffer.from_numpy(np.ascontiguousarray(self.b, dtype=np.float64))
        bb = float(self.b.dot(self.b))
        cap = max(pos.nnz + self._ps_users.size, 1)
        d_rows = DeviceBuffer((cap, 5), np.int32)
        d_sums, d_loss = DeviceBuffer.zeros(2, np.float64), DeviceBuffer.zeros(2, np.float64)
        # users of PositiveSet without a positive item still pass through the loop (kItems, the per-user loss terms): a bare visit row each
        empty = self._ps_users[np.diff(pos.indptr)[self._ps_users] == 0]
        epoch = 0
        while epoch < self.maxEpoch:
            state = random.getstate()
            words = capi.state_from_python(state)
            rows = capi.mt_sbpr_sample_epoch(words, self._ps_users, pos.indptr, pos.indices, n_items, *self._fp, self._item_key_user, self._is_key)
            random.setstate(capi.state_to_python(words, state[2]))
            if empty.size:                                                               # merge the visits at their place in PositiveSet's order
                place = np.full(len(self.data.user), -1, np.int64); place[self._ps_users] = np.arange(self._ps_users.size)
                visits = np.full((empty.size, 5), -1, np.int32); visits[:, 0] = empty; visits[:, 4] = 0
                both = np.concatenate([rows, visits])
                rows = both[np.argsort(place[both[:, 0]], kind="stable")]
            n = int(rows.shape[0])
            d_rows.upload_head(np.ascontiguousarray(rows))
            capi.sumsq(tables.P, tables.code, tables.n_users, tables.d, tables.ld, d_sums.ptr)
            capi.sumsq(tables.Q, tables.code, tables.n_items, tables.d, tables.ld, d_sums.ptr + 8)
            capi.sbpr_sgd_ordered(tables.P, tables.Q, d_bias, tables.code, tables.d, tables.ld, d_rows, n, self.lRate, self.regU, self.regI, bb,
                                  d_sums, d_loss)
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
