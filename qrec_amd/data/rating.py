"""The rating data model: same public surface as the reference's ``Rating``
(data/rating.py:5-190) -- name<->id maps in first-appearance order, ``trainSet_u/i``,
``testSet_u/i``, means, rating scale -- plus the array/CSR views the device kernels use
(``train_uid``, ``train_iid``, ``positive_csr()``, ``rated_csr()``).
"""
from __future__ import annotations

import random
from collections import defaultdict

import numpy as np

from ..interactions import CSR, dedup_user_item, user_item_csr
from ..util.config import OptionConf
from .rows import RatingRows


class _LazyNested(dict):
    """``trainSet_u`` / ``trainSet_i`` / ``testSet_i`` of the array-backed data model: the reference's dict of
    dicts (data/rating.py:14-17), built from the arrays the first time anybody looks inside."""

    def __init__(self, build):
        super().__init__()
        self._build = build

    def _fill(self):
        if self._build is not None:
            build, self._build = self._build, None
            dict.update(self, build())

    def _wrap(name):
        def method(self, *a, **kw):
            self._fill()
            return getattr(dict, name)(self, *a, **kw)
        method.__name__ = name
        return method

    for _m in ("__getitem__", "__contains__", "__iter__", "__len__", "keys", "values", "items", "get", "__repr__",
               "__eq__", "__setitem__", "__delitem__", "pop", "setdefault", "update", "copy"):
        locals()[_m] = _wrap(_m)
    del _m, _wrap

    def __missing__(self, key):              # defaultdict(dict) behaviour of the reference
        row = self[key] = {}
        return row


def _nested(outer_names, outer_idx, inner_names, inner_idx, values):
    """{outer: {inner: value}} with the reference's insertion orders (rows in file order)."""
    order = np.argsort(outer_idx, kind="stable")
    bounds = np.flatnonzero(np.diff(outer_idx[order])) + 1
    inner_sorted = [inner_names[k] for k in inner_idx[order].tolist()]
    val_sorted = values[order].tolist()
    starts = [0] + bounds.tolist() + [order.size]
    firsts = outer_idx[order][[s for s in starts[:-1]]].tolist() if order.size else []
    groups = {g: dict(zip(inner_sorted[a:b], val_sorted[a:b])) for g, a, b in zip(firsts, starts[:-1], starts[1:])}
    # outer keys in order of first appearance
    _, first = np.unique(outer_idx, return_index=True)
    return {outer_names[g]: groups[g] for g in outer_idx[np.sort(first)].tolist()}


class Rating:
    def __init__(self, config, trainingSet, testSet):
        self.config = config
        self.evalSettings = OptionConf(config["evaluation.setup"])
        self.user, self.item, self.id2user, self.id2item = {}, {}, {}, {}
        self.userMeans, self.itemMeans, self.globalMean = {}, {}, 0
        self.trainSet_u, self.trainSet_i = defaultdict(dict), defaultdict(dict)
        self.testSet_u, self.testSet_i = defaultdict(dict), defaultdict(dict)
        self.rScale = []
        self._order = None                   # pending permutation of _rows (see permute_training_data)
        self._arrays = None                  # (uid, iid, rating) of _rows in list order
        self._csr_cache = {}
        self._dedup = None
        special = any(self.evalSettings.contains(k) for k in ("-val", "-cold", "-predict"))
        if isinstance(trainingSet, RatingRows) and isinstance(testSet, RatingRows) and not special:
            self._rows = trainingSet
            self._test_rows = testSet
            self._ingest_compact(trainingSet, testSet)
            return
        # list form (also what the options that rewrite the lists themselves work on)
        if isinstance(trainingSet, RatingRows):
            trainingSet = trainingSet.to_list()
        if isinstance(testSet, RatingRows):
            testSet = testSet.to_list()
        self._rows = trainingSet[:]          # the list form of trainingData (kept in sync lazily)
        self._test_rows = None
        self.testData = testSet[:]
        self._ingest()
        self._means()
        if self.evalSettings.contains("-cold"):
            self._keep_cold_start_users(int(self.evalSettings["-cold"]))

    # ``testData`` (data/rating.py:25): a plain list in the list-backed model; materialised on first use in the
    # array-backed one (evalRatings appends the prediction to every row, base/recommender.py:105-110)
    @property
    def testData(self):
        if self._test_list is None and self._test_rows is not None:
            self._test_list = self._test_rows.to_list()
        return self._test_list

    @testData.setter
    def testData(self, rows):
        self._test_list = rows

    _test_list = None

    def _ingest_compact(self, tr: RatingRows, te: RatingRows):
        """``_ingest`` + ``_means`` on arrays: same ids (first appearance in the TRAINING rows, data/rating.py:48-54),
        same dict orders, same floating-point sums (sequential, in dict order) -- without touching 1.2 M Python rows."""
        uorder, uremap = RatingRows.first_appearance(tr.user_idx, len(tr.user_names))
        iorder, iremap = RatingRows.first_appearance(tr.item_idx, len(tr.item_names))
        unames = [tr.user_names[f] for f in uorder.tolist()]
        inames = [tr.item_names[f] for f in iorder.tolist()]
        self.user = dict(zip(unames, range(len(unames))))
        self.item = dict(zip(inames, range(len(inames))))
        self.id2user = dict(enumerate(unames))
        self.id2item = dict(enumerate(inames))
        uid, iid, r = uremap[tr.user_idx], iremap[tr.item_idx], tr.rating
        self._arrays = (np.ascontiguousarray(uid), np.ascontiguousarray(iid), r.copy())
        self.rScale = np.unique(r).tolist()
        nu, ni = len(unames), len(inames)
        du, di, dr = self._dedup = dedup_user_item(uid, iid, r, max(ni, 1))
        # means: sum(row.values()) / len(row) in dict order == bincount's sequential accumulation in row order
        with np.errstate(invalid="ignore", divide="ignore"):
            um = np.bincount(du, weights=dr, minlength=nu) / np.bincount(du, minlength=nu)
            im = np.bincount(di, weights=dr, minlength=ni) / np.bincount(di, minlength=ni)
        self.userMeans = dict(zip(unames, um.tolist()))
        self.itemMeans = dict(zip(inames, im.tolist()))
        total = sum(self.userMeans.values())
        self.globalMean = 0 if total == 0 else total / len(self.userMeans)
        self.trainSet_u = _LazyNested(lambda: _nested(unames, du, inames, di, dr))
        self.trainSet_i = _LazyNested(lambda: _nested(inames, di, unames, du, dr))
        # test side: names stay names (test users/items may be unknown to the training set)
        tun, tin = te.user_names, te.item_names
        self.testSet_u = _nested(tun, te.user_idx, tin, te.item_idx, te.rating) if len(te) else {}
        self.testSet_i = _LazyNested(lambda: _nested(tin, te.item_idx, tun, te.user_idx, te.rating) if len(te) else {})

    # ``trainingData`` is the reference's public list (data/rating.py:24).  isConverged /
    # next_batch_pairwise reshuffle it every epoch (base/iterativeRecommender.py:101,
    # base/deepRecommender.py:30); rebuilding a 1.2 M-row Python list and re-deriving id arrays from it
    # costs ~0.7 s per epoch, so the order lives in a permutation over cached id arrays and the list
    # itself is only rebuilt when somebody reads it.
    @property
    def trainingData(self):
        if self._order is not None:
            rows = self._rows
            self._rows = rows.take(self._order) if isinstance(rows, RatingRows) else [rows[k] for k in self._order]
            if self._arrays is not None:
                self._arrays = tuple(a[self._order] for a in self._arrays)
            self._order = None
        return self._rows

    @trainingData.setter
    def trainingData(self, rows):
        self._rows = rows
        self._order = None
        self._arrays = None

    def permute_training_data(self, perm: np.ndarray):
        """new_list[k] = old_list[perm[k]] -- applied lazily."""
        self._order = perm if self._order is None else self._order[perm]

    # ---- construction -----------------------------------------------------------------
    def _ingest(self):
        if self.evalSettings.contains("-val"):
            # validation split carved out of the training rows (data/rating.py:37-41)
            random.shuffle(self.trainingData)
            cut = int(len(self.trainingData) * float(self.evalSettings["-val"]))
            self.testData = self.trainingData[:cut]
            self.trainingData = self.trainingData[cut:]
        user, item = self.user, self.item
        scale = set()
        for userName, itemName, rating in self.trainingData:
            if userName not in user:
                user[userName] = len(user)
            if itemName not in item:
                item[itemName] = len(item)
            self.trainSet_u[userName][itemName] = rating
            self.trainSet_i[itemName][userName] = rating
            scale.add(float(rating))
        self.id2user = {v: k for k, v in user.items()}
        self.id2item = {v: k for k, v in item.items()}
        self.rScale = sorted(scale)
        predict_only = self.evalSettings.contains("-predict")
        for entry in self.testData:
            if predict_only:
                self.testSet_u[entry] = {}
            else:
                userName, itemName, rating = entry
                self.testSet_u[userName][itemName] = rating
                self.testSet_i[itemName][userName] = rating

    def _means(self):
        for u in self.user:
            row = self.trainSet_u[u]
            self.userMeans[u] = sum(row.values()) / len(row)
        for i in self.item:
            col = self.trainSet_i[i]
            self.itemMeans[i] = sum(col.values()) / len(col)
        total = sum(self.userMeans.values())
        self.globalMean = 0 if total == 0 else total / len(self.userMeans)

    def _keep_cold_start_users(self, threshold: int):
        warm = {u for u in self.testSet_u if u in self.trainSet_u and len(self.trainSet_u[u]) > threshold}
        for u in warm:
            del self.testSet_u[u]
        self.testData = [row for row in self.testData if row[0] not in warm]

    # ---- reference accessors -------------------------------------------------------------
    def getUserId(self, u):
        return self.user.get(u)

    def getItemId(self, i):
        return self.item.get(i)

    def trainingSize(self):
        return (len(self.user), len(self.item), len(self._rows))

    def testSize(self):
        return (len(self.testSet_u), len(self.testSet_i), len(self.testData))

    def contains(self, u, i):
        return u in self.user and i in self.trainSet_u[u]

    def containsUser(self, u):
        return u in self.user

    def containsItem(self, i):
        return i in self.item

    def userRated(self, u):
        row = self.trainSet_u[u]
        return list(row.keys()), list(row.values())

    def itemRated(self, i):
        col = self.trainSet_i[i]
        return list(col.keys()), list(col.values())

    def row(self, u):
        vec = np.zeros(len(self.item))
        for name, r in self.trainSet_u[u].items():
            vec[self.item[name]] = r
        return vec

    def col(self, i):
        vec = np.zeros(len(self.user))
        for name, r in self.trainSet_i[i].items():
            vec[self.user[name]] = r
        return vec

    def matrix(self):
        m = np.zeros((len(self.user), len(self.item)))
        for u, uid in self.user.items():
            m[uid] = self.row(u)
        return m

    def sRow(self, u):
        return self.trainSet_u[u]

    def sCol(self, c):
        return self.trainSet_i[c]

    def rating(self, u, c):
        return self.trainSet_u[u][c] if self.contains(u, c) else -1

    def ratingScale(self):
        return (self.rScale[0], self.rScale[1])

    def elemCount(self):
        return len(self._rows)

    # ---- array views for the kernels -----------------------------------------------------
    def training_arrays(self):
        """(uid int32[n], iid int32[n], rating float64[n]) of ``trainingData`` in its
        CURRENT order (isConverged reshuffles that list every epoch)."""
        if self._arrays is None:
            rows = self._rows
            n = len(rows)
            self._arrays = (np.fromiter((self.user[r[0]] for r in rows), dtype=np.int32, count=n),
                            np.fromiter((self.item[r[1]] for r in rows), dtype=np.int32, count=n),
                            np.fromiter((r[2] for r in rows), dtype=np.float64, count=n))
        if self._order is None:
            return tuple(a.copy() for a in self._arrays)
        return tuple(np.ascontiguousarray(a[self._order]) for a in self._arrays)

    def _dict_csr(self, min_rating):
        if self._dedup is not None:       # array-backed model: the same orders from the de-duplicated arrays
            du, di, dr = self._dedup
            return user_item_csr(du, di, dr, len(self.user), len(self.item), min_rating, assume_unique=True)
        # walk the dicts themselves: their iteration order IS the contract
        nu = len(self.user)
        counts = np.zeros(nu + 1, dtype=np.int64)
        cols, vals = [], []
        for u, uid in self.user.items():  # id order == insertion order
            row = self.trainSet_u[u]
            if min_rating is None:
                items = list(row.items())
            else:
                items = [(i, r) for i, r in row.items() if r >= min_rating]
            counts[uid + 1] = len(items)
            cols.extend(self.item[i] for i, _ in items)
            vals.extend(r for _, r in items)
        return CSR(np.cumsum(counts), np.asarray(cols, dtype=np.int32), np.asarray(vals, dtype=np.float64))

    def positive_csr(self) -> CSR:
        """BPR's PositiveSet (rating >= 1) in its iteration order (model/ranking/BPR.py:21-33)."""
        if "pos" not in self._csr_cache:
            self._csr_cache["pos"] = self._dict_csr(1)
        return self._csr_cache["pos"]

    def rated_csr(self) -> CSR:
        """All train items per user, dict order (mask list of evalRanking, sampler membership)."""
        if "rated" not in self._csr_cache:
            self._csr_cache["rated"] = self._dict_csr(None)
        return self._csr_cache["rated"]
