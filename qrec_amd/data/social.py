"""Social relations behind the reference's ``Social`` accessors (data/social.py:4-74): who follows whom, user order
of first appearance, and the trust matrix views.  ``relation`` = list of ``[follower, followee, weight]``."""
from __future__ import annotations

from collections import defaultdict

import numpy as np


class Social:
    def __init__(self, conf, relation=None):
        self.config = conf
        self.user = {}                                   # name -> index, first appearance (data/social.py:21-24)
        self.relation = relation if relation is not None else []
        self.followees = defaultdict(dict)
        self.followers = defaultdict(dict)
        for u1, u2, weight in self.relation:
            self.followees[u1][u2] = weight
            self.followers[u2][u1] = weight
            if u1 not in self.user:
                self.user[u1] = len(self.user)
            if u2 not in self.user:
                self.user[u2] = len(self.user)
        # the trust matrix' shape counts distinct followers x distinct followees (util/structure/new_sparseMatrix.py:19);
        # duplicates of a pair overwrite each other but all count in ``elemNum``
        self._by_row = defaultdict(dict); self._by_col = defaultdict(dict)
        for u1, u2, weight in self.relation:
            self._by_row[self.user[u1]][self.user[u2]] = weight
            self._by_col[self.user[u2]][self.user[u1]] = weight
        self._size = (len(self._by_row), len(self._by_col))

    def _dense(self, entries, width):
        out = np.zeros((1, width))
        if entries:
            out[0][list(entries.keys())] = list(entries.values())
        return out

    def row(self, u):
        """user u's followees as a 1 x n array"""
        return self._dense(self._by_row.get(self.user[u]), self._size[1])

    def col(self, u):
        """user u's followers as a 1 x n array"""
        return self._dense(self._by_col.get(self.user[u]), self._size[0])

    def elem(self, u1, u2):
        return self._by_row.get(u1, {}).get(u2, 0)

    def weight(self, u1, u2):
        if u1 in self.followees and u2 in self.followees[u1]:
            return self.followees[u1][u2]
        return 0

    def trustSize(self):
        return self._size

    def getFollowers(self, u):
        return self.followers[u] if u in self.followers else {}

    def getFollowees(self, u):
        return self.followees[u] if u in self.followees else {}

    def hasFollowee(self, u1, u2):
        return u1 in self.followees and u2 in self.followees[u1]

    def hasFollower(self, u1, u2):
        return u1 in self.followers and u2 in self.followers[u1]
