"""Graph recommenders: the joint normalized user-item adjacency of the reference's
``GraphRecommender`` (base/graphRecommender.py:5-60)."""
from __future__ import annotations

import numpy as np

from ..graph import joint_norm_adjacency
from .deepRecommender import DeepRecommender


class GraphRecommender(DeepRecommender):
    def __init__(self, conf, trainingSet, testSet, fold="[1]"):
        super().__init__(conf, trainingSet, testSet, fold)

    def create_joint_sparse_adjaceny(self):
        """CSR triple (indptr, indices, values) of D^-1/2 (R+R^T) D^-1/2 over users+items."""
        uid, iid, _ = self.data.training_arrays()
        return joint_norm_adjacency(self.num_users, self.num_items, uid, iid)
