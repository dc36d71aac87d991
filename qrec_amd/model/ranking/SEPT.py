"""SEPT behind the reference's class name and hooks (model/ranking/SEPT.py:19-323): socially-aware self-supervised
tri-training.  Four LightGCN-structured views with per-layer l2-normalisation -- friends, item sharing, user-item
preference, and a per-epoch perturbed joint graph -- the first third of the epochs train the recommendation task
alone, the rest add the neighbour-discrimination loss whose positives are the other two encoders' top-k pseudo
labels.  Needs the ``social`` file (``social.setup``); evaluated every epoch, best epoch kept."""
from __future__ import annotations

import os
import random

import numpy as np
import scipy.sparse as sp

from ... import capi
from ...base.deepRecommender import truncated_normal
from ...base.graphRecommender import GraphRecommender
from ...base.socialRecommender import SocialRecommender
from ...capi import DeviceBuffer
from ...graph import SEPTTrainer, sept_perturbed_adjacency, sept_user_views, unique_first_appearance
from ...util import config


class SEPT(SocialRecommender, GraphRecommender):
    def __init__(self, conf, trainingSet=None, testSet=None, relation=None, fold="[1]"):
        # one cooperative chain: SocialRecommender -> GraphRecommender -> DeepRecommender -> IterativeRecommender
        SocialRecommender.__init__(self, conf, trainingSet, testSet, relation if relation is not None else [], fold)

    def readConfiguration(self):
        super().readConfiguration()
        args = config.OptionConf(self.config["SEPT"])
        self.n_layers = int(args["-n_layer"])
        self.ss_rate = float(args["-ss_rate"])
        self.drop_rate = float(args["-drop_rate"])
        self.instance_cnt = int(args["-ins_cnt"])

    def initModel(self):
        super().initModel()
        uid, iid, _ = self.data.training_arrays()
        fo, fe = self.relation_ids()
        friend, sharing = sept_user_views(self.num_users, self.num_items, uid, iid, fo, fe)
        indptr, indices, values = self.create_joint_sparse_adjaceny()
        n = self.num_users + self.num_items
        adj = sp.csr_matrix((values, indices, indptr), shape=(n, n))
        # the reference re-creates both variables here (SEPT.py:129-130); every view starts from Variable / 2
        self.user_embeddings = truncated_normal((self.num_users, self.emb_size), 0.005)
        self.item_embeddings = truncated_normal((self.num_items, self.emb_size), 0.005)
        self.trainer = self.build_trainer(SEPTTrainer, self.user_embeddings, self.item_embeddings, adj, friend, sharing, self.n_layers, self.lRate,
                                   self.regU, self.ss_rate, self.instance_cnt, max_unique=max(self._step_rows(), 64))
        self._epochs_drawn = 0

    def _step_rows(self) -> int:
        """rows of the batch stream one training step covers: batch_size, times the world size in a multi-GPU run"""
        dp = self.data_parallel()
        return self.batch_size * (dp.world if dp else 1)

    def get_adj_mat(self, is_subgraph=False):
        """scipy CSR of the (perturbed) joint graph (SEPT.py:79-114); the perturbed one consumes the CPython generator
        exactly as the reference's two random.sample calls do, on the CURRENT trainingData order."""
        uid, iid, _ = self.data.training_arrays()
        fo, fe = self.relation_ids()
        state = random.getstate()
        words = capi.state_from_python(state)
        M = sept_perturbed_adjacency(words, self.num_users, self.num_items, uid, iid, fo, fe, self.drop_rate if is_subgraph else 0.0)
        random.setstate(capi.state_to_python(words, state[2]))
        return M

    def _draw_epoch(self):
        """one epoch's host-side randomness in the reference's order (SEPT.py:274-301), on the sampler thread: the
        perturbed graph when the epoch trains jointly (epoch > maxEpoch / 3), then shuffle + negatives, then tf.unique
        of every batch's users."""
        epoch = self._epochs_drawn
        self._epochs_drawn += 1
        joint = epoch > self.maxEpoch / 3
        sub = self.get_adj_mat(is_subgraph=True) if joint else None
        u, i, j = self.sample_epoch_pairwise()
        step = self._step_rows()
        starts = list(range(0, u.size, step))
        uu = [unique_first_appearance(u[s:s + step]) for s in starts] if joint else [np.zeros(0, np.int32) for _ in starts]
        off = np.concatenate([[0], np.cumsum([x.size for x in uu])])
        return sub, u, i, j, starts, np.concatenate(uu).astype(np.int32) if uu else np.zeros(0, np.int32), off

    def saveModel(self):
        self.bestU, self.bestV = self.U, self.V

    def trainModel(self):
        quiet = os.environ.get("QREC_QUIET") == "1"
        tr = self.trainer
        dp = tr.dp = self.data_parallel()
        step = self._step_rows()
        for epoch, (sub, u, i, j, starts, uu, off) in enumerate(self.iter_epoch_samples(self.maxEpoch, self._draw_epoch)):
            joint = sub is not None
            if joint:
                tr.set_perturbed_graph(sub)
            d_u, d_i, d_j = DeviceBuffer.from_numpy(u), DeviceBuffer.from_numpy(i), DeviceBuffer.from_numpy(j)
            d_uu = DeviceBuffer.from_numpy(uu) if uu.size else None
            for n, s in enumerate(starts):
                B = min(step, u.size - s)
                n_uu = int(off[n + 1] - off[n])
                if joint and n_uu < self.instance_cnt:
                    print("SEPT: a batch with fewer distinct users than -ins_cnt cannot be pseudo-labelled")
                    raise SystemExit(-1)
                tr.train_step_async(d_u.ptr + 4 * s, d_i.ptr + 4 * s, d_j.ptr + 4 * s, B, joint,
                                    d_uu.ptr + 4 * int(off[n]) if joint else None, n_uu, share=self.step_share(dp, B) if dp else None)
                if not quiet:
                    rec_l, con_l = tr.losses()
                    if joint:
                        print(self.foldInfo, "training:", epoch + 1, "batch", n, "rec loss:", rec_l, "con_loss:", con_l)
                    else:
                        print(self.foldInfo, "training:", epoch + 1, "batch", n, "rec loss:", rec_l)
            self.U, self.V = tr.rec_embeddings()
            self.ranking_performance(epoch)
        self.U, self.V = self.bestU, self.bestV

    def predictForRanking(self, u):
        if self.data.containsUser(u):
            return self.V.dot(self.U[self.data.getUserId(u)])
        return [self.data.globalMean] * self.num_items
