"""Deep (batch-trained) recommenders: fp32 embedding tables and the pairwise batch sampler of
the reference's ``DeepRecommender`` (base/deepRecommender.py:5-83).  There is no TensorFlow
graph/session here; the tables live on the host until a model's trainer moves them to HBM."""
from __future__ import annotations

import random

import numpy as np

from .. import capi
from .iterativeRecommender import IterativeRecommender


def truncated_normal(shape, stddev: float) -> np.ndarray:
    """tf.truncated_normal(stddev): N(0, stddev^2) re-drawn while |x| > 2 stddev
    (base/deepRecommender.py:21-22).  TF's own generator is not reproducible outside TF, so
    this draws from numpy's global RNG; parity tests inject the initial tables instead."""
    out = np.random.standard_normal(shape)
    bad = np.abs(out) > 2
    while bad.any():
        out[bad] = np.random.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2
    return (out * stddev).astype(np.float32)


class DeepRecommender(IterativeRecommender):
    def __init__(self, conf, trainingSet, testSet, fold="[1]"):
        super().__init__(conf, trainingSet, testSet, fold)

    def readConfiguration(self):
        super().readConfiguration()
        self.batch_size = int(self.config["batch_size"])

    def initModel(self):
        super().initModel()
        self.user_embeddings = truncated_normal((self.num_users, self.emb_size), 0.005)
        self.item_embeddings = truncated_normal((self.num_items, self.emb_size), 0.005)

    def sample_epoch_pairwise(self):
        """One epoch of ``next_batch_pairwise`` (base/deepRecommender.py:29-52) as three int32
        arrays in visiting order: ``shuffle(trainingData)``, then one negative per row, with
        the exact CPython draw sequence (done natively; the Python generator stays in
        lock-step).  Batches are consecutive ``batch_size`` slices; the last one is short."""
        self.shuffle_training_data()
        u, i, _ = self.data.training_arrays()
        rated = self._rated_sorted()
        state = random.getstate()
        words = capi.state_from_python(state)
        j = capi.mt_pairwise_sample_epoch(words, u, rated.indptr, rated.indices, self.num_items)
        random.setstate(capi.state_to_python(words, state[2]))
        return u, i, j

    def iter_epoch_samples(self, n_epochs: int, draw=None):
        """Yield ``draw()`` (default ``sample_epoch_pairwise``) for epochs 0..n_epochs-1, computing epoch
        k+1 on a worker thread while the caller trains on epoch k.  The draw sequence does not depend on
        the model, so running it ahead leaves the CPython stream exactly where the reference's would be;
        nothing is drawn beyond the last epoch.
        The draw reads and writes process-global state -- the ``random`` generator and the order of
        ``self.data.trainingData`` -- so it holds ``self.sampling_lock`` while it runs.  Code on the training thread that
        touches either between two epochs (a subclass hook calling ``random``, ``training_arrays()``, ...) must take the
        same lock; what it then sees is the state one epoch AHEAD of the reference's at that point (the draws of the
        next epoch have been consumed already): the stream as a whole is the reference's, its interleaving with such
        calls is not.  The drop-in models make no such calls."""
        import threading
        from concurrent.futures import ThreadPoolExecutor
        inner = draw or self.sample_epoch_pairwise
        if n_epochs <= 0:
            return
        lock = self.sampling_lock = getattr(self, "sampling_lock", None) or threading.RLock()

        def draw():
            with lock:
                return inner()
        with ThreadPoolExecutor(max_workers=1) as pool:
            pending = pool.submit(draw)
            for epoch in range(n_epochs):
                sample = pending.result()
                if epoch + 1 < n_epochs:
                    pending = pool.submit(draw)
                yield sample

    def next_batch_pairwise(self):
        """Generator with the reference's signature: yields (u_idx, i_idx, j_idx) lists."""
        u, i, j = self.sample_epoch_pairwise()
        for s in range(0, u.size, self.batch_size):
            e = min(s + self.batch_size, u.size)
            yield u[s:e].tolist(), i[s:e].tolist(), j[s:e].tolist()

    def _rated_sorted(self):
        if not hasattr(self, "_rated_sorted_csr"):
            self._rated_sorted_csr = self.data.rated_csr().sorted_rows()
        return self._rated_sorted_csr
