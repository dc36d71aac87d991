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

    def build_trainer(self, cls, *args, **kwargs):
        """exact mode (the default) = parity mode: the trainer adds its batch gradients in the reference's CPU order (csrc/ordered.hip),
        bit-reproducible from run to run; throughput mode keeps the float atomics"""
        import os
        from ..graph import ordered_reductions
        forced = os.environ.get("QREC_REDUCTIONS")            # "ordered" / "atomic": override the mode's choice (measurements)
        with ordered_reductions(forced == "ordered" if forced in ("ordered", "atomic") else not self.throughput_mode()):
            return cls(*args, **kwargs)

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

    # ---- throughput mode (QREC_MODE=throughput): the epoch's batch stream drawn on the device ------------------------
    def throughput_mode(self) -> bool:
        import os
        return (self.config["qrec.mode"] if self.config.contains("qrec.mode") else os.environ.get("QREC_MODE", "exact")) == "throughput"

    def iter_epoch_samples_device(self, n_epochs: int, stream=None):
        """``next_batch_pairwise`` without the host: per epoch a uniform shuffle of the training rows (one device sort of
        Philox keys), the rows gathered into that order, one negative per row by rejection against the user's rated
        items (Philox, counter = row).  Same distribution as the reference's shuffle + choice loop
        (base/deepRecommender.py:29-52), not its CPython stream -- judged on the measures, like BPR's throughput mode.
        Yields DEVICE buffers (d_u, d_i, d_j); epoch k + 1 is drawn on a side stream while the caller trains on epoch k.
        Python's generator is not consumed.
        ``stream``: the stream the caller's training steps are enqueued on (None = the null stream).  Contract: all steps that
        read an epoch's buffers are enqueued on THAT stream before the generator is advanced -- the hand-off (the training
        stream waits for the draw; the draw of epoch k + 1 into the other buffer waits for epoch k - 1's readers) is ordered
        with events recorded on it."""
        import os
        from ..capi import DeviceBuffer
        if n_epochs <= 0:
            return
        u, i, _ = self.data.training_arrays()
        n = int(u.size)
        seed = int(os.environ.get("QREC_SEED", "0"))
        rated = self._rated_sorted()
        d_ptr, d_items = DeviceBuffer.from_numpy(rated.indptr.astype(np.int64)), DeviceBuffer.from_numpy(rated.indices.astype(np.int32))
        d_u0, d_i0 = DeviceBuffer.from_numpy(u.astype(np.int32)), DeviceBuffer.from_numpy(i.astype(np.int32))
        scratch = DeviceBuffer(capi.random_permutations_scratch_bytes(n, 1), np.uint8)
        d_perm = DeviceBuffer(max(n, 1), np.int32)
        side, bufs, ready = capi.Stream(), [], []
        for _ in range(2):
            bufs.append(tuple(DeviceBuffer(max(n, 1), np.int32) for _ in range(3))); ready.append(capi.Event())

        def draw(epoch):
            du, di, dj = bufs[epoch % 2]
            capi.random_permutations(n, 1, seed, 2 * epoch, scratch, d_perm, None, side)
            capi.gather_pairs(d_perm, d_u0, d_i0, n, du, di, side)
            capi.philox_bpr_sample(d_ptr, d_items, du, n, self.num_items, seed, 2 * epoch + 1, dj, side)
            ready[epoch % 2].record(side)
        draw(0)
        for epoch in range(n_epochs):
            capi.stream_wait_event(stream, ready[epoch % 2])        # the training stream waits on the device, not the host
            if epoch + 1 < n_epochs:
                done = capi.Event(); done.record(stream)             # epoch k-1's steps (readers of the other buffer) are enqueued before this
                capi.stream_wait_event(side, done)
                draw(epoch + 1)
            yield bufs[epoch % 2]

    def iter_epoch_device_samples(self, n_epochs: int, stream=None):
        """(n_rows, d_u, d_i, d_j) per epoch, resident on the device: the reference's CPython batch stream replayed on the
        host one epoch ahead and uploaded (exact mode, default), or drawn on the device (QREC_MODE=throughput)."""
        from ..capi import DeviceBuffer
        if self.throughput_mode():
            n = int(self.data.training_arrays()[0].size)
            for d_u, d_i, d_j in self.iter_epoch_samples_device(n_epochs, stream):
                yield n, d_u, d_i, d_j
        else:
            for u, i, j in self.iter_epoch_samples(n_epochs):
                yield int(u.size), DeviceBuffer.from_numpy(u), DeviceBuffer.from_numpy(i), DeviceBuffer.from_numpy(j)

    def next_batch_pairwise(self):
        """Generator with the reference's signature: yields (u_idx, i_idx, j_idx) lists."""
        u, i, j = self.sample_epoch_pairwise()
        for s in range(0, u.size, self.batch_size):
            e = min(s + self.batch_size, u.size)
            yield u[s:e].tolist(), i[s:e].tolist(), j[s:e].tolist()

    def sample_epoch_pointwise(self, negatives: int = 4):
        """One pass of ``next_batch_pointwise`` (base/deepRecommender.py:54-77) as three int32 arrays of length
        (1 + negatives) * n: per training row -- in stored order, this sampler does not shuffle -- the positive (label 1), then
        ``negatives`` items drawn with ``randint(0, num_items - 1)`` and redrawn while the user rated them (label 0).
        ``randint(0, n - 1)`` and ``choice`` of an n-list both come down to ``_randbelow(n)``, so the native replay of the
        pairwise draw loop over every row repeated ``negatives`` times IS this stream; the Python generator stays in lock-step."""
        u, i, _ = self.data.training_arrays()
        rated = self._rated_sorted()
        state = random.getstate()
        words = capi.state_from_python(state)
        neg = capi.mt_pairwise_sample_epoch(words, np.repeat(u, negatives), rated.indptr, rated.indices, self.num_items)
        random.setstate(capi.state_to_python(words, state[2]))
        n, w = int(u.size), 1 + negatives
        uu = np.repeat(u, w).astype(np.int32)
        ii = np.empty(n * w, np.int32); ii.reshape(n, w)[:, 0] = i; ii.reshape(n, w)[:, 1:] = neg.reshape(n, negatives)
        y = np.zeros(n * w, np.int32); y[::w] = 1
        return uu, ii, y

    def next_batch_pointwise(self):
        """Generator with the reference's signature: yields (u_idx, i_idx, y) lists, 5 entries per training row, batches of
        ``batch_size`` ROWS (base/deepRecommender.py:54-77)."""
        uu, ii, y = self.sample_epoch_pointwise(4)
        step = 5 * self.batch_size
        for s in range(0, uu.size, step):
            e = min(s + step, uu.size)
            yield uu[s:e].tolist(), ii[s:e].tolist(), y[s:e].tolist()

    def _rated_sorted(self):
        if not hasattr(self, "_rated_sorted_csr"):
            self._rated_sorted_csr = self.data.rated_csr().sorted_rows()
        return self._rated_sorted_csr
