"""BPR-MF (Rendle et al.) behind the reference's class name and hooks
(model/ranking/BPR.py:8-104), trained on the MI355X.

Two execution modes, chosen by the optional conf key ``qrec.mode`` or env ``QREC_MODE``:

``exact`` (default)
    The reference's numpy-path semantics: negatives from the CPython ``random`` stream
    (bit-identical (u,i,j) sequence), triplets applied strictly in order by the
    order-exact kernel, fp64 tables (``QREC_DTYPE=f32`` for fp32).  Same loss, learning
    rate schedule, P and Q as the reference to rounding.
``throughput``
    Device Philox sampler + Hogwild kernel (fp32, exact per-sample deltas applied with
    atomic adds).  Same algorithm and sampling distribution, not the same random stream;
    judged on Recall@N.  ``QREC_SCHEDULE=auto`` (default), ``item`` (triplets visited item-major) or ``user`` (the
    reference's user-major visiting order): item-major wins when a few items collect most interactions (their rows
    would take the per-triplet atomics), user-major when popularity is flat (measured: 2.1 vs 1.6 G/s at the
    Zipf-0.6 Yelp2018 shape, 1.16 vs 1.21 G/s on a uniform 1 M-item catalogue); ``auto`` looks at max/mean item degree.
    ``QREC_P_UPDATE=auto`` (default), ``atomic`` or ``rmw`` (item-major only): how a user's row is written -- an atomic delta, or an
    sc1 load + store (one atomic row update per triplet instead of two: 0.63 vs 0.46 of the roofline on tables that live in HBM) that
    loses an update when two groups hold the same user at once; ``auto`` takes it only where that is rare (collision density
    <= 0.01: engine.resolve_p_update and the measurements in its comment) -- never at the reference's own dataset sizes.
"""
from __future__ import annotations

import os
import random

import numpy as np

from ... import capi
from ...base.iterativeRecommender import IterativeRecommender
from ...engine import BprSgd, DeviceTables


class BPR(IterativeRecommender):
    def __init__(self, conf, trainingSet=None, testSet=None, fold="[1]"):
        super().__init__(conf, trainingSet, testSet, fold)

    def readConfiguration(self):
        super().readConfiguration()
        mode = self.config["qrec.mode"] if self.config.contains("qrec.mode") else os.environ.get("QREC_MODE", "exact")
        if mode not in ("exact", "throughput"):
            print("parameter qrec.mode is invalid!")
            raise SystemExit(-1)
        self.mode = mode
        dt = os.environ.get("QREC_DTYPE", "f64" if mode == "exact" else "f32")
        self.table_dtype = np.float64 if (dt == "f64" and mode == "exact") else np.float32
        self.sampler_seed = int(os.environ.get("QREC_SEED", "0"))
        self.schedule = os.environ.get("QREC_SCHEDULE", "auto") if mode == "throughput" else "user"
        if self.schedule not in ("auto", "item", "user"):
            print("QREC_SCHEDULE must be auto, item or user")
            raise SystemExit(-1)
        self.p_update = os.environ.get("QREC_P_UPDATE", "auto") if mode == "throughput" else "atomic"
        if self.p_update not in ("auto", "atomic", "rmw"):
            print("QREC_P_UPDATE must be auto, atomic or rmw")
            raise SystemExit(-1)

    def initModel(self):
        super().initModel()

    def trainModel(self):
        print("Preparing item sets...")
        pos = self.data.positive_csr()           # PositiveSet, BPR.py:21-25
        u, i = pos.row_ids(), pos.indices
        dp = self.data_parallel()
        if dp is not None:       # one process per GPU: this rank trains its block of users (qrec_amd/dist.py)
            if self.mode != "throughput":
                print("exact mode is single-GPU only: run with QREC_MODE=throughput on several GPUs")
                raise SystemExit(-1)
            from ...dist import user_block
            lo, hi = user_block(pos.indptr.size - 1, dp.world, dp.rank)
            u, i = u[pos.indptr[lo]:pos.indptr[hi]], i[pos.indptr[lo]:pos.indptr[hi]]
            self.sampler_seed += 7919 * dp.rank
        print("training...")
        from ...engine import resolve_schedule
        schedule, _ = resolve_schedule(int(u.size), np.bincount(i, minlength=len(self.data.item)), self.schedule)
        n_items = len(self.data.item)
        layout = os.environ.get("QREC_DIST_MODE", "replicated")
        if layout not in ("replicated", "sharded"):
            print("QREC_DIST_MODE must be replicated or sharded")
            raise SystemExit(-1)
        if dp is not None and layout == "sharded":
            self._train_sharded(pos, schedule, n_items, dp)
            return
        tables = DeviceTables(self.P, self.Q, self.table_dtype)
        # several ranks, replicated tables: QREC_REPLICATED_SYNCS = K reconciliations of the replicas per epoch (default 0 = 1 up to two ranks,
        # 2 beyond -- dist.reconciliations_per_epoch: with ONE per epoch the paired Recall@20 runs leave the +-0.002 bar at 4 and 8 ranks)
        from ...dist import reconciliations_per_epoch
        syncs = reconciliations_per_epoch(dp.world, int(os.environ.get("QREC_REPLICATED_SYNCS", "0"))) if dp is not None else 1
        # the chunk the epoch is LAUNCHED with is the chunk the item-major list is dealt to the reconciliation batches in (the
        # constructor's default 32 against balanced_chunk's 26..40 at launch left launch chunks straddling the dealt ones)
        from ...engine import balanced_chunk
        sgd = BprSgd(tables, u, i, pos, schedule=schedule, batches=syncs, chunk=balanced_chunk(int(u.size)), p_update=self.p_update)
        epoch = 0
        if self.mode == "throughput" and (self.ranking.isMainOn() or dp is not None):
            self._train_throughput_pipelined(sgd, dp=dp)
            self.P, self.Q = tables.download(np.float64)
            return
        if self.mode == "throughput":
            sgd.prefetch_negatives_device(self.sampler_seed, 0)

        # Exact mode, pipelined: the reference draws epoch k's negatives when epoch k starts, from the global `random` stream,
        # and neither the draws nor the shuffle that closes an epoch (isConverged) look at the embeddings -- so the host side of
        # epoch k + 1 (the shuffle's draws replayed on a scratch permutation, the CPython-stream negatives, the list schedule,
        # the upload into the other schedule buffer on a side stream) runs while the kernel of epoch k executes.  The global
        # stream itself is only ever moved by the code that moves it in the reference: a prefetched draw is committed when its
        # epoch starts AND the stream is where the prefetch assumed it would be; otherwise it is dropped and redone in place.
        def draw(state):
            words = capi.state_from_python(state)
            jj = capi.mt_bpr_sample_epoch(words, pos.indptr, pos.indices, n_items)
            return jj, capi.state_to_python(words, state[2])

        def after_epoch_close(state):              # base/iterativeRecommender.py:101 shuffle(trainingData): its draws only
            words = capi.state_from_python(state)
            capi.mt_shuffle(words, self.data.elemCount(), np.arange(self.data.elemCount(), dtype=np.int64))
            return capi.state_to_python(words, state[2])

        pipelined = self.mode == "exact" and sgd.exact_width() > 1 and u.size > 0
        ahead = None
        if pipelined:
            side = capi.Stream()
        while epoch < self.maxEpoch:
            if pipelined:
                now = random.getstate()
                if ahead is None or ahead[0] != now:                                      # first epoch, or the stream is not where assumed
                    j, after = draw(now)
                    prep = sgd.prepare_ordered(j, slot=epoch & 1, stream=side.handle); side.sync()
                else:
                    _, j, after, prep = ahead
                random.setstate(after)
                sgd.run_prepared(prep, self.lRate, self.regU, self.regI)                  # enqueued; the host goes on
                ahead = None
                if epoch + 1 < self.maxEpoch:
                    start2 = after_epoch_close(after)
                    j2, after2 = draw(start2)
                    ahead = (start2, j2, after2, sgd.prepare_ordered(j2, slot=(epoch + 1) & 1, stream=side.handle)); side.sync()
            elif self.mode == "exact":
                state = random.getstate()
                words = capi.state_from_python(state)
                j = capi.mt_bpr_sample_epoch(words, pos.indptr, pos.indices, n_items)
                random.setstate(capi.state_to_python(words, state[2]))
                sgd.set_negatives(j)
                sgd.epoch_ordered(self.lRate, self.regU, self.regI)
            else:
                sgd.take_prefetched_negatives(epoch)
                sgd.epoch_throughput_async(self.lRate, self.regU, self.regI, chunk=sgd.launch_grid()[0], groups=sgd.launch_grid()[1])
                sgd.prefetch_negatives_device(self.sampler_seed, epoch + 1)   # overlaps the SGD kernel
            nll, sp, sq = sgd.epoch_stats()
            self.loss = nll + self.regU * sp + self.regI * sq
            epoch += 1
            if not self.ranking.isMainOn():      # isConverged then prints MAE/RMSE from the host tables (reference: live values)
                self.P, self.Q = tables.download(np.float64)
            if self.isConverged(epoch):
                break
        self.P, self.Q = tables.download(np.float64)

    def _train_sharded(self, pos, schedule, n_items, dp):
        """``QREC_DIST_MODE=sharded`` (BASELINE.json's north star layout, round 3 for the drop-in class): a rank holds ITS users'
        rows of P and its interleaved share of the item rows (item r * G + o = local row r of rank o); an epoch's batches fetch
        the distinct item rows they touch from the owners and return their updates (qrec_amd/dist.py ShardedItemExchange).
        Nothing is whole during training; the tables are assembled on every rank afterwards, because every rank evaluates
        (test users are sharded, the item side of a ranking is the whole catalogue)."""
        from ...dist import ShardedItemExchange, ShardedStep, agree_on_batches, shard_item_rows, shard_positive_csr
        from ...engine import balanced_chunk
        from ...interactions import CSR
        G, rank = dp.world, dp.rank
        lo, hi, lp, li = shard_positive_csr(pos.indptr, pos.indices, G, rank)
        u = np.repeat(np.arange(hi - lo, dtype=np.int32), np.diff(lp)).astype(np.int32)
        tables = DeviceTables(np.ascontiguousarray(self.P[lo:hi]), shard_item_rows(self.Q, G, rank), self.table_dtype)
        chunk = balanced_chunk(int(u.size))
        from ...dist import reconciliations_per_epoch
        n_batches = agree_on_batches(dp.control, int(u.size), 1 << 20, split_from=1 << 19,
                                     min_batches=reconciliations_per_epoch(G, int(os.environ.get("QREC_REPLICATED_SYNCS", "0"))))
        sgd = BprSgd(tables, u, li, CSR(lp, li), schedule=schedule, n_items=n_items, batches=n_batches, chunk=chunk, p_update=self.p_update)
        step = ShardedStep(dp.comm, ShardedItemExchange(dp.comm, n_items, tables.ld, tables.Q), n_batches)
        self._train_throughput_pipelined(sgd, dp=dp, step=step)
        P_loc, Q_loc = tables.download(np.float64)
        # assemble: user blocks in rank order, item rows interleaved (host control plane; sizes differ by at most one row)
        d = P_loc.shape[1]
        rows_p, rows_q = -(-len(self.data.user) // G), -(-n_items // G)
        pad = lambda a, r: np.concatenate([a, np.zeros((r - a.shape[0], d))]) if a.shape[0] < r else a
        allP, allQ = dp.control.allgather_host(pad(P_loc, rows_p)), dp.control.allgather_host(pad(Q_loc, rows_q))
        from ...dist import user_block
        self.P = np.concatenate([allP[r][:user_block(len(self.data.user), G, r)[1] - user_block(len(self.data.user), G, r)[0]] for r in range(G)])
        Q = np.empty((n_items, d))
        for r in range(G):
            Q[r::G] = allQ[r][:len(range(r, n_items, G))]
        self.Q = Q

    def _train_throughput_pipelined(self, sgd, depth: int = 3, dp=None, step=None):
        """Throughput mode without a host round trip per epoch: the epoch's loss (BPR.py:40,53), isConverged and
        updateLearningRate (base/iterativeRecommender.py:56-63,88-104) run on the device (qrec_epoch_close); the
        host enqueues epochs ``depth`` ahead and prints the reference's per-epoch line from the device log as the
        epochs retire.  Epochs enqueued past the converged one are no-ops on the device, so the tables are
        those of the converged epoch exactly as if the loop had stopped there.  (The reference's per-epoch
        ``shuffle(trainingData)`` has no effect on this model's visiting order and is not replayed here.)
        ``dp`` (one process per GPU): every rank keeps both tables whole and trains its own users' triplets; after the
        SGD kernel the replicas are reconciled by summing the ranks' deltas (users' rows: disjoint, exact; item rows:
        every rank's updates kept -- qrec_amd/dist.py), sum(-log sigma) is added over the ranks, and every rank's
        device-side driver then takes the same decision on identical tables."""
        chunk, groups = sgd.launch_grid()      # the chunk the stored order was dealt in; one launch per epoch: >= 8 rounds of the grid (engine.grid_for_epoch)
        sharded = step is not None
        if dp is not None and step is None:
            from ...dist import ReplicatedStep, ReplicatedTableSync
            step = ReplicatedStep(dp.comm, ReplicatedTableSync(dp.comm, sgd.t.Q), ReplicatedTableSync(dp.comm, sgd.t.P))
        sgd.start_device_driver(self.lRate, log_capacity=self.maxEpoch)
        sgd.prefetch_negatives_device(self.sampler_seed, 0)
        closed = []
        # several ranks: an explicit stream -- the null stream synchronises implicitly with the communicator's own
        # streams (60 us per epoch measured, bench.py)
        stream = capi.Stream() if dp is not None else None
        capi.device_sync()                  # set-up work (uploads, clears) sits on the null stream

        def retire(k):
            """print epoch k+1 once the device has closed it; True when training is over"""
            closed[k].sync()
            st = sgd.driver_state(stream)
            if st["failed"]:
                print("Loss = NaN or Infinity: current settings does not fit the recommender! Change the settings and try again!")
                raise SystemExit(-1)
            if st["epochs"] <= k:            # an earlier epoch converged: this one never ran
                return True
            loss, lr_used, _, delta = sgd.d_log.numpy(stream)[k, :4]
            self.loss, self.lastLoss = float(loss), float(loss)
            print("%s %s epoch %d: loss = %.4f, delta_loss = %.5f learning_Rate = %.5f"
                  % (self.modelName, self.foldInfo, k + 1, loss, delta, lr_used))
            return st["converged"] and st["epochs"] == k + 1

        done, retired = False, 0
        for epoch in range(self.maxEpoch):
            sgd.take_prefetched_negatives(epoch, stream)
            if sharded:        # plan (adopting the one begun inside the previous epoch), then the batches; the next epoch's sampler is
                step.prepare(sgd, stream)         # enqueued from inside, in front of the first SGD grid (engine.epoch_device_async)
                sgd.epoch_device_async(self.regU, self.regI, self.maxLRate, tol=1e-3, chunk=chunk, dist=step, stream=stream,
                                       after_start=lambda e=epoch: sgd.prefetch_negatives_device(self.sampler_seed, e + 1))
            else:
                sgd.epoch_device_async(self.regU, self.regI, self.maxLRate, tol=1e-3, chunk=chunk, dist=step, stream=stream, groups=groups)
                sgd.prefetch_negatives_device(self.sampler_seed, epoch + 1)      # released under this epoch's SGD kernel
            ev = capi.Event(); ev.record(stream); closed.append(ev)
            if epoch >= depth:
                done = retire(retired); retired += 1
                if done:
                    break
        while not done and retired < len(closed):
            done = retire(retired); retired += 1
        if stream is not None:
            stream.sync()
        self.lRate = sgd.driver_state(stream)["lr"]

    def trainModel_tf(self):
        """The reference's TensorFlow variant (model/ranking/BPR.py:77-96), taken when the conf has
        ``-tf``: truncated-normal(0.005) tables (base/iterativeRecommender.py:47-48), batches are
        consecutive slices of ``trainingData`` in its current order (``next_batch`` does NOT shuffle,
        BPR.py:55-65) with one negative per row drawn from the CPython stream, Adam at a constant
        learning rate, no convergence test."""
        from ...base.deepRecommender import truncated_normal
        from ...capi import DeviceBuffer
        from ...graph import BprTfTrainer, ordered_reductions
        self.batch_size = int(self.config["batch_size"])
        U0 = truncated_normal((self.num_users, self.emb_size), 0.005)
        V0 = truncated_normal((self.num_items, self.emb_size), 0.005)
        with ordered_reductions():          # the -tf path replays the CPython stream: always the parity mode (bit-reproducible gradient sums)
            tr = self._tf_trainer = BprTfTrainer(U0, V0, self.lRate, self.regU)
        rated = self.data.rated_csr().sorted_rows()
        quiet = os.environ.get("QREC_QUIET") == "1"
        for epoch in range(self.maxEpoch):
            u, i, _ = self.data.training_arrays()
            state = random.getstate()
            words = capi.state_from_python(state)
            j = capi.mt_pairwise_sample_epoch(words, u, rated.indptr, rated.indices, self.num_items)
            random.setstate(capi.state_to_python(words, state[2]))
            d_u, d_i, d_j = DeviceBuffer.from_numpy(u), DeviceBuffer.from_numpy(i), DeviceBuffer.from_numpy(j)
            for n, s in enumerate(range(0, u.size, self.batch_size)):
                B = min(self.batch_size, u.size - s)
                tr.train_step_async(d_u.ptr + 4 * s, d_i.ptr + 4 * s, d_j.ptr + 4 * s, B)
                if not quiet:
                    print("training:", epoch + 1, "batch", n, "loss:", tr.loss())
        self.P, self.Q = tr.tables()

    def predictForRanking(self, u):
        if self.data.containsUser(u):
            return self.Q.dot(self.P[self.data.getUserId(u)])
        return [self.data.globalMean] * self.num_items
