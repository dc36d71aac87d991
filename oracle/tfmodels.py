"""numpy/scipy restatement of the reference's TensorFlow-1.14 graph models (LightGCN first).
TEST INFRASTRUCTURE ONLY (see oracle/qrec_oracle.c header).

PARITY STATUS: **pinned to runs of the reference's own model classes, executed through a stand-in for the tensorflow
module.**  tensorflow==1.14.0 (README.md:57) cannot be installed here (no cp310 wheel, no network).  tests/golden/tf1shim.py
provides the TF-1.14 API names the reference's eight TF-path classes call, as a lazy graph evaluated with torch (float32,
autograd); tests/golden/gen_golden_tf.py runs model/ranking/{BPR (trainModel_tf), LightGCN, NGCF, SimGCL, SGL, BUIR, SEPT,
MHCN}.py unmodified through it and commits initial / final variables, batches, printed losses, scoring tables and the keys
of every random draw; tests/test_oracle_tf_golden.py holds every class below to those runs, step by step.  What remains a
restatement of published semantics is each primitive op (and TF's float32 summation orders, inside the tolerances); the
model graphs are the reference's code.  Also cross-checked against torch-CPU autograd written independently
(tests/test_oracle_tfmodels.py).  Pinned to the live reference as well: the joint normalized adjacency
(base/graphRecommender.py:10-29 is pure scipy -> tests/golden/pairwise_adj_filmtrust.npz) and the batch sampler stream
(base/deepRecommender.py:29-52 -> same fixture, via oracle/qrec_oracle.c).

TF 1.14 op semantics used (third-party, not vendored):
  tf.sparse_tensor_dense_matmul(A, X)      = A @ X                     (fp32)
  tf.reduce_mean([E0..EL], axis=0)         = (E0 + ... + EL) / (L+1)
  tf.nn.embedding_lookup(E, idx)           = E[idx]; gradient = scatter-add (duplicates sum)
  tf.nn.l2_loss(x)                         = sum(x**2) / 2
  tf.sigmoid, tf.log                       = elementwise fp32
  tf.train.AdamOptimizer(lr) (beta1=.9, beta2=.999, epsilon=1e-8): see AdamTF114 below
      (ApplyAdam functor form, all fp32)
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def joint_norm_adjacency(n_users: int, n_items: int, uid: np.ndarray, iid: np.ndarray) -> sp.csr_matrix:
    """base/graphRecommender.py:10-29: R = ones at (u, U+i) (duplicates sum), A = R + R^T,
    d = rowsum^-1/2 (inf -> 0), A_hat = diag(d) A diag(d), all float32, CSR."""
    n = n_users + n_items
    ones = np.ones_like(uid, dtype=np.float32)
    R = sp.csr_matrix((ones, (uid, iid + n_users)), shape=(n, n))
    A = R + R.T
    rowsum = np.array(A.sum(1))
    with np.errstate(divide="ignore"):
        d_inv = np.power(rowsum, -0.5).flatten()
    d_inv[np.isinf(d_inv)] = 0.0
    D = sp.diags(d_inv)
    return D.dot(A).dot(D).tocsr()


def philox_subgraph_rows(n_users: int, n_items: int, uid, iid, aug_type: int, drop_rate: float, seed: int, stream_id: int) -> np.ndarray:
    """The training rows a device-drawn sub-graph keeps (throughput mode of SGL / BUIR; qrec_amd.graph.SubgraphSampler's contract,
    restated): the subsets of SGL.py:118-130 / BUIR.py:46 -- random.sample of int(U rate) users and int(I rate) items to drop, or of
    int(E (1 - rate)) rows to keep -- taken as the first K entries of the Philox permutation of (seed, stream_id) [items: stream_id + 1]
    instead of CPython's stream.  Returns the kept row indices in ascending order."""
    from . import c as O
    uid, iid = np.asarray(uid), np.asarray(iid)
    if drop_rate <= 0:
        return np.arange(uid.size)
    if aug_type == 0:
        drop_u = O.philox_permutation(n_users, seed, stream_id)[:int(n_users * drop_rate)]
        drop_i = O.philox_permutation(n_items, seed, stream_id + 1)[:int(n_items * drop_rate)]
        ku = np.ones(n_users, bool); ku[drop_u] = False
        ki = np.ones(n_items, bool); ki[drop_i] = False
        return np.flatnonzero(ku[uid] & ki[iid])
    if aug_type in (1, 2):
        return np.sort(O.philox_permutation(uid.size, seed, stream_id)[:int(uid.size * (1 - drop_rate))])
    raise ValueError(aug_type)


def subgraph_values_on_full_structure(n_users: int, n_items: int, uid, iid, kept_rows) -> np.ndarray:
    """The reference's re-normalised sub-adjacency of the kept training rows (SGL.py:131-155: R' + R'^T, d = rowsum^-1/2, inf -> 0,
    diag(d) A diag(d), float32, scipy CSR) as a value array over the FULL graph's sorted CSR structure -- the layout the device
    augmentation writes (csrc/augment.hip): kept entries carry the sub-graph's values, every other entry of the full graph is 0."""
    uid, iid = np.asarray(uid), np.asarray(iid)
    full = joint_norm_adjacency(n_users, n_items, uid, iid)
    with np.errstate(divide="ignore"):
        sub = joint_norm_adjacency(n_users, n_items, uid[kept_rows], iid[kept_rows])
    full.sort_indices(); sub.sort_indices()
    n = n_users + n_items
    fk = np.repeat(np.arange(n, dtype=np.int64), np.diff(full.indptr)) * n + full.indices
    sk = np.repeat(np.arange(n, dtype=np.int64), np.diff(sub.indptr)) * n + sub.indices
    out = np.zeros(fk.size, np.float32)
    pos = np.searchsorted(fk, sk)
    assert np.array_equal(fk[pos], sk)
    out[pos] = sub.data
    return out


def bpr_batch_loss_and_grads(u_b, i_b, j_b, reg, eps=np.float32(1e-7)):
    """util/loss.py:3-6 (eps = 10e-8) + the batch L2 term of model/ranking/LightGCN.py:28-30;
    returns (loss, du_b, di_b, dj_b) -- hand-derived gradients, fp32."""
    f = np.float32
    score = (u_b * i_b).sum(1, dtype=f) - (u_b * j_b).sum(1, dtype=f)
    s = (f(1) / (f(1) + np.exp(-score, dtype=f))).astype(f)
    loss = -np.log(s + eps, dtype=f).sum(dtype=np.float64)
    loss += float(reg) * 0.5 * float((u_b.astype(np.float64) ** 2).sum() + (i_b.astype(np.float64) ** 2).sum()
                                     + (j_b.astype(np.float64) ** 2).sum())
    g = (-(s * (f(1) - s)) / (s + eps)).astype(f)[:, None]       # d loss / d score
    du = g * (i_b - j_b) + f(reg) * u_b
    di = g * u_b + f(reg) * i_b
    dj = -g * u_b + f(reg) * j_b
    return loss, du.astype(f), di.astype(f), dj.astype(f)


class AdamTF114:
    """tf.train.AdamOptimizer's dense update in the form of TF 1.14's ApplyAdam functor
    (tensorflow/core/kernels/training_ops.cc), everything in the variable's dtype (fp32):
        alpha = lr * sqrt(1 - beta2_power) / (1 - beta1_power)
        m += (g - m) * (1 - beta1);  v += (g*g - v) * (1 - beta2)
        var -= (m * alpha) / (sqrt(v) + epsilon)
    beta{1,2}_power are fp32 variables that start at beta and are multiplied by beta after every
    step (Adam._finish), i.e. beta^t by repeated fp32 multiplication."""

    def __init__(self, shape, lr, beta1=0.9, beta2=0.999, eps=1e-8, dtype=np.float32):
        f = self.dtype = dtype
        self.m = np.zeros(shape, dtype); self.v = np.zeros(shape, dtype)
        self.lr, self.b1, self.b2, self.eps = f(lr), f(beta1), f(beta2), f(eps)
        self.b1p, self.b2p = f(beta1), f(beta2)
        self.t = 0

    def alpha(self):
        f = self.dtype
        return f(self.lr * np.sqrt(f(1) - self.b2p, dtype=f) / (f(1) - self.b1p))

    def step(self, theta, g):
        f = self.dtype
        alpha = self.alpha()
        self.m += (g - self.m) * (f(1) - self.b1)
        self.v += (g * g - self.v) * (f(1) - self.b2)
        theta -= (self.m * alpha) / (np.sqrt(self.v) + self.eps)
        self.b1p = f(self.b1p * self.b1); self.b2p = f(self.b2p * self.b2)
        self.t += 1


class LightGCN:
    """model/ranking/LightGCN.py:11-41 restated: L propagation layers over the joint
    adjacency, mean over [E0..EL], batch BPR loss + batch L2, dense Adam on [U;V]."""

    def __init__(self, U0, V0, adj: sp.csr_matrix, n_layers: int, lr: float, reg: float):
        self.nu, self.ni = U0.shape[0], V0.shape[0]
        self.E = np.concatenate([U0, V0]).astype(np.float32)      # ego embeddings [U;V]
        self.adj = adj.astype(np.float32).tocsr()
        self.L, self.reg = n_layers, reg
        self.opt = AdamTF114(self.E.shape, lr)

    def propagate(self, E=None):
        E = self.E if E is None else E
        layers = [E]
        for _ in range(self.L):
            layers.append(self.adj.dot(layers[-1]).astype(np.float32))     # LightGCN.py:17
        acc = layers[0].copy()
        for x in layers[1:]:
            acc += x
        return (acc / np.float32(self.L + 1)).astype(np.float32)           # LightGCN.py:19

    def loss_and_grad(self, u_idx, i_idx, j_idx):
        Ebar = self.propagate()
        ui, ii, ji = np.asarray(u_idx), np.asarray(i_idx) + self.nu, np.asarray(j_idx) + self.nu
        loss, du, di, dj = bpr_batch_loss_and_grads(Ebar[ui], Ebar[ii], Ebar[ji], self.reg)
        dEbar = np.zeros_like(Ebar)
        np.add.at(dEbar, ui, du); np.add.at(dEbar, ii, di); np.add.at(dEbar, ji, dj)
        # back through the mean and the L SpMMs (A_hat is symmetric): dE0 = sum_k A^k (dEbar/(L+1))
        c = dEbar / np.float32(self.L + 1)
        G = c.copy()
        for _ in range(self.L):
            G = (c + self.adj.T.dot(G)).astype(np.float32)
        return loss, G

    def train_step(self, u_idx, i_idx, j_idx):
        loss, g = self.loss_and_grad(u_idx, i_idx, j_idx)
        self.opt.step(self.E, g)
        return loss

    def final_embeddings(self):
        Ebar = self.propagate()
        return Ebar[:self.nu], Ebar[self.nu:]                               # LightGCN.py:41


# ======================================================================================
# SimGCL  (model/ranking/SimGCL.py:15-118)
# ======================================================================================
def l2_normalize_rows(x, eps=np.float32(1e-12)):
    """tf.nn.l2_normalize(x, 1) = x * rsqrt(max(sum(x^2), 1e-12)); returns (z, inv_norm)."""
    ss = (x * x).sum(1, dtype=np.float32)
    inv = (np.float32(1) / np.sqrt(np.maximum(ss, eps), dtype=np.float32)).astype(np.float32)
    return (x * inv[:, None]).astype(np.float32), inv


def info_nce_loss_and_grads(x1, x2, tau=np.float32(0.2)):
    """One side (users or items) of SimGCL.calc_cl_loss (SimGCL.py:60-90) on already gathered
    rows x1, x2 [n, d] of the two perturbed views:
        z = l2_normalize(x);  pos = exp(sum(z1*z2)/tau);  ttl = sum_cols exp(z1 z2^T / tau)
        loss = -sum log(pos / ttl)
    Returns (loss, dx1, dx2) with hand-derived gradients (fp32)."""
    f = np.float32
    z1, r1 = l2_normalize_rows(x1); z2, r2 = l2_normalize_rows(x2)
    S = (z1 @ z2.T).astype(f) / f(tau)
    Ex = np.exp(S, dtype=f)
    ttl = Ex.sum(1, dtype=f)
    pos = np.exp((z1 * z2).sum(1, dtype=f) / f(tau), dtype=f)
    loss = float(-np.log(pos / ttl, dtype=f).sum(dtype=np.float64))
    G = (Ex / ttl[:, None]).astype(f)                       # row softmax
    G[np.arange(G.shape[0]), np.arange(G.shape[0])] -= f(1)
    dz1 = (G @ z2).astype(f) / f(tau)
    dz2 = (G.T @ z1).astype(f) / f(tau)
    # back through z = x * inv:  dx = (dz - z (z.dz)) * inv     (clamp inactive for non-zero rows)
    dx1 = ((dz1 - z1 * (z1 * dz1).sum(1, dtype=f)[:, None]) * r1[:, None]).astype(f)
    dx2 = ((dz2 - z2 * (z2 * dz2).sum(1, dtype=f)[:, None]) * r2[:, None]).astype(f)
    return loss, dx1, dx2


def unique_first_appearance(idx):
    """tf.unique(x)[0]: distinct values in order of first occurrence."""
    idx = np.asarray(idx)
    _, first = np.unique(idx, return_index=True)
    return idx[np.sort(first)]


class SimGCL:
    """model/ranking/SimGCL.py restated.  ``noise`` for a step is a list of 2*L arrays [N, d]
    of U[0,1) draws (views 1 and 2, layer by layer); TF draws them inside the graph, tests
    inject them so that both sides see the same numbers."""

    def __init__(self, U0, V0, adj: sp.csr_matrix, n_layers, lr, reg, cl_rate, eps):
        self.nu, self.ni = U0.shape[0], V0.shape[0]
        self.E = np.concatenate([U0, V0]).astype(np.float32)
        self.adj = adj.astype(np.float32).tocsr()
        self.L, self.reg, self.cl_rate, self.eps = n_layers, reg, np.float32(cl_rate), np.float32(eps)
        self.opt = AdamTF114(self.E.shape, lr)

    def encoder(self, noises=None, signs=None):
        """LightGCN_encoder / perturbed_LightGCN_encoder (SimGCL.py:22-38): mean over the L
        propagated layers (ego layer excluded); with noise, emb += sign(emb)*normalize(noise)*eps
        after every layer and the perturbed emb feeds the next one.  ``signs`` (L arrays [N, d] of -1 / 0 / +1): use these
        in place of sign(emb) -- the pattern a recorded run of the reference used (sign() is discontinuous at 0)."""
        emb = self.E
        acc = np.zeros_like(emb)
        for k in range(self.L):
            emb = self.adj.dot(emb).astype(np.float32)
            if noises is not None:
                nz, _ = l2_normalize_rows(noises[k].astype(np.float32))
                sg = np.sign(emb) if signs is None else signs[k].astype(np.float32)
                emb = (emb + (sg * nz) * self.eps).astype(np.float32)
            acc += emb
        return (acc / np.float32(self.L)).astype(np.float32)

    def loss_and_grad(self, u_idx, i_idx, j_idx, noises, signs=None):
        nu = self.nu
        main = self.encoder()
        p1 = self.encoder(noises[:self.L], None if signs is None else signs[:self.L])
        p2 = self.encoder(noises[self.L:], None if signs is None else signs[self.L:])
        ui, ii, ji = np.asarray(u_idx), np.asarray(i_idx) + nu, np.asarray(j_idx) + nu
        rec, du, di, dj = bpr_batch_loss_and_grads(main[ui], main[ii], main[ji], self.reg)
        d_out = np.zeros_like(main)                     # gradients w.r.t. the three encoder outputs, summed:
        np.add.at(d_out, ui, du); np.add.at(d_out, ii, di); np.add.at(d_out, ji, dj)
        uu = unique_first_appearance(u_idx); vv = unique_first_appearance(i_idx) + nu
        cl = 0.0
        for rows in (uu, vv):
            l, d1, d2 = info_nce_loss_and_grads(p1[rows], p2[rows])
            cl += l
            d_out[rows] += self.cl_rate * d1 + self.cl_rate * d2   # both views back-propagate through the same linear map
        cl *= float(self.cl_rate)
        # encoder backward (identical for clean and perturbed: sign() has zero derivative):
        # dE0 = (1/L) sum_{k=1..L} A^k d_out
        c = d_out / np.float32(self.L)
        W = c.copy()
        for _ in range(self.L - 1):
            W = (c + self.adj.T.dot(W)).astype(np.float32)
        G = self.adj.T.dot(W).astype(np.float32)
        return rec + cl, rec, cl, G

    def train_step(self, u_idx, i_idx, j_idx, noises, signs=None):
        loss, rec, cl, g = self.loss_and_grad(u_idx, i_idx, j_idx, noises, signs)
        self.opt.step(self.E, g)
        return loss, rec, cl

    def final_embeddings(self):
        m = self.encoder()
        return m[:self.nu], m[self.nu:]


# ======================================================================================
# BPR, TensorFlow variant  (model/ranking/BPR.py:77-96)
# ======================================================================================
class BprTF:
    """loss = -sum log(sigmoid(y) + 1e-6) + reg*(l2_loss(U) + l2_loss(V)) on the FULL tables;
    gradients = batch scatter + reg*theta; dense Adam on U and V."""

    def __init__(self, U0, V0, lr, reg):
        self.nu = U0.shape[0]
        self.E = np.concatenate([U0, V0]).astype(np.float32)
        self.reg = np.float32(reg)
        self.opt = AdamTF114(self.E.shape, lr)

    def train_step(self, u_idx, i_idx, j_idx):
        f = np.float32
        ui, ii, ji = np.asarray(u_idx), np.asarray(i_idx) + self.nu, np.asarray(j_idx) + self.nu
        loss, du, di, dj = bpr_batch_loss_and_grads(self.E[ui], self.E[ii], self.E[ji], 0.0, eps=f(1e-6))
        loss += float(self.reg) * 0.5 * float((self.E.astype(np.float64) ** 2).sum())
        g = np.zeros_like(self.E)
        np.add.at(g, ui, du); np.add.at(g, ii, di); np.add.at(g, ji, dj)
        g = (g + self.reg * self.E).astype(f)
        self.opt.step(self.E, g)
        return loss


# ======================================================================================
# NGCF  (model/ranking/NGCF.py:9-71)
# ======================================================================================
def leaky_relu(x, alpha=np.float32(0.2)):
    """tf.nn.leaky_relu (default alpha = 0.2) = max(alpha*x, x)."""
    return np.maximum(alpha * x, x).astype(np.float32)


class NGCF:
    """Two propagation layers (fixed, NGCF.py:19) with weights W1_k, W2_k (d x d, Xavier):
        side = A E_k ;  pre = (side + E_k) W1_k + (E_k * side) W2_k ;  E_{k+1} = dropout(leaky_relu(pre), keep .9)
        out  = concat[E_0, l2_normalize(E_1), l2_normalize(E_2)]                 (N x 3d)
    batch BPR loss + batch L2 on the 3d-wide rows, Adam on U, V and the four weight matrices.
    ``masks`` for a step: two arrays [N, d] of 0/1 (the dropout keep decisions) or None for the
    inference graph (isTraining = 0)."""

    KEEP = np.float32(0.9)

    def __init__(self, U0, V0, W, adj: sp.csr_matrix, lr, reg):
        self.nu, self.ni, self.d = U0.shape[0], V0.shape[0], U0.shape[1]
        self.E = np.concatenate([U0, V0]).astype(np.float32)
        self.W = [[w.astype(np.float32).copy() for w in pair] for pair in W]      # [[W1_0, W2_0], [W1_1, W2_1]]
        self.adj = adj.astype(np.float32).tocsr()
        self.reg = reg
        self.optE = AdamTF114(self.E.shape, lr)
        self.optW = [[AdamTF114(w.shape, lr) for w in pair] for pair in self.W]

    def forward(self, masks=None, keep_cache=False):
        f = np.float32
        E = self.E
        outs = [E]
        cache = []
        for k in range(2):
            side = self.adj.dot(E).astype(f)
            A1 = (side + E).astype(f); A2 = (E * side).astype(f)
            pre = (A1 @ self.W[k][0] + A2 @ self.W[k][1]).astype(f)
            act = leaky_relu(pre)
            if masks is not None:
                fac = (masks[k].astype(f) / self.KEEP).astype(f)
                nxt = (act * fac).astype(f)
            else:
                fac = np.ones_like(act); nxt = act
            z, inv = l2_normalize_rows(nxt)
            outs.append(z)
            if keep_cache:
                cache.append(dict(E=E, side=side, A1=A1, A2=A2, pre=pre, fac=fac, nxt=nxt, z=z, inv=inv))
            E = nxt
        return np.concatenate(outs, axis=1).astype(f), cache

    def loss_and_grads(self, u_idx, i_idx, j_idx, masks):
        f = np.float32
        d, nu = self.d, self.nu
        allE, cache = self.forward(masks, keep_cache=True)
        ui, ii, ji = np.asarray(u_idx), np.asarray(i_idx) + nu, np.asarray(j_idx) + nu
        loss, du, di, dj = bpr_batch_loss_and_grads(allE[ui], allE[ii], allE[ji], self.reg)
        dAll = np.zeros_like(allE)
        np.add.at(dAll, ui, du); np.add.at(dAll, ii, di); np.add.at(dAll, ji, dj)
        dE_next = np.zeros((allE.shape[0], d), f)          # gradient w.r.t. E_{k+1} from later layers
        gW = [[None, None], [None, None]]
        for k in (1, 0):
            c = cache[k]
            dz = dAll[:, (k + 1) * d:(k + 2) * d]
            dnxt = dE_next + ((dz - c["z"] * (c["z"] * dz).sum(1, dtype=f)[:, None]) * c["inv"][:, None]).astype(f)
            dpre = (dnxt * c["fac"] * np.where(c["pre"] > 0, f(1), f(0.2))).astype(f)
            gW[k][0] = (c["A1"].T @ dpre).astype(f); gW[k][1] = (c["A2"].T @ dpre).astype(f)
            dA1 = (dpre @ self.W[k][0].T).astype(f); dA2 = (dpre @ self.W[k][1].T).astype(f)
            dside = (dA1 + dA2 * c["E"]).astype(f)
            dE_next = (dA1 + dA2 * c["side"] + self.adj.T.dot(dside)).astype(f)
        gE = (dE_next + dAll[:, :d]).astype(f)
        return loss, gE, gW

    def train_step(self, u_idx, i_idx, j_idx, masks):
        loss, gE, gW = self.loss_and_grads(u_idx, i_idx, j_idx, masks)
        self.optE.step(self.E, gE)
        for k in range(2):
            for t in range(2):
                self.optW[k][t].step(self.W[k][t], gW[k][t])
        return loss

    def inference_embeddings(self):
        """the 3d-wide tables the reference scores with at test time (isTraining = 0, NGCF.py:65-69)"""
        allE, _ = self.forward(None)
        return allE[:self.nu], allE[self.nu:]


# ======================================================================================
# SGL  (model/ranking/SGL.py:10-293)
# ======================================================================================
class SGL:
    """Three LightGCN views (mean over [E0..EL], ego layer included): the recommendation view on the full
    adjacency and two views on per-epoch augmented sub-graphs (``mats1[k]``, ``mats2[k]`` = the matrix of
    layer k; the same matrix for every k with node/edge dropout, a fresh one per layer with random walk).
    BPR loss on the main view, InfoNCE (calc_ssl_loss_v3: users and items of the batch merged into ONE
    contrast set, temperature -temp) between the two sub-graph views, weighted by -lambda; Adam."""

    def __init__(self, U0, V0, adj: sp.csr_matrix, n_layers, lr, reg, ssl_reg, temp):
        self.nu, self.ni = U0.shape[0], V0.shape[0]
        self.E = np.concatenate([U0, V0]).astype(np.float32)
        self.adj = adj.astype(np.float32).tocsr()
        self.L, self.reg, self.ssl_reg, self.temp = n_layers, reg, np.float32(ssl_reg), np.float32(temp)
        self.opt = AdamTF114(self.E.shape, lr)

    def view(self, mats):
        layers = [self.E]
        for k in range(self.L):
            layers.append(mats[k].dot(layers[-1]).astype(np.float32))
        acc = layers[0].copy()
        for x in layers[1:]:
            acc += x
        return (acc / np.float32(self.L + 1)).astype(np.float32)

    def _view_backward(self, mats, d_out):
        c = d_out / np.float32(self.L + 1)
        G = c.copy()
        for k in range(self.L - 1, -1, -1):          # E_{k+1} = M_k E_k  =>  dE_k = c + M_k^T dE_{k+1}
            G = (c + mats[k].T.dot(G)).astype(np.float32)
        return G

    def loss_and_grad(self, u_idx, i_idx, j_idx, mats1, mats2):
        nu = self.nu
        main_mats = [self.adj] * self.L
        main, s1, s2 = self.view(main_mats), self.view(mats1), self.view(mats2)
        ui, ii, ji = np.asarray(u_idx), np.asarray(i_idx) + nu, np.asarray(j_idx) + nu
        rec, du, di, dj = bpr_batch_loss_and_grads(main[ui], main[ii], main[ji], self.reg)
        d_main = np.zeros_like(main)
        np.add.at(d_main, ui, du); np.add.at(d_main, ii, di); np.add.at(d_main, ji, dj)
        rows = np.concatenate([unique_first_appearance(u_idx), unique_first_appearance(i_idx) + nu])
        l, d1, d2 = info_nce_loss_and_grads(s1[rows], s2[rows], tau=self.temp)
        d_s1 = np.zeros_like(main); d_s2 = np.zeros_like(main)
        d_s1[rows] = self.ssl_reg * d1; d_s2[rows] = self.ssl_reg * d2
        g = self._view_backward(main_mats, d_main) + self._view_backward(mats1, d_s1) + self._view_backward(mats2, d_s2)
        ssl = float(self.ssl_reg) * l
        return rec + ssl, rec, ssl, g.astype(np.float32)

    def train_step(self, u_idx, i_idx, j_idx, mats1, mats2):
        loss, rec, ssl, g = self.loss_and_grad(u_idx, i_idx, j_idx, mats1, mats2)
        self.opt.step(self.E, g)
        return loss, rec, ssl

    def final_embeddings(self):
        m = self.view([self.adj] * self.L)
        return m[:self.nu], m[self.nu:]


class BUIR:
    """model/ranking/BUIR.py:13-172 restated (pinned to a run of the reference's class through tests/golden/tf1shim.py,
    tests/test_oracle_tf_golden.py; gradients also cross-checked against torch autograd in tests/test_oracle_tfmodels.py).

    Online encoder: LightGCN over this epoch's sub-graph ``mat_o`` (mean over [E0..EL], BUIR.py:90-103), then
    q = tanh(online W + b) (:105-107).  Target encoder: the same propagation of the TARGET tables over ``mat_t``,
    no gradient (:98-100,109-113).  Loss (:127-130): sum_b [ (1 - cos(q[u_b], tar[i_b])) + (1 - cos(q[i_b], tar[u_b])) ] / 2
    with tf.math.l2_normalize (x / sqrt(max(sum x^2, 1e-12))).  Adam on the online tables, W and b (the target
    tables receive no gradient); after every step target = target*tau + online*(1 - tau) (:120-123,159)."""

    def __init__(self, U0, V0, W0, b0, n_layers, lr, tau):
        f = np.float32
        self.nu, self.ni = U0.shape[0], V0.shape[0]
        self.E = np.concatenate([U0, V0]).astype(f)          # online tables
        self.T = self.E.copy()                                # target tables (initialized_value of the online ones)
        self.W, self.b = W0.astype(f).copy(), b0.astype(f).reshape(1, -1).copy()
        self.L, self.tau = n_layers, f(tau)
        self.optE, self.optW, self.optb = AdamTF114(self.E.shape, lr), AdamTF114(self.W.shape, lr), AdamTF114(self.b.shape, lr)

    def _mean_prop(self, mat, X):
        acc, x = X.copy(), X
        for _ in range(self.L):
            x = mat.dot(x).astype(np.float32)
            acc += x
        return (acc / np.float32(self.L + 1)).astype(np.float32)

    def _mean_prop_backward(self, mat, d_out):
        c = (d_out / np.float32(self.L + 1)).astype(np.float32)
        G = c.copy()
        for _ in range(self.L):
            G = (c + mat.T.dot(G)).astype(np.float32)
        return G

    @staticmethod
    def _cos_loss_grad(q, t):
        """sum_b (1 - qhat.that)/2 and its gradient w.r.t. q (t is constant)"""
        f = np.float32
        nq = np.sqrt(np.maximum((q * q).sum(1, dtype=f), f(1e-12)), dtype=f)[:, None]
        nt = np.sqrt(np.maximum((t * t).sum(1, dtype=f), f(1e-12)), dtype=f)[:, None]
        qh, th = q / nq, t / nt
        c = (qh * th).sum(1, dtype=f)[:, None]
        loss = float(((f(1) - c[:, 0]) * f(0.5)).sum(dtype=np.float64))
        dq = -(th - c * qh) / nq * f(0.5)
        return loss, dq.astype(f)

    def loss_and_grads(self, u_idx, i_idx, mat_o, mat_t):
        f = np.float32
        nu = self.nu
        ui, ii = np.asarray(u_idx), np.asarray(i_idx) + nu
        online = self._mean_prop(mat_o, self.E)
        target = self._mean_prop(mat_t, self.T)
        xu, xi = online[ui], online[ii]
        qu = np.tanh(xu @ self.W + self.b, dtype=f); qi = np.tanh(xi @ self.W + self.b, dtype=f)
        l1, dqu = self._cos_loss_grad(qu, target[ii])
        l2, dqi = self._cos_loss_grad(qi, target[ui])
        dpu = dqu * (f(1) - qu * qu); dpi = dqi * (f(1) - qi * qi)        # through tanh
        gW = (xu.T @ dpu + xi.T @ dpi).astype(f)
        gb = (dpu.sum(0, dtype=f) + dpi.sum(0, dtype=f)).reshape(1, -1).astype(f)
        d_online = np.zeros_like(online)
        np.add.at(d_online, ui, (dpu @ self.W.T).astype(f)); np.add.at(d_online, ii, (dpi @ self.W.T).astype(f))
        gE = self._mean_prop_backward(mat_o, d_online)
        return l1 + l2, gE, gW, gb

    def train_step(self, u_idx, i_idx, mat_o, mat_t):
        loss, gE, gW, gb = self.loss_and_grads(u_idx, i_idx, mat_o, mat_t)
        self.optE.step(self.E, gE); self.optW.step(self.W, gW); self.optb.step(self.b, gb)
        self.T = (self.T * self.tau + self.E * (np.float32(1) - self.tau)).astype(np.float32)     # BUIR.py:120-123,159
        return loss

    def final_tables(self, adj):
        """(q_user, q_item, o_user, o_item) on the full adjacency (BUIR.py:160-167); score(u, .) =
        q_item . o_user[u] + o_item . q_user[u] (:172)"""
        online = self._mean_prop(adj, self.E)
        q = np.tanh(online @ self.W + self.b, dtype=np.float32)
        return q[:self.nu], q[self.nu:], online[:self.nu], online[self.nu:]


# ======================================================================================
# SEPT  (model/ranking/SEPT.py:19-323) -- social data, four LightGCN-style views with per-layer
# l2-normalisation, tri-training pseudo-labels (top-k of averaged softmax rows), neighbour-
# discrimination contrastive loss with several positives, two Adam optimizers.
# The graph builders are pure scipy in the reference and ARE pinned to it
# (tests/golden/sept_graphs_filmtrust.npz); the TF arithmetic is restated and held to a run of the reference's class (see header).
# ======================================================================================
def sept_row_normalised(M: sp.spmatrix) -> sp.csr_matrix:
    """``normalization`` of SEPT.py:53-59 / the tail of get_adj_mat (:107-113): D^-1/2 M D^-1/2 with D = ROW sums
    (M need not be symmetric), inf -> 0, in M's dtype."""
    rowsum = np.array(M.sum(1))
    with np.errstate(divide="ignore"):
        d_inv = np.power(rowsum, -0.5).flatten()
    d_inv[np.isinf(d_inv)] = 0.0
    D = sp.diags(d_inv)
    return D.dot(M).dot(D).tocsr()


def sept_social_views(n_users: int, n_items: int, uid, iid, follower, followee):
    """get_birectional_social_matrix + buildSparseRatingMatrix + get_social_related_views (SEPT.py:32-67):
    B = A o A (A = follower->followee counts, so B is A for 0/1 data -- the name notwithstanding, nothing is made
    bidirectional); friend view = (B B) o B + I, sharing view = (R R^T) o B + I, both row-normalised.
    Returns (social, sharing) CSR float64 (the reference converts to float32 when it builds the tensors, :117)."""
    follower, followee = np.asarray(follower), np.asarray(followee)
    tmp = sp.csr_matrix((np.ones_like(follower, dtype=np.float32), (follower, followee)), shape=(n_users, n_users))
    B = tmp.multiply(tmp)
    R = sp.coo_matrix((np.ones(len(uid), np.float32), (np.asarray(uid), np.asarray(iid))), shape=(n_users, n_items), dtype=np.float32)
    social = B.dot(B).multiply(B) + sp.eye(n_users)
    sharing = R.dot(R.T).multiply(B) + sp.eye(n_users)
    return sept_row_normalised(social), sept_row_normalised(sharing)


def sept_sub_adjacency(n_users: int, n_items: int, uid, iid, follower, followee, keep_idx=None, skeep_idx=None):
    """get_adj_mat (SEPT.py:79-114).  With keep lists (is_subgraph, drop_rate > 0): kept rating edges both ways plus
    the kept follow edges, squared, in the user-user block, row-normalised; without: the plain joint adjacency."""
    n = n_users + n_items
    uid, iid = np.asarray(uid), np.asarray(iid)
    if keep_idx is not None:
        u, i = uid[keep_idx], iid[keep_idx]
        tmp = sp.csr_matrix((np.ones_like(u, dtype=np.float32), (u, n_users + i)), shape=(n, n))
        adj = tmp + tmp.T
        fo, fe = np.asarray(follower)[skeep_idx], np.asarray(followee)[skeep_idx]
        soc = sp.csr_matrix((np.ones_like(fo, dtype=np.float32), (fo, fe)), shape=(n, n))
        adj = adj + soc.multiply(soc)
    else:
        tmp = sp.csr_matrix((np.ones_like(uid, dtype=np.float32), (uid, iid + n_users)), shape=(n, n))
        adj = tmp + tmp.T
    return sept_row_normalised(adj)


def l2_normalize_bwd(x, inv, d):
    """gradient of z = x * rsqrt(max(sum x^2, 1e-12)) given dz: (dz - z (z.dz)) * inv (rows at the clamp are all-zero
    rows here, where this reduces to dz * inv as TF's does)"""
    f = np.float32
    z = (x * inv[:, None]).astype(f)
    return ((d - z * (z * d).sum(1, dtype=f)[:, None]) * inv[:, None]).astype(f)


def top_k_rows(score, k):
    """tf.math.top_k(score, k)[1]: the k largest per row, equal values in index order"""
    return np.argsort(-score, axis=1, kind="stable")[:, :k]


def neighbour_discrimination(z, a, pos, tau=np.float32(0.1)):
    """SEPT.neighbor_discrimination (SEPT.py:233-248) on normalised rows z (one encoder) and a (the augmented view),
    pos[i] = the row's pseudo-labelled positives:  -sum_i log( sum_{k in pos[i]} e^{z_i.a_k/tau} / sum_j e^{z_i.a_j/tau} ).
    Returns (loss, dz, da), hand-derived."""
    f = np.float32
    E = np.exp((z @ a.T).astype(f) / f(tau), dtype=f)
    ttl = E.sum(1, dtype=f)
    Ep = np.take_along_axis(E, pos, 1)
    ps = Ep.sum(1, dtype=f)
    loss = float(-np.log(ps / ttl, dtype=f).sum(dtype=np.float64))
    G = (E / ttl[:, None]).astype(f)
    np.put_along_axis(G, pos, np.take_along_axis(G, pos, 1) - Ep / ps[:, None], 1)
    return loss, (G @ a).astype(f) / f(tau), (G.T @ z).astype(f) / f(tau)


def sept_ssl_loss_and_grads(xf, xh, xe, xg, k, labels=None):
    """label_prediction + generate_pesudo_labels + neighbor_discrimination (SEPT.py:214-262) on the gathered rows of the
    friend / sharing / preference view sums and of the perturbed-graph view (xg).  Returns (loss, labels,
    [dxf, dxh, dxe, dxg]) -- unscaled gradients w.r.t. the gathered rows; labels = [f_pos, sh_pos, r_pos]."""
    f = np.float32
    a, ra = l2_normalize_rows(xg)
    zs, rs = zip(*(l2_normalize_rows(x) for x in (xf, xh, xe)))
    if labels is None:
        def prob(z):                                                               # tf.nn.softmax(emb aug^T), :219-223
            x = (z @ a.T).astype(f)
            e = np.exp(x - x.max(1, keepdims=True), dtype=f)
            return (e / e.sum(1, dtype=f)[:, None]).astype(f)
        p_soc, p_sh, p_rec = (prob(z) for z in zs)
        labels = [top_k_rows((p_sh + p_rec) / f(2), k), top_k_rows((p_soc + p_rec) / f(2), k),
                  top_k_rows((p_soc + p_sh) / f(2), k)]                             # f_pos, sh_pos, r_pos (:258-260)
    nd, da, dx = 0.0, np.zeros_like(a), []
    for z, r, pos in zip(zs, rs, labels):
        l, dz, dai = neighbour_discrimination(z, a, np.asarray(pos))
        nd += l; da += dai
        dx.append(((dz - z * (z * dz).sum(1, dtype=f)[:, None]) * r[:, None]).astype(f))
    dx.append(((da - a * (a * da).sum(1, dtype=f)[:, None]) * ra[:, None]).astype(f))
    return nd, labels, dx


class SEPT:
    """model/ranking/SEPT.py restated (LightGCN-structured views, the NGCF-structured block is commented out in the
    reference).  Variables Uv, Vv; every view starts from Uv/2, Vv/2 (:129-130)."""

    def __init__(self, U0, V0, adj, social, sharing, n_layers, lr, reg, ss_rate, ins_cnt):
        f = np.float32
        self.nu, self.ni = U0.shape[0], V0.shape[0]
        self.W = np.concatenate([U0, V0]).astype(f)                    # the two tf.Variables, stacked
        self.adj, self.social, self.sharing = (m.astype(f).tocsr() for m in (adj, social, sharing))
        self.L, self.reg, self.ss_rate, self.k = n_layers, float(reg), f(ss_rate), int(ins_cnt)
        self.opt1, self.opt2 = AdamTF114(self.W.shape, lr), AdamTF114(self.W.shape, lr)   # v1_opt / v2_opt (:267-270)

    # ---- one view: x_0 = X0; x_k = M x_{k-1}; S = x_0 + sum_k l2_normalize(x_k)   (SEPT.py:142-160, 205-211)
    def chain(self, M, X0):
        xs, invs, S = [X0], [], X0.copy()
        for _ in range(self.L):
            x = M.dot(xs[-1]).astype(np.float32)
            z, inv = l2_normalize_rows(x)
            S += z; xs.append(x); invs.append(inv)
        return S, xs, invs

    def chain_bwd(self, M, xs, invs, dS):
        g = None
        for k in range(self.L, 0, -1):
            nb = l2_normalize_bwd(xs[k], invs[k - 1], dS)
            g = nb if g is None else (nb + M.T.dot(g)).astype(np.float32)
        return (dS + M.T.dot(g)).astype(np.float32)

    def rec_embeddings(self):
        """(rec_user_embeddings, rec_item_embeddings): what the reference ranks with (SEPT.py:207-208, 307)"""
        S, _, _ = self.chain(self.adj, (self.W / np.float32(2)).astype(np.float32))
        return S[:self.nu], S[self.nu:]

    def loss_and_grad(self, u_idx, i_idx, j_idx, sub_adj=None, labels=None):
        """sub_adj None: rec_loss only (epochs <= maxEpoch/3); else rec_loss + ss_rate * neighbor_dis_loss.
        ``labels``: use these pseudo labels instead of computing them (they carry no gradient; tests pass the other
        implementation's so that a near-tie in the top-k cannot fork the two trajectories).
        Returns (rec_loss, neighbor_dis_loss, dW, pseudo-label index lists)."""
        f = np.float32
        nu = self.nu
        E0 = (self.W / f(2)).astype(f)
        Se, xe, ie = self.chain(self.adj, E0)
        ui, ii, ji = np.asarray(u_idx), np.asarray(i_idx) + nu, np.asarray(j_idx) + nu
        rec, du, di, dj = bpr_batch_loss_and_grads(Se[ui], Se[ii], Se[ji], 0.0)
        rec += self.reg * 0.5 * float((E0.astype(np.float64) ** 2).sum())            # regU (l2(U/2) + l2(V/2)), :252
        dSe = np.zeros_like(Se)
        np.add.at(dSe, ui, du); np.add.at(dSe, ii, di); np.add.at(dSe, ji, dj)
        dE0 = f(self.reg) * E0
        nd = 0.0
        if sub_adj is not None:
            sub = sub_adj.astype(f).tocsr()
            Sf, xf, if_ = self.chain(self.social, E0[:nu]); Sh, xh, ih = self.chain(self.sharing, E0[:nu])
            Sg, xg, ig = self.chain(sub, E0)
            rows = unique_first_appearance(u_idx)
            nd, labels, (dxf, dxh, dxe, dxg) = sept_ssl_loss_and_grads(Sf[rows], Sh[rows], Se[rows], Sg[rows], self.k, labels)
            dSf, dSh, dSg = np.zeros_like(Sf), np.zeros_like(Sh), np.zeros_like(Sg)
            w = self.ss_rate
            dSf[rows] = w * dxf; dSh[rows] = w * dxh; dSe[rows] += w * dxe; dSg[rows] = w * dxg
            dE0[:nu] += self.chain_bwd(self.social, xf, if_, dSf) + self.chain_bwd(self.sharing, xh, ih, dSh)
            dE0 += self.chain_bwd(sub, xg, ig, dSg)
        dE0 += self.chain_bwd(self.adj, xe, ie, dSe)
        return rec, nd, (dE0 / f(2)).astype(f), labels

    def train_step(self, u_idx, i_idx, j_idx, sub_adj=None, labels=None):
        """returns (rec_loss, ss_rate * neighbor_dis_loss) as the reference prints them (SEPT.py:292, 301)"""
        rec, nd, g, _ = self.loss_and_grad(u_idx, i_idx, j_idx, sub_adj, labels)
        (self.opt1 if sub_adj is None else self.opt2).step(self.W, g)
        return rec, float(self.ss_rate) * nd


# ======================================================================================
# MHCN  (model/ranking/MHCN.py:15-240) -- multi-channel hypergraph convolution over three motif-induced user-user
# adjacencies plus the user-item graph, self-gating, channel attention, hierarchical mutual-information maximisation.
# The graph builders are pure scipy / python in the reference and ARE pinned to it (tests/golden/mhcn_graphs_filmtrust.npz);
# the TF arithmetic is restated and held to a run of the reference's class (see header).  tf.random.shuffle is not reproducible outside TF: the five
# permutations each hierarchical_self_supervision call draws per step are inputs here.
# ======================================================================================
def mhcn_motif_adjacencies(n_users: int, n_items: int, uid, iid, follower, followee):
    """buildSparseRelationMatrix / buildSparseRatingMatrix / buildMotifInducedAdjacencyMatrix (MHCN.py:26-85):
    the ten triangle motifs over the follow graph S and the purchase graph Y, summed into the social (H_s), joint (H_j)
    and purchase (H_p, co-purchase counts > 1) channels, each divided by its row sums.  float32 CSR."""
    S = sp.coo_matrix((np.ones(len(follower), np.float32), (np.asarray(follower), np.asarray(followee))), shape=(n_users, n_users), dtype=np.float32)
    Y = sp.coo_matrix((np.ones(len(uid), np.float32), (np.asarray(uid), np.asarray(iid))), shape=(n_users, n_items), dtype=np.float32)
    B = S.multiply(S.T)
    U = S - B
    C1 = (U.dot(U)).multiply(U.T); A1 = C1 + C1.T
    C2 = (B.dot(U)).multiply(U.T) + (U.dot(B)).multiply(U.T) + (U.dot(U)).multiply(B); A2 = C2 + C2.T
    C3 = (B.dot(B)).multiply(U) + (B.dot(U)).multiply(B) + (U.dot(B)).multiply(B); A3 = C3 + C3.T
    A4 = (B.dot(B)).multiply(B)
    C5 = (U.dot(U)).multiply(U) + (U.dot(U.T)).multiply(U) + (U.T.dot(U)).multiply(U); A5 = C5 + C5.T
    A6 = (U.dot(B)).multiply(U) + (B.dot(U.T)).multiply(U.T) + (U.T.dot(U)).multiply(B)
    A7 = (U.T.dot(B)).multiply(U.T) + (B.dot(U)).multiply(U) + (U.dot(U.T)).multiply(B)
    A8 = (Y.dot(Y.T)).multiply(B)
    A9 = (Y.dot(Y.T)).multiply(U); A9 = A9 + A9.T
    A10 = Y.dot(Y.T) - A8 - A9
    with np.errstate(divide="ignore"):
        H_s = sum([A1, A2, A3, A4, A5, A6, A7]); H_s = H_s.multiply(1.0 / H_s.sum(axis=1).reshape(-1, 1))
        H_j = sum([A8, A9]); H_j = H_j.multiply(1.0 / H_j.sum(axis=1).reshape(-1, 1))
        H_p = A10.multiply(A10 > 1); H_p = H_p.multiply(1.0 / H_p.sum(axis=1).reshape(-1, 1))
    return [sp.csr_matrix(H) for H in (H_s, H_j, H_p)]


def mhcn_joint_adjacency(n_users: int, n_items: int, uid, iid, ratings):
    """buildJointAdjacency (MHCN.py:46-52): one entry per training row, rating / sqrt(#items of u) / sqrt(#users of i)
    (counts of DISTINCT partners, python floats -> float32 tensor); duplicate rows add up in the sparse product."""
    uid, iid = np.asarray(uid), np.asarray(iid)
    pairs = np.unique(np.stack([uid, iid], 1), axis=0)
    du = np.bincount(pairs[:, 0], minlength=n_users); di = np.bincount(pairs[:, 1], minlength=n_items)
    from math import sqrt
    vals = [float(r) / sqrt(du[u]) / sqrt(di[i]) for u, i, r in zip(uid.tolist(), iid.tolist(), np.asarray(ratings).tolist())]
    return sp.csr_matrix((np.asarray(vals, np.float32), (uid, iid)), shape=(n_users, n_items))


def _sigmoid(x):
    return (np.float32(1) / (np.float32(1) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def _nls(x):
    """-log(sigmoid(x)) as tf evaluates it: -log(1 / (1 + exp(-x)))"""
    return -np.log(_sigmoid(x), dtype=np.float32)


class MHCN:
    """model/ranking/MHCN.py:93-229 restated.  ``weights``: dict with the reference's keys (gating1-4, gating_bias1-4,
    sgating1-4, sgating_bias1-4, attention, attention_mat).  ``perms`` for a step: for each of the three channels
    (row_pi1, col_k2, row_pi2, col_k3, row_pi3) -- row_shuffle, row_column_shuffle (neg2), row_column_shuffle (global)."""

    def __init__(self, U0, V0, weights, H, R, n_layers, lr, reg, ss_rate):
        f = np.float32
        self.nu, self.ni, self.d = U0.shape[0], V0.shape[0], U0.shape[1]
        self.U, self.V = U0.astype(f).copy(), V0.astype(f).copy()
        self.w = {k: np.asarray(v, f).copy() for k, v in weights.items()}
        self.H = [h.astype(f).tocsr() for h in H]; self.R = R.astype(f).tocsr()
        self.L, self.reg, self.ss_rate = n_layers, f(reg), f(ss_rate)
        self.opt = {k: AdamTF114(v.shape, lr) for k, v in list(self.w.items()) + [("U", self.U), ("V", self.V)]}

    # ---- pieces with their backward closures ----------------------------------------------------------------
    def gate(self, X, W, b):
        s = _sigmoid((X @ W + b).astype(np.float32))
        def bwd(dY):
            q = (dY * X * s * (np.float32(1) - s)).astype(np.float32)
            return (dY * s + q @ W.T).astype(np.float32), (X.T @ q).astype(np.float32), q.sum(0, keepdims=True).astype(np.float32)
        return (X * s).astype(np.float32), bwd

    def attention(self, es):
        f = np.float32
        a, M = self.w["attention"], self.w["attention_mat"]
        v = (M @ a[0]).astype(f)                                        # sum(a * (e M), 1) = e . (M a)
        w = np.stack([(e @ M * a).sum(1, dtype=f) for e in es], 1)      # [n, 3], as the reference evaluates it
        ex = np.exp(w - w.max(1, keepdims=True), dtype=f)
        sc = (ex / ex.sum(1, keepdims=True, dtype=f)).astype(f)
        out = sum(sc[:, k:k + 1] * es[k] for k in range(3)).astype(f)
        def bwd(dOut):
            dsc = np.stack([(dOut * e).sum(1, dtype=f) for e in es], 1)
            dw = (sc * (dsc - (sc * dsc).sum(1, keepdims=True, dtype=f))).astype(f)
            des = [(sc[:, k:k + 1] * dOut + dw[:, k:k + 1] * v[None, :]).astype(f) for k in range(3)]
            dv = sum((dw[:, k:k + 1] * es[k]).sum(0, dtype=f) for k in range(3)).astype(f)
            return des, (M.T @ dv)[None, :].astype(f), np.outer(dv, a[0]).astype(f)      # d attention, d attention_mat
        return out, sc, bwd

    def hss(self, em, H, perm):
        """hierarchical_self_supervision (MHCN.py:184-206); returns (loss, d em)"""
        f = np.float32
        p1, k2, p2, k3, p3 = perm
        n = em.shape[0]
        edge = H.dot(em).astype(f)
        e2, e3 = edge[:, k2][p2], edge[:, k3][p3]
        pos, neg1, neg2 = (em * edge).sum(1, dtype=f), (em[p1] * edge).sum(1, dtype=f), (e2 * em).sum(1, dtype=f)
        graph = edge.mean(0, dtype=f)
        pg, ng = edge @ graph, e3 @ graph
        loss = float((_nls(pos - neg1) + _nls(neg1 - neg2)).sum(dtype=np.float64) + _nls(pg - ng).sum(dtype=np.float64))
        c1, c2, c3 = -(f(1) - _sigmoid(pos - neg1)), -(f(1) - _sigmoid(neg1 - neg2)), -(f(1) - _sigmoid(pg - ng))
        d_pos, d_neg1, d_neg2 = c1, c2 - c1, -c2
        dem = (d_pos[:, None] * edge + d_neg2[:, None] * e2).astype(f)
        np.add.at(dem, p1, d_neg1[:, None] * edge)
        dedge = (d_pos[:, None] * em + d_neg1[:, None] * em[p1] + c3[:, None] * graph[None, :]).astype(f)
        t = np.zeros_like(edge); np.add.at(t, p2, d_neg2[:, None] * em)                  # back through rows p2 ...
        inv2 = np.empty_like(k2); inv2[k2] = np.arange(k2.size); dedge += t[:, inv2]         # ... and columns k2
        t = np.zeros_like(edge); np.add.at(t, p3, (-c3)[:, None] * graph[None, :])
        inv3 = np.empty_like(k3); inv3[k3] = np.arange(k3.size); dedge += t[:, inv3]
        dgraph = (c3[:, None] * (edge - e3)).sum(0, dtype=f)
        dedge += dgraph[None, :] / f(n)
        return loss, (dem + H.T.dot(dedge)).astype(f)

    def forward(self):
        """-> final_user, final_item, cache"""
        f = np.float32
        w = self.w
        gates = [self.gate(self.U, w[f"gating{k}"], w[f"gating_bias{k}"]) for k in (1, 2, 3, 4)]
        c = [g[0] for g in gates[:3]]; s = gates[3][0]; t = self.V
        lay = dict(c=[c], s=[s], t=[t], att=[])
        sums = [x.copy() for x in c] + [s.copy(), t.copy()]
        invs = []
        for _ in range(self.L):
            mix, _, att_b = self.attention(c)
            mixed = (mix + s / f(2)).astype(f)
            c = [self.H[k].dot(c[k]).astype(f) for k in range(3)]
            t_new = self.R.T.dot(mixed).astype(f)
            s = self.R.dot(t).astype(f)
            t = t_new
            inv = []
            for idx, x in enumerate(c + [s, t]):
                z, r = l2_normalize_rows(x); sums[idx] += z; inv.append(r)
            invs.append(inv); lay["c"].append(c); lay["s"].append(s); lay["t"].append(t); lay["att"].append(att_b)
        fu, score, att_f = self.attention(sums[:3])
        fu = (fu + sums[3] / f(2)).astype(f)
        return fu, sums[4], dict(gates=gates, lay=lay, invs=invs, att_f=att_f, score=score)

    def loss_and_grads(self, u_idx, i_idx, j_idx, perms):
        """-> (rec_loss, ss_loss, reg_loss, grads dict keyed like self.opt)"""
        f = np.float32
        w = self.w
        fu, fi, cache = self.forward()
        ub, ib, jb = fu[u_idx], fi[i_idx], fi[j_idx]
        rec, du, di, dj = bpr_batch_loss_and_grads(ub, ib, jb, 0.0)
        dfu, dfi = np.zeros_like(fu), np.zeros_like(fi)
        np.add.at(dfu, u_idx, du); np.add.at(dfi, i_idx, di); np.add.at(dfi, j_idx, dj)
        g = {k: (f(0.001) * v).astype(f) for k, v in w.items()}                         # 0.001 * l2_loss of every weight (:211-212)
        reg = 0.001 * 0.5 * float(sum((v.astype(np.float64) ** 2).sum() for v in w.values()))
        reg += float(self.reg) * 0.5 * float((self.U.astype(np.float64) ** 2).sum() + (self.V.astype(np.float64) ** 2).sum())
        ss = 0.0
        for k in range(3):
            sg, sg_b = self.gate(fu, w[f"sgating{k + 1}"], w[f"sgating_bias{k + 1}"])
            l, dsg = self.hss(sg, self.H[k], perms[k])
            ss += l
            dx, dW, db = sg_b(self.ss_rate * dsg)
            dfu += dx; g[f"sgating{k + 1}"] += dW; g[f"sgating_bias{k + 1}"] += db
        # final aggregation
        des, da, dM = cache["att_f"](dfu)
        g["attention"] += da; g["attention_mat"] += dM
        dsum = des + [(dfu / f(2)).astype(f), dfi]                                     # d sums of c1, c2, c3, s, t
        lay, invs = cache["lay"], cache["invs"]
        L = self.L
        gc = [None] * 3; gs = gt = None
        for l in range(L, 0, -1):
            nb = [l2_normalize_bwd(x, r, d) for x, r, d in zip(lay["c"][l] + [lay["s"][l], lay["t"][l]], invs[l - 1], dsum)]
            gc = [nb[k] if gc[k] is None else (nb[k] + gc[k]).astype(f) for k in range(3)]
            gs = nb[3] if gs is None else (nb[3] + gs).astype(f)
            gt = nb[4] if gt is None else (nb[4] + gt).astype(f)
            dmixed = self.R.dot(gt).astype(f)                                            # t^(l) = R^T mixed
            gt_prev = self.R.T.dot(gs).astype(f)                                         # s^(l) = R t^(l-1)
            gc_prev = [self.H[k].T.dot(gc[k]).astype(f) for k in range(3)]               # c^(l) = H c^(l-1)
            des, da, dM = lay["att"][l - 1](dmixed)
            g["attention"] += da; g["attention_mat"] += dM
            gc = [(gc_prev[k] + des[k]).astype(f) for k in range(3)]; gs = (dmixed / f(2)).astype(f); gt = gt_prev
        dG = [dsum[k] if gc[k] is None else (dsum[k] + gc[k]).astype(f) for k in range(3)]
        dG.append(dsum[3] if gs is None else (dsum[3] + gs).astype(f))
        dV = dsum[4] if gt is None else (dsum[4] + gt).astype(f)
        dU = (self.reg * self.U).astype(f)
        for k in range(4):
            dx, dW, db = cache["gates"][k][1](dG[k])
            dU += dx; g[f"gating{k + 1}"] += dW; g[f"gating_bias{k + 1}"] += db
        g["U"] = dU; g["V"] = (dV + self.reg * self.V).astype(f)
        return rec, ss, reg, g

    def train_step(self, u_idx, i_idx, j_idx, perms):
        """returns rec_loss, what the reference prints (MHCN.py:223-225)"""
        rec, ss, reg, g = self.loss_and_grads(u_idx, i_idx, j_idx, perms)
        for k, opt in self.opt.items():
            opt.step(self.U if k == "U" else self.V if k == "V" else self.w[k], g[k])
        return rec
