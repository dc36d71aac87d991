"""numpy/scipy restatement of the reference's TensorFlow-1.14 graph models (LightGCN first).
TEST INFRASTRUCTURE ONLY (see oracle/qrec_oracle.c header).

PARITY STATUS: **partly unpinned.**  tensorflow==1.14.0 (README.md:57) cannot be installed
here (no cp310 wheel, no network), so the TF arithmetic below is a restatement of TF's
published op semantics, anchored on the reference's call sites; it is cross-checked against
torch-CPU autograd (tests/test_oracle_tfmodels.py), not against a TF run.  What IS pinned to
the live reference: the joint normalized adjacency (base/graphRecommender.py:10-29 is pure
scipy -> tests/golden/pairwise_adj_filmtrust.npz) and the batch sampler stream
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
