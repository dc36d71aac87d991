"""A small stand-in for the ``tensorflow`` 1.14 module, for ONE purpose: to let the golden-vector generator
(tests/golden/gen_golden_tf.py) execute the reference's TensorFlow model classes UNMODIFIED in this container, where
tensorflow==1.14.0 (reference README.md:57) cannot be installed (no cp310 wheel, no network).

TEST INFRASTRUCTURE ONLY.  Nothing in the product, in bench.py or in the GPU tests imports this file; it runs when the
fixtures are (re)generated, never on the GPU box.

What it is: the graph-construction API the reference's hot-path models call (``tf.placeholder``, ``tf.Variable``,
``tf.nn.embedding_lookup``, ``tf.sparse_tensor_dense_matmul``, ``tf.train.AdamOptimizer(...).minimize`` ... -- the list is the
set of names below, nothing is guessed beyond it) as a LAZY graph of ``Tensor`` nodes; ``Session.run`` evaluates the
fetched nodes with torch on the CPU (float32 like TF's kernels, reverse-mode autograd for ``minimize``).  So the model
code that builds the graph -- which ops, in which order, with which constants -- is the reference's own; what is restated
here is only the published semantics of each primitive op (TF 1.14 python/ops sources are quoted where a detail
matters: dropout's scale-then-mask, l2_normalize's clamp, unique's first-appearance order, ApplyAdam's functor).

Randomness.  TF's in-graph random ops cannot be reproduced; every random op here draws from
``numpy.random.default_rng([seed, run_index, op_index])`` where ``op_index`` numbers the random ops in the order the
reference's code CREATED them and ``run_index`` counts ``Session.run`` calls -- so a test can regenerate the very numbers
a step saw (``random_uniform(seed, run_index, op_index, shape)`` below) and feed them to the implementation under test.
Initial variable values are drawn at creation (``initial_draw``) and recorded by the generator.
"""
from __future__ import annotations

import builtins as _b
import os
import types

import numpy as np
import torch

__version__ = "1.14.0-shim"
DT = torch.float64 if os.environ.get("TF1SHIM_DTYPE") == "float64" else torch.float32
NPDT = np.float64 if DT == torch.float64 else np.float32


class DType:
    def __init__(self, name, is_int=False, is_bool=False):
        self.name, self.is_int, self.is_bool = name, is_int, is_bool

    def __repr__(self):
        return f"tf.{self.name}"


int32 = DType("int32", is_int=True)
int64 = DType("int64", is_int=True)
float32 = DType("float32")
bool = DType("bool", is_bool=True)          # noqa: A001  (the reference writes tf.bool)


class _State:
    def __init__(self):
        self.reset(0)

    def reset(self, seed):
        self.seed = int(seed)
        self.variables = []          # in creation order
        self.n_random_ops = 0
        self.run_index = 0
        self.run_log = []            # (run_index, [names of random ops evaluated]) per Session.run
        self.n_sign_ops = 0
        self.sign_log = {}           # run_index -> {sign op index (creation order): int8 array of the signs it produced}


STATE = _State()


def reset(seed=0):
    """forget every variable / random op of earlier graphs; seed the draws of the next one"""
    STATE.reset(seed)


def random_uniform(seed, run_index, op_index, shape):
    """the U[0,1) float32 numbers random op ``op_index`` produced in Session.run number ``run_index``"""
    return np.random.default_rng([int(seed), int(run_index), int(op_index)]).random(tuple(int(s) for s in shape), dtype=np.float32)


def initial_draw(seed, var_index, kind, shape, scale):
    """initial value of the ``var_index``-th created variable: kind 'truncated_normal' (scale = stddev; values beyond two
    standard deviations are redrawn, tf.truncated_normal) or 'xavier_uniform' (scale = limit)"""
    rng = np.random.default_rng([int(seed), 1 << 30, int(var_index)])
    shape = tuple(int(s) for s in shape)
    if kind == "truncated_normal":
        x = rng.standard_normal(shape)
        bad = np.abs(x) > 2.0
        while bad.any():
            x[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(x) > 2.0
        return (x * scale).astype(np.float32)
    if kind == "xavier_uniform":
        return rng.uniform(-scale, scale, shape).astype(np.float32)
    raise ValueError(kind)


# ---------------------------------------------------------------------------------------------------------------------
class Tensor:
    """a node of the lazy graph: ``fn(ctx, *input_values) -> value`` (torch tensor, or a python tuple of them)"""

    def __init__(self, fn, inputs=(), name=None, dtype=None):
        self.fn, self.inputs, self.name, self.dtype = fn, tuple(inputs), name, dtype

    def _eval(self, ctx):
        if self in ctx:
            return ctx[self]
        vals = [i._eval(ctx) for i in self.inputs]
        v = self.fn(ctx, *vals)
        ctx[self] = v
        return v

    @property
    def shape(self):
        """static shape, by evaluating the node without feeds (enough for what the reference asks: SimGCL.py:34
        ``tf.random.uniform(emb.shape)`` on a tensor that depends on variables and the adjacency only)"""
        with torch.no_grad():
            v = self._eval({"__run__": (1 << 31) - 1})      # a run index no Session.run ever has: shapes only
        return tuple(v.shape)

    def get_shape(self):
        return self.shape

    # python operators, as tf.Tensor overloads them
    def __add__(self, o): return _binary(torch.add, self, o)
    def __radd__(self, o): return _binary(torch.add, o, self)
    def __sub__(self, o): return _binary(torch.sub, self, o)
    def __rsub__(self, o): return _binary(torch.sub, o, self)
    def __mul__(self, o): return _binary(torch.mul, self, o)
    def __rmul__(self, o): return _binary(torch.mul, o, self)
    def __truediv__(self, o): return _binary(torch.div, self, o)
    def __rtruediv__(self, o): return _binary(torch.div, o, self)
    def __neg__(self): return Tensor(lambda ctx, a: -a, [self])
    def __getitem__(self, k): return Tensor(lambda ctx, a: a[k], [self], dtype=self.dtype)
    def __matmul__(self, o): return matmul(self, o)
    __hash__ = object.__hash__

    def __eq__(self, o):         # tf.Tensor.__eq__ is identity in 1.x
        return self is o


class SparseTensor:
    """indices / values / dense_shape given as arrays (a constant matrix) or as tensors, e.g. placeholders (SGL.py:36-55 feeds
    a fresh sub-graph every epoch): then the matrix is assembled when a run needs it"""

    def __init__(self, indices, values, dense_shape):
        self.dynamic = any(isinstance(a, Tensor) for a in (indices, values, dense_shape))
        if self.dynamic:
            self.parts = [_t(indices), _t(values), _t(dense_shape)]
            return
        idx = np.asarray(indices, dtype=np.int64)
        self.dense_shape = tuple(int(s) for s in dense_shape)
        self.indices, self.values = idx, np.asarray(values)
        self._t = self._assemble(torch.from_numpy(idx.copy()), torch.as_tensor(self.values, dtype=DT), self.dense_shape)

    @staticmethod
    def _assemble(idx, val, shape):
        return torch.sparse_coo_tensor(idx.to(torch.int64).T, val.to(DT), tuple(int(s) for s in shape)).coalesce()


def _const(x):
    a = np.asarray(x)
    if a.dtype.kind in "iu":
        t = torch.as_tensor(a.astype(np.int64))
    elif a.dtype.kind == "b":
        t = torch.as_tensor(a)
    else:
        t = torch.as_tensor(a.astype(NPDT))
    return Tensor(lambda ctx: t)


def _t(x):
    if isinstance(x, Tensor):
        return x
    if isinstance(x, (list, tuple)) and any(isinstance(e, Tensor) for e in x):
        return stack([_t(e) for e in x], axis=0)          # tf.convert_to_tensor packs a list of tensors
    return _const(x)


def _binary(op, a, b):
    return Tensor(lambda ctx, x, y: op(x, y), [_t(a), _t(b)])


def _unary(op):
    def f(x, name=None):
        return Tensor(lambda ctx, a: op(a), [_t(x)], name=name)
    return f


def _axes(axis):
    if axis is None:
        return None
    return tuple(axis) if isinstance(axis, (list, tuple)) else int(axis)


# ---- graph inputs and state -----------------------------------------------------------------------------------------
def placeholder(dtype, shape=None, name=None):
    def missing(ctx):
        raise RuntimeError(f"placeholder {name!r} was not fed")
    return Tensor(missing, name=name, dtype=dtype)


class _Init:
    """what a random initializer hands to tf.Variable: the draw is made when the variable is created"""

    def __init__(self, kind, shape, scale):
        self.kind, self.shape, self.scale = kind, tuple(int(s) for s in shape), float(scale)


def truncated_normal(shape, mean=0.0, stddev=1.0, dtype=float32, seed=None, name=None):
    assert mean == 0.0
    return _Init("truncated_normal", shape, stddev)


def _xavier_initializer(uniform=True, seed=None, dtype=float32):
    assert uniform

    def init(shape, dtype=None, partition_info=None):
        fan_in, fan_out = (shape[0], shape[1]) if len(shape) == 2 else (shape[-2], shape[-1])
        return _Init("xavier_uniform", shape, np.sqrt(6.0 / (fan_in + fan_out)))     # contrib/layers/python/layers/initializers.py: variance_scaling(factor=1, FAN_AVG, uniform)
    return init


class Variable(Tensor):
    def __init__(self, initial_value, name=None, trainable=True, dtype=None):
        idx = len(STATE.variables)
        if isinstance(initial_value, _Init):
            v = initial_draw(STATE.seed, idx, initial_value.kind, initial_value.shape, initial_value.scale)
            self.init_spec = (initial_value.kind, initial_value.shape, initial_value.scale)
        else:
            v = np.asarray(initial_value)
            self.init_spec = ("given", v.shape, 0.0)
        self.index, self.trainable = idx, trainable
        self.value = torch.tensor(v.astype(NPDT), dtype=DT, requires_grad=True)
        self.initial = self.value.detach().numpy().copy()
        super().__init__(lambda ctx: self.value, name=name or f"Variable_{idx}", dtype=float32)
        STATE.variables.append(self)

    def initialized_value(self):
        return self.initial.copy()

    def assign(self, value, use_locking=None, name=None):
        """an op: evaluated in a run, it replaces the variable's value (BUIR.py:122-123, the target tables' moving average)"""
        def f(ctx, v):
            with torch.no_grad():
                self.value.copy_(v.detach())
            return self.value
        return Tensor(f, [_t(value)], name="assign")


def global_variables_initializer():
    return Tensor(lambda ctx: None, name="init")           # variables hold their initial values from creation on


def all_variables():
    return list(STATE.variables)


# ---- element-wise / reductions / linear algebra ---------------------------------------------------------------------
exp = _unary(torch.exp)
log = _unary(torch.log)
sigmoid = _unary(torch.sigmoid)
# {(Session.run index, sign op index): int8 pattern}: a replay of a recorded run in other arithmetic (gen_golden_tf.py --float64-yardstick)
# takes the recorded run's signs instead of re-deciding them -- sign() is the one discontinuous op of SimGCL's graph
FORCED_SIGNS = {}


def sign(x, name=None):
    """tf.sign.  Its OUTPUT is recorded per Session.run (STATE.sign_log): sign() is discontinuous, so an input within rounding of
    zero may come out with the other sign in another float32 evaluation of the same graph -- a test that wants to follow a recorded
    run past such a step feeds the recorded pattern to the implementation under test (SimGCL.py:35)."""
    op_index = STATE.n_sign_ops
    STATE.n_sign_ops += 1

    def f(ctx, a):
        forced = FORCED_SIGNS.get((ctx.get("__run__"), op_index))
        out = torch.sign(a) if forced is None else torch.as_tensor(np.asarray(forced)).to(a.dtype)
        if ctx.get("__run__", (1 << 31) - 1) != (1 << 31) - 1:
            STATE.sign_log.setdefault(ctx["__run__"], {})[op_index] = out.detach().numpy().astype(np.int8)
        return out
    return Tensor(f, [_t(x)], name=name)


tanh = _unary(torch.tanh)
square = _unary(torch.square)
sqrt = _unary(torch.sqrt)
stop_gradient = _unary(lambda a: a.detach())


def multiply(x, y, name=None): return _binary(torch.mul, x, y)
def add(x, y, name=None): return _binary(torch.add, x, y)
def subtract(x, y, name=None): return _binary(torch.sub, x, y)
def divide(x, y, name=None): return _binary(torch.div, x, y)


def reduce_sum(x, axis=None, keepdims=False, name=None, keep_dims=None):
    kd = keepdims if keep_dims is None else keep_dims
    ax = _axes(axis)
    return Tensor(lambda ctx, a: a.sum() if ax is None else a.sum(dim=ax, keepdim=kd), [_t(x)])


def reduce_mean(x, axis=None, keepdims=False, name=None):
    ax = _axes(axis)
    return Tensor(lambda ctx, a: a.mean() if ax is None else a.mean(dim=ax, keepdim=keepdims), [_t(x)])


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    def f(ctx, x, y):
        return (x.T if transpose_a else x) @ (y.T if transpose_b else y)
    return Tensor(f, [_t(a), _t(b)])


def transpose(x, perm=None, name=None):
    return Tensor(lambda ctx, a: a.T if perm is None else a.permute(*perm), [_t(x)])


def concat(values, axis, name=None):
    vs = [_t(v) for v in values]
    return Tensor(lambda ctx, *xs: torch.cat(xs, dim=axis), vs)


def stack(values, axis=0, name=None):
    vs = [_t(v) for v in values]
    return Tensor(lambda ctx, *xs: torch.stack(xs, dim=axis), vs)


def split(value, num_or_size_splits, axis=0, name=None):
    v = _t(value)
    if isinstance(num_or_size_splits, int):
        whole = Tensor(lambda ctx, a: torch.chunk(a, num_or_size_splits, dim=axis), [v])
        n = num_or_size_splits
    else:
        sizes = [int(s) for s in num_or_size_splits]
        whole = Tensor(lambda ctx, a: torch.split(a, sizes, dim=axis), [v])
        n = len(sizes)
    return [Tensor(lambda ctx, parts, k=k: parts[k], [whole]) for k in _b.range(n)]


def reshape(x, shape, name=None):
    return Tensor(lambda ctx, a: a.reshape(tuple(int(s) for s in shape)), [_t(x)])


def tile(x, multiples, name=None):
    return Tensor(lambda ctx, a: a.repeat(*[int(m) for m in multiples]), [_t(x)])


def cast(x, dtype, name=None):
    def f(ctx, a):
        if dtype.is_bool:
            return a != 0
        return a.to(torch.int64) if dtype.is_int else a.to(DT)
    return Tensor(f, [_t(x)], dtype=dtype)


def cond(pred, true_fn=None, false_fn=None, name=None):
    t, f = _t(true_fn()), _t(false_fn())                  # both branches are built, as tf.cond does; one is evaluated

    def sel(ctx, p):
        take = _b.bool(p.item()) if isinstance(p, torch.Tensor) else _b.bool(p)
        return (t if take else f)._eval(ctx)
    return Tensor(sel, [_t(pred)])


def unique(x, out_idx=int32, name=None):
    """tf.unique: y = the distinct values in order of first occurrence, idx = position of every element in y"""
    def f(ctx, a):
        arr = a.detach().numpy()
        _, first, inv = np.unique(arr, return_index=True, return_inverse=True)
        order = np.argsort(first, kind="stable")
        rank = np.empty_like(order); rank[order] = np.arange(order.size)
        return torch.as_tensor(arr[np.sort(first)]), torch.as_tensor(rank[inv])
    whole = Tensor(f, [_t(x)])
    return (Tensor(lambda ctx, p: p[0], [whole], dtype=int64), Tensor(lambda ctx, p: p[1], [whole], dtype=int64))


def gather(params, indices, axis=0, name=None):
    return Tensor(lambda ctx, p, i: p.index_select(axis, i.reshape(-1)).reshape(*i.shape, *p.shape[1:]) if axis == 0 else p.index_select(axis, i), [_t(params), _t(indices)])


def shape(x, name=None):          # noqa: F811  (tf.shape: the dynamic shape)
    return Tensor(lambda ctx, a: torch.as_tensor(list(a.shape)), [_t(x)], dtype=int32)


def range(*args, **kw):           # noqa: A001
    def f(ctx, *a):
        return torch.arange(*[int(v) for v in a])
    return Tensor(f, [_t(a) for a in args], dtype=int32)


def diag_part(x, name=None):
    return Tensor(lambda ctx, a: torch.diagonal(a), [_t(x)])


def matrix_diag(x, name=None):
    return Tensor(lambda ctx, a: torch.diag_embed(a), [_t(x)])


def sparse_tensor_dense_matmul(sp_a, b, adjoint_a=False, adjoint_b=False, name=None):
    assert isinstance(sp_a, SparseTensor) and not adjoint_b
    if sp_a.dynamic:
        def f(ctx, idx, val, shp, x):
            A = SparseTensor._assemble(idx, val, shp.tolist())
            return torch.sparse.mm(A.t().coalesce() if adjoint_a else A, x)
        return Tensor(f, sp_a.parts + [_t(b)])
    A = sp_a._t.t().coalesce() if adjoint_a else sp_a._t
    return Tensor(lambda ctx, x: torch.sparse.mm(A, x), [_t(b)])


def _sparse_transpose(sp_a):
    assert not sp_a.dynamic
    return SparseTensor(np.ascontiguousarray(sp_a.indices[:, ::-1]), sp_a.values, sp_a.dense_shape[::-1])


# ---- random ops ------------------------------------------------------------------------------------------------------
def _random_node(shape_of, name):
    op_index = STATE.n_random_ops
    STATE.n_random_ops += 1

    def f(ctx, *a):
        shp = shape_of(*a)
        ctx.setdefault("__random__", []).append((op_index, name, tuple(int(s) for s in shp)))
        return torch.as_tensor(random_uniform(STATE.seed, ctx["__run__"], op_index, shp).astype(NPDT))
    return f, op_index


def _random_uniform(shape, minval=0, maxval=None, dtype=float32, seed=None, name=None):
    assert minval == 0 and maxval in (None, 1, 1.0)
    shp = tuple(int(s) for s in shape)
    f, k = _random_node(lambda: shp, "random_uniform")
    t = Tensor(f)
    t.random_op_index = k
    return t


def _random_shuffle(value, seed=None, name=None):
    """tf.random.shuffle along axis 0; the permutation is argsort of the op's uniform draw (documented, regenerable)"""
    f, k = _random_node(lambda a: (a.shape[0],), "random_shuffle")

    def g(ctx, a):
        u = f(ctx, a)
        return a[torch.argsort(u, stable=True)]
    t = Tensor(g, [_t(value)])
    t.random_op_index = k
    return t


def _dropout(x, keep_prob=None, noise_shape=None, seed=None, name=None, rate=None):
    """TF 1.14 nn_ops.dropout_v2: ret = x * (1 / keep_prob); keep_mask = random_uniform(shape) >= rate; ret * cast(mask).
    dropout(x, keep_prob=p) sets rate = 1 - p (python float arithmetic)."""
    if rate is None:
        rate = 1.0 - keep_prob
    f, k = _random_node(lambda a: a.shape, "dropout")

    def g(ctx, a):
        u = f(ctx, a)
        scale = torch.as_tensor(1.0 / (1.0 - rate), dtype=DT)
        return (a * scale) * (u >= torch.as_tensor(rate, dtype=DT)).to(DT)
    t = Tensor(g, [_t(x)])
    t.random_op_index = k
    return t


# ---- tf.nn -----------------------------------------------------------------------------------------------------------
def _embedding_lookup(params, ids, partition_strategy="mod", name=None, validate_indices=True, max_norm=None):
    return Tensor(lambda ctx, p, i: p[i.to(torch.int64)], [_t(params), _t(ids)])


def _l2_loss(t, name=None):
    return Tensor(lambda ctx, a: (a * a).sum() / 2, [_t(t)])


def _l2_normalize(x, axis=None, epsilon=1e-12, name=None, dim=None):
    """nn_impl.l2_normalize: square_sum = reduce_sum(square(x), axis, keepdims=True);
    x_inv_norm = rsqrt(maximum(square_sum, epsilon)); x * x_inv_norm"""
    ax = _axes(dim if axis is None else axis)

    def f(ctx, a):
        ss = (a * a).sum(dim=ax, keepdim=True)
        return a * torch.rsqrt(torch.maximum(ss, torch.as_tensor(epsilon, dtype=DT)))
    return Tensor(f, [_t(x)])


def _leaky_relu(features, alpha=0.2, name=None):
    return Tensor(lambda ctx, a: torch.maximum(alpha * a, a), [_t(features)])      # nn_ops.leaky_relu: max(alpha * x, x)


def _softmax(logits, axis=-1, name=None):
    return Tensor(lambda ctx, a: torch.softmax(a, dim=axis), [_t(logits)])


def _top_k(input, k=1, sorted=True, name=None):      # noqa: A002
    """nn_ops.top_k: "If two elements are equal, the lower-index element appears first" -- torch.topk gives no such promise
    (found by tests/test_tf1shim.py's anchor from TF's topk_op_test.py), a stable descending sort does"""
    def f(ctx, a):
        vals, idx = torch.sort(a, dim=-1, descending=True, stable=True)
        return vals[..., :int(k)], idx[..., :int(k)]
    whole = Tensor(f, [_t(input)])
    return (Tensor(lambda ctx, p: p[0], [whole]), Tensor(lambda ctx, p: p[1], [whole], dtype=int32))


nn = types.SimpleNamespace(embedding_lookup=_embedding_lookup, l2_loss=_l2_loss, l2_normalize=_l2_normalize, leaky_relu=_leaky_relu,
                           dropout=_dropout, softmax=_softmax, sigmoid=sigmoid, tanh=tanh, relu=_unary(torch.relu))
math = types.SimpleNamespace(l2_normalize=_l2_normalize, top_k=_top_k, log=log, exp=exp)
random = types.SimpleNamespace(uniform=_random_uniform, shuffle=_random_shuffle)
sparse = types.SimpleNamespace(transpose=_sparse_transpose)
contrib = types.SimpleNamespace(layers=types.SimpleNamespace(xavier_initializer=_xavier_initializer))
random_uniform_op = _random_uniform


# ---- tf.train.AdamOptimizer ------------------------------------------------------------------------------------------
class _TrainOp(Tensor):
    def __init__(self, opt, loss, var_list):
        super().__init__(lambda ctx: None, name="train")
        self.opt, self.loss, self.var_list = opt, loss, var_list


class AdamOptimizer:
    """training/adam.py + core/kernels/training_ops.cc ApplyAdam (float32 throughout):
        alpha = lr * sqrt(1 - beta2_power) / (1 - beta1_power)
        m += (g - m) * (1 - beta1);  v += (g * g - v) * (1 - beta2);  var -= (m * alpha) / (sqrt(v) + epsilon)
    beta powers start at beta1 / beta2 and are multiplied once per step AFTER the variables are updated (_finish).
    A variable the loss does not depend on gets no update (compute_gradients returns None for it)."""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, use_locking=False, name="Adam"):
        f = NPDT
        self.lr, self.b1, self.b2, self.eps = f(learning_rate), f(beta1), f(beta2), f(epsilon)
        self.b1p, self.b2p = f(beta1), f(beta2)
        self.slots = {}

    def minimize(self, loss, global_step=None, var_list=None, name=None):
        return _TrainOp(self, loss, var_list)

    def _apply(self, loss_value, variables):
        f = NPDT
        vs = [v for v in variables if v.trainable]
        grads = torch.autograd.grad(loss_value, [v.value for v in vs], allow_unused=True, retain_graph=True)
        alpha = f(self.lr * np.sqrt(f(1) - self.b2p, dtype=f) / (f(1) - self.b1p))
        applied, applied_idx = {}, {}          # by name (names may repeat: iterativeRecommender.py:47-48) and by creation index
        with torch.no_grad():
            for v, g in zip(vs, grads):
                if g is None:
                    continue
                m, s = self.slots.setdefault(v, (torch.zeros_like(v.value), torch.zeros_like(v.value)))
                m += (g - m) * float(f(1) - self.b1)
                s += (g * g - s) * float(f(1) - self.b2)
                v.value -= (m * float(alpha)) / (torch.sqrt(s) + float(self.eps))
                applied[v.name] = g.detach().numpy().copy()
                applied_idx[v.index] = applied[v.name]
        self.b1p, self.b2p = f(self.b1p * self.b1), f(self.b2p * self.b2)
        self.last_grads, self.last_grads_by_index = applied, applied_idx


train = types.SimpleNamespace(AdamOptimizer=AdamOptimizer)


# ---- Session ---------------------------------------------------------------------------------------------------------
class ConfigProto:
    def __init__(self, **kw):
        self.gpu_options = types.SimpleNamespace(allow_growth=False, per_process_gpu_memory_fraction=1.0)


def _feed(t, v):
    if isinstance(t, Tensor) and t.dtype is not None and t.dtype.is_bool:
        return _b.bool(v)
    a = np.asarray(v)
    if a.dtype.kind in "iub" or (isinstance(t, Tensor) and t.dtype is not None and t.dtype.is_int):
        return torch.as_tensor(a.astype(np.int64))
    return torch.as_tensor(a.astype(NPDT))


def _out(v):
    if v is None:
        return None
    if isinstance(v, tuple):
        return tuple(_out(e) for e in v)
    if isinstance(v, torch.Tensor):
        a = v.detach().numpy()
        return a.copy() if a.ndim else a[()]
    return v


class Session:
    def __init__(self, target="", graph=None, config=None):
        self.config = config

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def close(self):
        pass

    def run(self, fetches, feed_dict=None, options=None, run_metadata=None):
        run_index = STATE.run_index
        STATE.run_index += 1
        ctx = {"__run__": run_index}
        for k, v in (feed_dict or {}).items():
            ctx[k] = _feed(k, v)
        single = not isinstance(fetches, (list, tuple))
        fl = [fetches] if single else list(fetches)
        # every fetch sees the variables as they were when the call started: losses fetched next to a train op are
        # pre-update values, as in TF (the update is ordered after the gradient computation, nothing after the update)
        vals = [None if isinstance(t, _TrainOp) else t._eval(ctx) for t in fl]
        outs = [_out(v) for v in vals]
        for t in fl:
            if isinstance(t, _TrainOp):
                t.opt._apply(t.loss._eval(ctx), t.var_list or STATE.variables)
        STATE.run_log.append((run_index, list(ctx.get("__random__", []))))
        return outs[0] if single else outs
