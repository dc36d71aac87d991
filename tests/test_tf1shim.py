"""Known answers for tests/golden/tf1shim.py, the stand-in for the tensorflow 1.14 module that the TF-path golden vectors are
generated through (tests/golden/gen_golden_tf.py).  Each case is an example or a definition from TensorFlow 1.14's own API
documentation / python sources, evaluated through the stand-in's Session.run -- so that what the fixtures rest on (the
primitive ops' semantics) is itself checked against the published behaviour, op by op."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import tf1shim as tf  # noqa: E402


@pytest.fixture(autouse=True)
def _fresh_graph():
    tf.reset(1)
    yield


def run(*fetches, feed=None):
    with tf.Session() as s:
        out = s.run(list(fetches), feed_dict=feed or {})
    return out[0] if len(out) == 1 else out


def test_unique_is_the_api_docs_example():
    # tf.unique docs: x = [1, 1, 2, 4, 4, 4, 7, 8, 8] -> y = [1, 2, 4, 7, 8], idx = [0, 0, 1, 2, 2, 2, 3, 4, 4]
    x = tf.placeholder(tf.int32)
    y, idx = tf.unique(x)
    yv, iv = run(y, idx, feed={x: [1, 1, 2, 4, 4, 4, 7, 8, 8]})
    assert yv.tolist() == [1, 2, 4, 7, 8] and iv.tolist() == [0, 0, 1, 2, 2, 2, 3, 4, 4]
    yv = run(y, feed={x: [5, 3, 5, 9, 3]})                 # first-appearance order, not sorted
    assert yv.tolist() == [5, 3, 9]


def test_l2_loss_l2_normalize_leaky_relu_definitions():
    x = tf.placeholder(tf.float32)
    assert run(tf.nn.l2_loss(x), feed={x: [[1.0, 2.0], [3.0, 4.0]]}) == pytest.approx(15.0)            # sum(t ** 2) / 2
    z = run(tf.nn.l2_normalize(x, 1), feed={x: [[3.0, 4.0], [0.0, 0.0]]})
    np.testing.assert_allclose(z, [[0.6, 0.8], [0.0, 0.0]], rtol=1e-6)                                   # x * rsqrt(max(sum x^2, 1e-12)): a zero row stays zero
    z = run(tf.math.l2_normalize(x, axis=1), feed={x: [[1e-8, 0.0]]})                                    # below the clamp: x / 1e-6
    np.testing.assert_allclose(z, [[1e-2, 0.0]], rtol=1e-5)
    np.testing.assert_allclose(run(tf.nn.leaky_relu(x), feed={x: [-2.0, 0.0, 3.0]}), [-0.4, 0.0, 3.0], rtol=1e-6)   # alpha = 0.2 by default


def test_embedding_lookup_gradient_accumulates_duplicates():
    E = tf.Variable(np.arange(12, dtype=np.float32).reshape(4, 3), name="E")
    ids = tf.placeholder(tf.int32)
    loss = tf.reduce_sum(tf.nn.embedding_lookup(E, ids) * np.float32(2.0))
    opt = tf.train.AdamOptimizer(0.1)
    train = opt.minimize(loss)
    run(train, loss, feed={ids: [1, 1, 3]})
    g = opt.last_grads["E"]
    np.testing.assert_array_equal(g, [[0, 0, 0], [4, 4, 4], [0, 0, 0], [2, 2, 2]])                       # IndexedSlices summed per row


def test_adam_first_steps_match_the_published_update():
    # training/adam.py docstring: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t); m_t = b1 m + (1 - b1) g; v_t = b2 v + (1 - b2) g^2;
    # variable -= lr_t * m_t / (sqrt(v_t) + epsilon)
    w0 = np.array([0.5, -1.5, 2.0], np.float32)
    w = tf.Variable(w0.copy(), name="w")
    loss = tf.reduce_sum(tf.multiply(w, w)) * np.float32(0.5)      # gradient = w
    train = tf.train.AdamOptimizer(0.01).minimize(loss)
    ref, m, v = w0.astype(np.float64), 0.0, 0.0
    with tf.Session() as s:
        for t in range(1, 6):
            s.run(train)
            g = ref.copy()
            m = 0.9 * m + 0.1 * g; v = 0.999 * v + 0.001 * g * g
            ref = ref - 0.01 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-8)
            np.testing.assert_allclose(s.run(w), ref, rtol=2e-6)
    # a variable the loss does not touch gets no update (compute_gradients -> None)
    tf.reset(2)
    a = tf.Variable(np.ones(2, np.float32), name="a"); b = tf.Variable(np.ones(2, np.float32), name="b")
    tr = tf.train.AdamOptimizer(0.1).minimize(tf.reduce_sum(a))
    av, bv = run(tr, a, b)[1:]
    av2, bv2 = run(a, b)
    assert (av == 1).all() and (av2 < 1).all() and (bv2 == 1).all()      # the fetch next to the train op is the pre-update value


def test_dropout_scales_then_masks_and_keeps_where_uniform_reaches_the_rate():
    x = tf.placeholder(tf.float32)
    d = tf.nn.dropout(x, keep_prob=0.9)
    v = np.ones((2000, 8), np.float32)
    with tf.Session() as s:
        out = s.run(d, feed_dict={x: v})
        u = tf.random_uniform(1, 0, d.random_op_index, v.shape)
    keep = u >= np.float32(1.0 - 0.9)
    np.testing.assert_array_equal(out != 0, keep)
    np.testing.assert_allclose(out[keep], np.float32(1.0) * np.float32(1.0 / (1.0 - (1.0 - 0.9))), rtol=1e-6)
    assert abs(keep.mean() - 0.9) < 0.01


def test_sparse_matmul_split_concat_cond_topk_softmax():
    A = tf.SparseTensor(indices=[[0, 1], [1, 0], [1, 2]], values=[2.0, 3.0, 4.0], dense_shape=[2, 3])
    X = tf.placeholder(tf.float32)
    xv = np.arange(6, dtype=np.float32).reshape(3, 2)
    np.testing.assert_allclose(run(tf.sparse_tensor_dense_matmul(A, X), feed={X: xv}), [[4, 6], [16, 23]])
    dense = np.array([[0, 2, 0], [3, 0, 4]], np.float32)
    np.testing.assert_allclose(run(tf.sparse_tensor_dense_matmul(tf.sparse.transpose(A), X), feed={X: xv[:2]}), dense.T @ xv[:2])     # MHCN.py:156
    a, b = tf.split(tf.concat([X, X], axis=0), [2, 4], 0)
    av, bv = run(a, b, feed={X: xv})
    assert av.shape == (2, 2) and bv.shape == (4, 2) and (bv[1:] == xv).all()
    flag = tf.cast(tf.placeholder(tf.int32), tf.bool)
    c = tf.cond(flag, lambda: X * 2.0, lambda: X)
    assert (run(c, feed={X: xv, flag: 1}) == 2 * xv).all() and (run(c, feed={X: xv, flag: 0}) == xv).all()      # feeding the cast tensor itself, as NGCF.py:60 does
    vals, idx = tf.math.top_k(X, 2)
    iv = run(idx, feed={X: [[0.1, 0.9, 0.5], [3.0, 1.0, 2.0]]})
    assert iv.tolist() == [[1, 2], [0, 2]]
    np.testing.assert_allclose(run(tf.nn.softmax(X), feed={X: [[0.0, np.log(3.0)]]}), [[0.25, 0.75]], rtol=1e-6)


def test_reduce_mean_of_a_list_stacks_it_and_variable_assign_is_an_op():
    a = tf.Variable(np.array([1.0, 2.0], np.float32)); b = tf.Variable(np.array([3.0, 6.0], np.float32))
    np.testing.assert_allclose(run(tf.reduce_mean([a, b], axis=0)), [2.0, 4.0])      # LightGCN.py:19
    upd = b.assign(b * 0.5 + a * 0.5)
    assert (run(b) == [3.0, 6.0]).all()               # building the op changes nothing
    run(upd)
    np.testing.assert_allclose(run(b), [2.0, 4.0])
    np.testing.assert_allclose(tf.Variable(a.initialized_value()).initial, [1.0, 2.0])
