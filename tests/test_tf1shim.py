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


# ---------------------------------------------------------------------------------------------------------------------
# Independent anchors (round 3): expected values that are NOT derived in this repository.  Each case restates a test of
# TensorFlow's own test-suite (r1.14 branch) -- its input vectors, and as the expectation the reference implementation that
# TF's test itself compares against, copied verbatim -- or an example printed in TF's API documentation.  A misconception
# shared by this stand-in and by oracle/tfmodels.py (both written here) cannot pass these.
# ---------------------------------------------------------------------------------------------------------------------
def adam_update_numpy(param, g_t, t, m, v, alpha=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
    """tensorflow/python/training/adam_test.py (r1.14), lines 41-49, verbatim: the numpy reference TF's own AdamOptimizer
    tests are held to"""
    alpha_t = alpha * np.sqrt(1 - beta2**t) / (1 - beta1**t)

    m_t = beta1 * m + (1 - beta1) * g_t
    v_t = beta2 * v + (1 - beta2) * g_t * g_t

    param_t = param - alpha_t * m_t / (np.sqrt(v_t) + epsilon)
    return param_t, m_t, v_t


def test_adam_is_tfs_own_adam_test_basic():
    """adam_test.py::AdamOptimizerTest.doTestBasic (r1.14): var0 = [1.0, 2.0], grads0 = [0.1, 0.1], var1 = [3.0, 4.0],
    grads1 = [0.01, 0.01], default hyper-parameters, three steps; after step t the variables equal adam_update_numpy's and
    the beta powers are 0.9**(t+1), 0.999**(t+1) (TF asserts the powers BEFORE each step: 0.9**t, 0.999**t)."""
    var0_np, grads0_np = np.array([1.0, 2.0], np.float32), np.array([0.1, 0.1], np.float32)
    var1_np, grads1_np = np.array([3.0, 4.0], np.float32), np.array([0.01, 0.01], np.float32)
    var0, var1 = tf.Variable(var0_np.copy(), name="var0"), tf.Variable(var1_np.copy(), name="var1")
    # constant gradients, as the TF test feeds them: loss = sum(var0 * grads0) + sum(var1 * grads1)
    loss = tf.reduce_sum(tf.multiply(var0, grads0_np)) + tf.reduce_sum(tf.multiply(var1, grads1_np))
    opt = tf.train.AdamOptimizer()
    update = opt.minimize(loss)
    m0, v0, m1, v1 = 0.0, 0.0, 0.0, 0.0
    with tf.Session() as s:
        for t in range(1, 4):
            assert opt.b1p == pytest.approx(0.9 ** t, rel=1e-6) and opt.b2p == pytest.approx(0.999 ** t, rel=1e-6)
            s.run(update)
            var0_np, m0, v0 = adam_update_numpy(var0_np, grads0_np, t, m0, v0)
            var1_np, m1, v1 = adam_update_numpy(var1_np, grads1_np, t, m1, v1)
            np.testing.assert_allclose(s.run(var0), var0_np, rtol=1e-6)          # assertAllCloseAccordingToType: float32 -> 1e-6
            np.testing.assert_allclose(s.run(var1), var1_np, rtol=1e-6)


def test_adam_sparse_is_tfs_own_adam_test_sparse_and_repeated_indices():
    """adam_test.py::testSparse: the same vectors through IndexedSlices(grads, indices=[0, 1]) -- here: the gradient of an
    embedding_lookup, which is what produces IndexedSlices in the reference's models (BPR.py:80).
    ::testSparseRepeatedIndices: var = [[1.0], [2.0]]; the gradient [[0.1], [0.1]] at indices [1, 1] must move the variable
    exactly like the aggregated gradient [[0.2]] at index [1], over three steps."""
    var0_np, grads0_np = np.array([1.0, 2.0], np.float32), np.array([0.1, 0.1], np.float32)
    var0 = tf.Variable(var0_np.reshape(2, 1).copy(), name="var0")
    ids = tf.placeholder(tf.int32)
    loss = tf.reduce_sum(tf.nn.embedding_lookup(var0, ids) * grads0_np.reshape(2, 1))
    update = tf.train.AdamOptimizer().minimize(loss)
    m0 = v0 = 0.0
    with tf.Session() as s:
        for t in range(1, 4):
            s.run(update, feed_dict={ids: [0, 1]})
            var0_np, m0, v0 = adam_update_numpy(var0_np, grads0_np, t, m0, v0)
            np.testing.assert_allclose(s.run(var0).ravel(), var0_np, rtol=1e-6)
    tf.reset(3)
    rep = tf.Variable(np.array([[1.0], [2.0]], np.float32), name="repeated")
    agg = tf.Variable(np.array([[1.0], [2.0]], np.float32), name="aggregated")
    ids = tf.placeholder(tf.int32)
    up_rep = tf.train.AdamOptimizer().minimize(tf.reduce_sum(tf.nn.embedding_lookup(rep, ids) * np.float32(0.1)))
    up_agg = tf.train.AdamOptimizer().minimize(tf.reduce_sum(tf.nn.embedding_lookup(agg, [1]) * np.float32(0.2)))
    with tf.Session() as s:
        np.testing.assert_array_equal(s.run(rep), s.run(agg))
        for _ in range(3):
            s.run(up_rep, feed_dict={ids: [1, 1]}); s.run(up_agg)
            np.testing.assert_allclose(s.run(rep), s.run(agg), rtol=1e-7)       # TF: assertAllClose
        # TF 1.14's sparse Adam (_apply_sparse_shared) updates m, v and the variable over ALL rows (m_t = assign(m, m * beta1) ...
        # var_update = assign_sub(var, lr * m_t / (sqrt(v_t) + epsilon))): a row that never received a gradient has m = v = 0 and
        # stays where it is, in TF as in the dense functor the stand-in applies
        assert s.run(rep)[0, 0] == 1.0 and s.run(agg)[0, 0] == 1.0


def test_top_k_is_tfs_own_topk_op_test_including_the_tie_rule():
    """tensorflow/python/kernel_tests/topk_op_test.py (r1.14): testTop2 / testTop3 / testTopAll vectors, and the documented
    tie rule of nn_ops.top_k ("If two elements are equal, the lower-index element appears first") on its stable-sort vector."""
    X = tf.placeholder(tf.float32)
    def topk(x, k):
        v, i = tf.math.top_k(X, k)
        vv, iv = run(v, i, feed={X: x})
        return np.asarray(vv).tolist(), np.asarray(iv).tolist()
    inputs = [[0.1, 0.3, 0.2, 0.4], [0.1, 0.3, 0.4, 0.2]]
    v, i = topk(inputs, 2)
    assert i == [[3, 1], [2, 1]]; np.testing.assert_allclose(v, [[0.4, 0.3], [0.4, 0.3]], rtol=1e-6)           # testTop2
    v, i = topk(inputs, 3)
    assert i == [[3, 1, 2], [2, 1, 3]]; np.testing.assert_allclose(v, [[0.4, 0.3, 0.2], [0.4, 0.3, 0.2]], rtol=1e-6)     # testTop3
    ties = [[0.1, 0.3, 0.2, 0.4], [0.1, 0.3, 0.3, 0.2]]
    v, i = topk(ties, 4)
    assert i == [[3, 1, 2, 0], [1, 2, 3, 0]]                                                                    # testTopAll: equal 0.3s -> index 1 before 2
    v, i = topk([[5.0, 5.0, 5.0, 5.0, 5.0, 5.0, 5.0, 5.0]], 3)
    assert i == [[0, 1, 2]]                                                                                     # all tied: the lowest indices, ascending


def test_dropout_is_tfs_own_nn_test_dropout():
    """tensorflow/python/ops/nn_test.py::DropoutTest.testDropout (r1.14): a 40 x 30 tensor of ones, keep_prob in
    {0.1, 0.5, 0.8}: every output is either 0 or 1 / keep_prob, and the kept fraction is keep_prob within 15 % relative."""
    x_dim, y_dim, num_iter = 40, 30, 10
    for keep_prob in (0.1, 0.5, 0.8):
        tf.reset(7)
        t = tf.placeholder(tf.float32)
        d = tf.nn.dropout(t, keep_prob)
        final_count = 0
        with tf.Session() as s:
            for _ in range(num_iter):
                value = s.run(d, feed_dict={t: np.ones((x_dim, y_dim), np.float32)})
                final_count += np.count_nonzero(value)
                sorted_value = np.unique(np.sort(value))
                assert sorted_value[0] == 0
                np.testing.assert_allclose(1 / keep_prob, sorted_value[1], rtol=1e-6)
        expected_count = x_dim * y_dim * keep_prob * num_iter
        assert abs(final_count - expected_count) / expected_count < 0.15


def test_sparse_tensor_dense_matmul_sums_duplicate_entries_like_tfs_kernel():
    """tensorflow/core/kernels/sparse_tensor_dense_matmul_op.cc (r1.14), the CPU functor: ``for i in range(nnz): for n:
    out(m, n) += a_values(i) * b(k, n)`` with (m, k) = a_indices(i) -- entries with the same (m, k) ADD UP, in any order of
    the index list (no canonical ordering is required by this op).  base/graphRecommender.py:36-38 feeds scipy's index list."""
    A = tf.SparseTensor(indices=[[1, 2], [0, 1], [1, 2], [1, 0]], values=[4.0, 2.0, 0.5, 3.0], dense_shape=[2, 3])     # (1, 2) twice, unsorted
    X = tf.placeholder(tf.float32)
    xv = np.arange(6, dtype=np.float32).reshape(3, 2)
    want = np.zeros((2, 2), np.float32)
    for (m, k), a in zip([[1, 2], [0, 1], [1, 2], [1, 0]], [4.0, 2.0, 0.5, 3.0]):
        want[m] += np.float32(a) * xv[k]
    np.testing.assert_allclose(run(tf.sparse_tensor_dense_matmul(A, X), feed={X: xv}), want, rtol=1e-6)
    # sparse_ops.sparse_tensor_dense_matmul docstring: "A is sparse, B is dense; computes A * B" with adjoint_a = False
    np.testing.assert_allclose(want, np.array([[0, 2, 0], [3, 0, 4.5]], np.float32) @ xv, rtol=1e-6)


def test_l2_normalize_is_tfs_own_nn_test_reference():
    """nn_test.py::L2NormalizeTest._l2Normalize (r1.14): ``norm = np.apply_along_axis(np.linalg.norm, dim, x); return x /
    np.expand_dims(norm, dim)`` on a random [20, 7, 3] tensor, every dim"""
    rng = np.random.RandomState(0)
    x_np = rng.random_sample((20, 7, 3)).astype(np.float32)
    for dim in range(3):
        norm = np.apply_along_axis(np.linalg.norm, dim, x_np)
        want = x_np / np.expand_dims(norm, dim)
        X = tf.placeholder(tf.float32)
        np.testing.assert_allclose(run(tf.nn.l2_normalize(X, dim), feed={X: x_np}), want, rtol=1e-5)
