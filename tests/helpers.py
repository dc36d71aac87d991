"""Shared test helpers: golden fixtures -> inputs for the product's host API."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    meta = json.load(open(os.path.join(GOLDEN, "golden_meta.json")))[name]
    return meta, np.load(os.path.join(GOLDEN, name + ".npz"))


def conf_from_text(text: str):
    from qrec_amd.util.config import ModelConf
    d = {}
    for line in text.strip().splitlines():
        k, v = line.strip().split("=")
        d[k] = v
    return ModelConf.from_dict(d)


def rows_from_golden(z, with_test=True):
    """Rebuild the reference's trainingSet/testSet row lists from id arrays.  Row order is
    preserved, so first-appearance id assignment reproduces the same ids."""
    train = [[f"u{u}", f"i{i}", float(r)] for u, i, r in zip(z["train_uid"].tolist(), z["train_iid"].tolist(), z["train_r"].tolist())]
    test = []
    if with_test:
        for u, i, un, inn in zip(z["test_uid"].tolist(), z["test_iid"].tolist(), z["test_uname"].tolist(), z["test_iname"].tolist()):
            test.append([f"u{u}" if u >= 0 else f"xu{un}", f"i{i}" if i >= 0 else f"xi{inn}", 1.0])
    return train, test


def pad_cols(a, ld):
    out = np.zeros((a.shape[0], ld), dtype=a.dtype)
    out[:, :a.shape[1]] = a
    return out


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


# ---- parity ledger -------------------------------------------------------------------------------------------------------
# Every numerical-agreement assertion of the GPU tests goes through ``check``: it asserts ``observed < bound`` AND, when
# QREC_PARITY_LOG names a file, appends {test, quantity, observed, bound, kind} to it as one JSON line.  A GPU run with the variable
# set leaves the observed error of every comparison behind (tools/summarize_parity.py folds the lines into
# profiles/rNN_parity_errors.json: per test and quantity, the worst observed value next to the bound it was held to).
# ``kind`` says what sort of statement the row is, so that the summary can count the parity contract on its own:
#   parity        same inputs, same algorithm, the reference's (or the oracle's) result: north_star's 1e-5 / bit-exact contract
#   floor         a bound derived from the reference's OWN distance to exact arithmetic (tf_f64_yardstick.npz): the quantity cannot
#                 be closer to the fixture than the fixture is to the truth
#   discontinuity a run that may legitimately leave the recorded one at a discontinuous op (sign(), top_k) -- the recorded-pattern
#                 variant of the same test is the parity row
#   statistical   throughput-mode quantities (another random stream / visiting order): Recall@20 and loss gaps, means over seeds
#   partition     the same step / run of THIS repository under another partition of the rows or of the batch over ranks, against its own
#                 single-GPU run (not against the reference): another fp32 summation structure, carried through the Adam steps in between
#   info          recorded, never a contract (bound is a sanity ceiling)
KINDS = ("parity", "floor", "discontinuity", "statistical", "partition", "info")


def check(quantity, observed, bound, ctx=None, inclusive=False, kind="parity"):
    assert kind in KINDS, kind
    observed = float(observed); bound = float(bound)
    log = os.environ.get("QREC_PARITY_LOG")
    if log:
        test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
        with open(log, "a") as f:
            f.write(json.dumps(dict(test=test, quantity=quantity, observed=observed, bound=bound, kind=kind)) + "\n")
    ok = observed <= bound if inclusive else observed < bound
    assert ok, f"{quantity}: observed {observed:.3e}, bound {bound:.1e}" + (f" [{ctx}]" if ctx is not None else "")


def check_rel(quantity, got, want, rel, ctx=None, abs_tol=0.0, kind="parity"):
    """element-wise |got - want| <= rel * |want| (+ abs_tol), recorded as the worst ratio |got - want| / (|want| + abs_tol / rel)"""
    got = np.asarray(got, dtype=np.float64); want = np.asarray(want, dtype=np.float64)
    den = np.abs(want) + (abs_tol / rel if abs_tol else 0.0)
    worst = float(np.max(np.abs(got - want) / np.maximum(den, 1e-300))) if got.size else 0.0
    check(quantity, worst, rel, ctx=ctx, inclusive=True, kind=kind)


def simgcl_recorded_signs(z, step):
    """the sign pattern (-1 / 0 / +1, int8 [N, d]) each of the four perturbations of training step ``step`` used in the recorded run of
    the reference's SimGCL (tests/golden/tf_simgcl_filmtrust.npz: sign ops in creation order = view 1 layer 1, layer 2, view 2 layer 1, 2)"""
    shape = tuple(int(x) for x in z["sign_shape"])
    n = int(np.prod(shape))
    neg = np.unpackbits(z["sign_neg_bits"])[:n].reshape(shape)[step].astype(bool)
    zero = np.unpackbits(z["sign_zero_bits"])[:n].reshape(shape)[step].astype(bool)
    sg = np.where(neg, -1, 1).astype(np.int8)
    sg[zero] = 0
    return [sg[k] for k in range(shape[1])]


def encode_forced_signs(noise, sign):
    """injected noise that carries the sign to use (include/qrec_hip.h, qrec_perturb_rows): u -> +-(2 + u) / 4 + u"""
    noise = np.asarray(noise, dtype=np.float32)
    return np.where(sign > 0, 2 + noise, np.where(sign < 0, -(2 + noise), 4 + noise)).astype(np.float32)


def same_bits(what, a: dict, b: dict):
    """two runs of a parity-mode trainer from the same inputs: every returned array bit-identical"""
    assert a.keys() == b.keys()
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.shape == y.shape and x.dtype == y.dtype, (what, k)
        diff = int(np.count_nonzero(x.view(np.uint8) != y.view(np.uint8))) if x.size else 0
        check(f"{what}: bytes of '{k}' that differ between two runs from the same inputs (ordered reductions)", diff, 0, inclusive=True)
