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
