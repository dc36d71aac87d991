"""INTEGRATION.md s2, executed: the stub a QRec maintainer would add (``model/ranking/BPR.py`` binding libqrec_hip.so with
ctypes) is cut out of INTEGRATION.md, dropped into an overlay of the REAL reference tree (every other file is the
reference's own, by symlink: base/, data/, util/, QRec.py, the datasets) and run through the reference's own
``QRec(conf).execute()`` on the FilmTrust conf of the recorded golden run.  It must reproduce that run of the unmodified
reference: final P and Q, every epoch's loss and learning rate, the measure strings, the state of Python's generator.

There is no GPU where the reference tree is (and no reference tree where the GPU is), so the device half of the C ABI is
answered by tests/integration/shim.c: host entry points forwarded to the real libqrec_hip.so, the SGD / sum-of-squares
entry points by the oracle's C restatement.  The GPU kernels behind those two symbols are held to the same oracle and the
same golden run in tests/test_gpu_bpr.py; this test is about the binding."""
import json
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "base")) or shutil.which("gcc") is None,
                                reason="needs the reference tree (build container only) and gcc")

RUNNER = r'''
import io, json, os, random, sys, types
from contextlib import redirect_stdout
import numpy as np
nb = types.ModuleType("numba"); nb.jit = lambda *a, **k: (lambda f: f)
mkl = types.ModuleType("mkl"); mkl.set_num_threads = lambda n: None; mkl.get_max_threads = lambda: 1
sys.modules.update(numba=nb, mkl=mkl, tensorflow=types.ModuleType("tensorflow"))
sys.dont_write_bytecode = True
sys.path.insert(0, os.getcwd())
from QRec import QRec
from util.config import ModelConf
import model.ranking.BPR as B
assert "QREC_HIP_LIB" in open(B.__file__).read()            # it IS the stub
epochs = []
orig = B.BPR.isConverged
def isConverged(self, epoch):
    before, lr = self.loss, self.lRate
    r = orig(self, epoch)
    epochs.append((float(before), float(lr), float(self.lRate)))
    return r
B.BPR.isConverged = isConverged
keep = {}
orig_eval = B.BPR.evalRanking
def evalRanking(self):
    keep["P"], keep["Q"] = self.P.copy(), self.Q.copy()
    r = orig_eval(self)
    keep["measure"] = list(self.measure)
    return r
B.BPR.evalRanking = evalRanking
seed = int(sys.argv[2])
random.seed(seed); np.random.seed(seed)
with redirect_stdout(io.StringIO()):
    QRec(ModelConf(sys.argv[1])).execute()
measure = keep["measure"]
np.savez(sys.argv[3], P=keep["P"], Q=keep["Q"], epochs=np.array(epochs), state=np.array(random.getstate()[1], dtype=np.uint64))
json.dump(measure, open(sys.argv[3] + ".json", "w"))
'''


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# model/ranking/BPR\.py.*?)```", text, re.S)
    assert block, "INTEGRATION.md no longer holds the model/ranking/BPR.py stub"
    return block.group(1)


def test_integration_md_stub_runs_inside_the_reference_tree_and_reproduces_the_reference_run(tmp_path):
    from helpers import load_golden
    meta, z = load_golden("bpr_filmtrust")
    # the overlay: the reference tree by symlink, model/ranking/BPR.py replaced by the stub
    tree = tmp_path / "tree"
    (tree / "model" / "ranking").mkdir(parents=True)
    for name in os.listdir(REF):
        if name not in ("model", ".git"):
            os.symlink(os.path.join(REF, name), tree / name)
    for name in os.listdir(os.path.join(REF, "model")):
        if name != "ranking":
            os.symlink(os.path.join(REF, "model", name), tree / "model" / name)
    for name in os.listdir(os.path.join(REF, "model", "ranking")):
        if name != "BPR.py":
            os.symlink(os.path.join(REF, "model", "ranking", name), tree / "model" / "ranking" / name)
    (tree / "model" / "ranking" / "BPR.py").write_text(_stub_source())
    (tree / "run_stub.py").write_text(RUNNER)
    (tree / "bpr.conf").write_text(meta["conf"])
    shim = tmp_path / "libqrec_shim.so"
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", str(shim), os.path.join(ROOT, "tests", "integration", "shim.c"), "-ldl"], check=True)
    from oracle import c as O
    O.lib()                                                       # builds oracle/libqrec_oracle.so if needed
    env = dict(os.environ, QREC_HIP_LIB=str(shim), QREC_REAL_LIB=os.path.join(ROOT, "qrec_amd", "libqrec_hip.so"),
               QREC_ORACLE_LIB=os.path.join(ROOT, "oracle", "libqrec_oracle.so"), PYTHONDONTWRITEBYTECODE="1")
    env.pop("PYTHONPATH", None)
    out = tmp_path / "out.npz"
    run = subprocess.run([sys.executable, "run_stub.py", "bpr.conf", str(meta["seed"]), str(out)], cwd=tree, env=env,
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    got = np.load(out)
    last = len(meta["epochs"])
    np.testing.assert_allclose(got["P"], z[f"P{last}"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(got["Q"], z[f"Q{last}"], rtol=1e-10, atol=1e-13)
    assert got["epochs"].shape[0] == last
    for (loss, lr_used, lr_next), want in zip(got["epochs"], meta["epochs"]):
        assert loss == pytest.approx(want["loss"], rel=1e-11) and lr_used == want["lr_used"] and lr_next == want["lr_next"]
    assert np.array_equal(got["state"].astype(np.uint32), z["py_state"])           # the generator was handed back in lock-step
    measure = json.load(open(str(out) + ".json"))
    assert [m.split(":")[0] for m in measure] == [m.split(":")[0] for m in meta["measure"]]
    for g, w in zip(measure, meta["measure"]):
        if ":" in g:
            assert float(g.split(":")[1]) == pytest.approx(float(w.split(":")[1]), rel=1e-9)
