#!/usr/bin/env python3
"""Wall-clock per epoch of the drop-in model CLASSES (sampler thread, uploads, training steps, per-epoch evaluation)
at the Yelp2018 shape -- what a user of `python -m qrec_amd.main` sees, next to the bare step times of the other tools."""
import io, json, os, sys, time
from contextlib import redirect_stdout
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["QREC_QUIET"] = "1"
from helpers import conf_from_text
from qrec_amd import capi
from qrec_amd.data.rows import RatingRows
from qrec_amd.synth import make_dataset
import importlib
capi.init(0)
d = make_dataset("yelp2018"); nu, ni = d["n_users"], d["n_items"]
un, inn = [f"u{k}" for k in range(nu)], [f"i{k}" for k in range(ni)]
train = RatingRows(d["train_u"], d["train_i"], np.ones(d["train_u"].size), un, inn)
test = RatingRows(d["test_u"], d["test_i"], np.ones(d["test_u"].size), un, inn)
EXTRA = {"LightGCN": "LightGCN=-n_layer 3", "SimGCL": "SimGCL=-n_layer 2 -lambda 0.5 -eps 0.1", "NGCF": "",
         "SGL": "SGL=-n_layer 3 -lambda 0.1 -droprate 0.1 -augtype 1 -temp 0.2", "BUIR": "BUIR=-n_layer 2 -tau 0.995 -drop_rate 0.5"}
out = {}
for name in sys.argv[1:] or ["LightGCN", "SimGCL", "BUIR"]:
    cls = getattr(importlib.import_module(f"qrec_amd.model.ranking.{name}"), name)
    E = int(os.environ.get("QREC_BENCH_EPOCHS", "4"))
    conf = conf_from_text(f"ratings=./x.txt\nratings.setup=-columns 0 1 2\nmodel.name={name}\nevaluation.setup=-testSet x -b 1\nitem.ranking=on -topN 20\n"
                          f"num.factors=64\nnum.max.epoch={E}\nbatch_size=2048\nlearnRate=-init 0.001 -max 1\nreg.lambda=-u 0.0001 -i 0.0001 -b 0.2 -s 0.2\n"
                          f"output.setup=off -dir ./results/\n{EXTRA[name]}".strip())
    np.random.seed(1); import random; random.seed(1)
    with redirect_stdout(io.StringIO()):
        t0 = time.perf_counter(); m = cls(conf, train, test); m.readConfiguration(); m.initModel(); t_init = time.perf_counter() - t0
        marks = []                                   # (start, end) of every per-epoch evaluation, device drained at both
        inner = m.ranking_performance
        def timed(epoch):
            capi.device_sync(); a = time.perf_counter(); r = inner(epoch); capi.device_sync(); marks.append((a, time.perf_counter())); return r
        m.ranking_performance = timed
        t0 = time.perf_counter(); m.trainModel(); capi.device_sync(); t_train = time.perf_counter() - t0
    train_parts = ([marks[0][0] - t0] + [marks[k][0] - marks[k - 1][1] for k in range(1, len(marks))]) if marks else [t_train]   # no marks: the class does not evaluate per epoch
    out[name] = dict(init_s=round(t_init, 3), train_s=round(t_train, 3), s_per_epoch=round(t_train / E, 3), steps_per_epoch=-(-len(train) // 2048), epochs=E,
                     mode=os.environ.get("QREC_MODE", "exact"), train_s_by_epoch=[round(x, 3) for x in train_parts],
                     eval_s_by_epoch=[round(b - a, 3) for a, b in marks])
print(json.dumps(out))
