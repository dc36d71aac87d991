#!/usr/bin/env python3
"""Per-epoch evaluation (ranking_performance, base/iterativeRecommender.py:115-185) at the Yelp2018 shape through
the drop-in model API: lists + Measure.rankingMeasure on the host vs hit counts / DCG sums taken on the device."""
import io, json, os, sys, time
from contextlib import redirect_stdout
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import conf_from_text
from qrec_amd import capi
from qrec_amd.model.ranking.BPR import BPR
from qrec_amd.synth import make_dataset
from qrec_amd.util.measure import Measure
capi.init(0)
d = make_dataset("yelp2018")
train = [[f"u{u}", f"i{i}", 1.0] for u, i in zip(d["train_u"].tolist(), d["train_i"].tolist())]
test = [[f"u{u}", f"i{i}", 1.0] for u, i in zip(d["test_u"].tolist(), d["test_i"].tolist())]
conf = conf_from_text("ratings=./x.txt\nmodel.name=BPR\nratings.setup=-columns 0 1 2\nevaluation.setup=-testSet x\nitem.ranking=on -topN 20\n"
                      "num.factors=64\nnum.max.epoch=1\nlearnRate=-init 0.05 -max 1\nreg.lambda=-u 0.01 -i 0.01 -b 0.2 -s 0.2\noutput.setup=off -dir ./results/")
out = {}
with redirect_stdout(io.StringIO()):
    m = BPR(conf, train, test); m.readConfiguration(); m.initModel()
    m.P = m.P.astype(np.float32); m.Q = m.Q.astype(np.float32)          # TF-path models rank fp32 tables
    m.rank_measure_all_test_users([20], 20)                              # warm-up: ranker, test CSR
    m.rank_measure_all_test_users([20], 20)
    t0 = time.perf_counter()
    for _ in range(5): fast = m.rank_measure_all_test_users([20], 20)
    out["device_hits_s"] = (time.perf_counter() - t0) / 5                # steady state (per-epoch evaluation)
    t0 = time.perf_counter(); rec = m.rank_all_test_users(20); t1 = time.perf_counter()
    slow = Measure.rankingMeasure(m.data.testSet_u, rec, [20]); t2 = time.perf_counter()
out["host_lists_s"], out["host_measure_s"] = t1 - t0, t2 - t1
out["identical_strings"] = fast == slow
out["test_users"] = len(m.data.testSet_u)
print(json.dumps(out))
