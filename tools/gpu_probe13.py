#!/usr/bin/env python3
"""cProfile of one steady-state rank_measure_all_test_users call at the Yelp shape (where do the 14 ms go?)."""
import cProfile, io, os, pstats, sys
from contextlib import redirect_stdout
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import conf_from_text
from qrec_amd import capi
from qrec_amd.model.ranking.BPR import BPR
from qrec_amd.synth import make_dataset
capi.init(0)
d = make_dataset("yelp2018")
train = [[f"u{u}", f"i{i}", 1.0] for u, i in zip(d["train_u"].tolist(), d["train_i"].tolist())]
test = [[f"u{u}", f"i{i}", 1.0] for u, i in zip(d["test_u"].tolist(), d["test_i"].tolist())]
conf = conf_from_text("ratings=./x.txt\nmodel.name=BPR\nratings.setup=-columns 0 1 2\nevaluation.setup=-testSet x\nitem.ranking=on -topN 20\n"
                      "num.factors=64\nnum.max.epoch=1\nlearnRate=-init 0.05 -max 1\nreg.lambda=-u 0.01 -i 0.01 -b 0.2 -s 0.2\noutput.setup=off -dir ./results/")
with redirect_stdout(io.StringIO()):
    m = BPR(conf, train, test); m.readConfiguration(); m.initModel()
m.P = m.P.astype(np.float32); m.Q = m.Q.astype(np.float32)
m.rank_measure_all_test_users([20], 20); m.rank_measure_all_test_users([20], 20)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): m.rank_measure_all_test_users([20], 20)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumtime").print_stats(18)
