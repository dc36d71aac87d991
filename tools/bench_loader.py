"""Host-side data path at the Yelp2018 shape: rating files -> rows -> data model -> CSR views, through the native
loader (qrec_ratings_load + array-backed Rating) and through the reference-style Python loop (QREC_NATIVE_LOADER=0).
CPU only.  usage: python tools/bench_loader.py [--python-too]"""
import contextlib
import io
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from qrec_amd.data.rating import Rating  # noqa: E402
from qrec_amd.synth import make_dataset, write_rating_file  # noqa: E402
from qrec_amd.util.config import ModelConf  # noqa: E402
from qrec_amd.util.io import FileIO  # noqa: E402


def once(conf, train_path, test_path):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        train = FileIO.loadDataSet(conf, train_path, binarized=True, threshold=1.0)
        test = FileIO.loadDataSet(conf, test_path, bTest=True, binarized=True, threshold=1.0)
        t1 = time.perf_counter()
        data = Rating(conf, train, test)
        t2 = time.perf_counter()
        uid, iid, r = data.training_arrays()
        pos, rated = data.positive_csr(), data.rated_csr()          # what the sampler and the evaluation mask read
        t3 = time.perf_counter()
    return dict(files_s=t1 - t0, data_model_s=t2 - t1, views_s=t3 - t2, total_s=t3 - t0, rows=int(np.asarray(uid).size))


def main():
    d = make_dataset("yelp2018")
    with tempfile.TemporaryDirectory() as tmp:
        tr, te = os.path.join(tmp, "train.txt"), os.path.join(tmp, "test.txt")
        write_rating_file(tr, d["train_u"], d["train_i"]); write_rating_file(te, d["test_u"], d["test_i"])
        cf = os.path.join(tmp, "y.conf")
        open(cf, "w").write(f"ratings={tr}\nratings.setup=-columns 0 1 2\nmodel.name=BPR\nevaluation.setup=-testSet {te} -b 1\n"
                            "item.ranking=on -topN 20\nnum.factors=64\nnum.max.epoch=1\nbatch_size=2048\nlearnRate=-init 0.01 -max 1\n"
                            "reg.lambda=-u 0.001 -i 0.001 -b 0.2 -s 0.2\noutput.setup=off -dir ./results/\n")
        conf = ModelConf(cf)
        runs = [once(conf, tr, te) for _ in range(4)]
        best = min(runs, key=lambda x: x["total_s"])
        print("native:", {k: round(v, 3) if isinstance(v, float) else v for k, v in best.items()}, "all totals", [round(x["total_s"], 3) for x in runs])
        if "--python-too" in sys.argv:
            os.environ["QREC_NATIVE_LOADER"] = "0"
            p = once(conf, tr, te)
            print("python:", {k: round(v, 3) if isinstance(v, float) else v for k, v in p.items()})


if __name__ == "__main__":
    main()
