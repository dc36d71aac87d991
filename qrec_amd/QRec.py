"""Run orchestrator with the reference's conf-driven behaviour (QRec.py:7-118): load the
ratings, pick the evaluation protocol (``-testSet`` / ``-ap`` / ``-cv`` / ``-predict``,
``-b`` binarize), resolve the model class by ``model.name`` and execute it; k-fold CV runs
one process per fold."""
from __future__ import annotations

import importlib
import os
import sys
from multiprocessing import Manager, Process
from time import localtime, strftime, time

from .util.config import OptionConf
from .util.dataSplit import DataSplit
from .util.io import FileIO


def resolve_model(name: str):
    """``from model.rating.<N> import <N>``, else ``model.ranking`` (QRec.py:51-56)."""
    for pkg in ("rating", "ranking"):
        try:
            return getattr(importlib.import_module(f"qrec_amd.model.{pkg}.{name}"), name)
        except ModuleNotFoundError:
            continue
    raise ImportError(f"model {name} is not provided by qrec_amd (hot-path models: BPR, TBPR, SBPR, BasicMF, PMF, SVD, SVDPlusPlus, EE, LightGCN, NGCF, SimGCL, SGL, BUIR, SEPT, MHCN)")


def _run_fold(results, model, order, spread=False):
    # one process <-> one device: with ``-cv k -p`` the folds run side by side (QRec.py:76-89), so they are dealt
    # round-robin over the node's GPUs unless the user pinned QREC_DEVICE
    if spread and "QREC_DEVICE" not in os.environ:
        from . import capi
        os.environ["QREC_DEVICE"] = str((order - 1) % max(capi.device_count(), 1))
    results[order] = model.execute()


class QRec:
    def __init__(self, config):
        self.trainingData, self.testData, self.relation, self.measure = [], [], [], []
        self.config = config
        self.ratingConfig = OptionConf(config["ratings.setup"])
        if not config.contains("evaluation.setup"):
            print("Wrong configuration of evaluation!")
            sys.exit(-1)
        ev = self.evaluation = OptionConf(config["evaluation.setup"])
        binarized = ev.contains("-b")
        bottom = float(ev["-b"]) if binarized else 0
        load = lambda path, **kw: FileIO.loadDataSet(config, path, binarized=binarized, threshold=bottom, **kw)
        if ev.contains("-testSet"):
            self.trainingData = load(config["ratings"])
            self.testData = load(ev["-testSet"], bTest=True)
        elif ev.contains("-ap"):
            self.trainingData = load(config["ratings"])
            self.trainingData, self.testData = DataSplit.dataSplit(
                self.trainingData, test_ratio=float(ev["-ap"]), binarized=binarized)
        elif ev.contains("-cv"):
            self.trainingData = load(config["ratings"])
        elif ev.contains("-predict"):
            self.trainingData = load(config["ratings"])
            self.testData = FileIO.loadUserList(ev["-predict"])
        if config.contains("social"):                      # QRec.py:44-46
            self.socialConfig = OptionConf(config["social.setup"])
            self.relation = FileIO.loadRelationship(config, config["social"])
        print("Reading data and preprocessing...")

    def execute(self):
        cls = resolve_model(self.config["model.name"])
        ev = self.evaluation
        social = self.config.contains("social")            # social models take the relation list (QRec.py:72-75,110-113)
        if not ev.contains("-cv"):
            if social:
                return cls(self.config, self.trainingData, self.testData, self.relation).execute()
            return cls(self.config, self.trainingData, self.testData).execute()
        k = int(ev["-cv"])
        if k < 2 or k > 10:
            print("k for cross-validation should not be greater than 10 or less than 2")
            sys.exit(-1)
        binarized = ev.contains("-b")
        folds = enumerate(DataSplit.crossValidation(self.trainingData, k, binarized=binarized), 1)
        build = lambda train, test, fold: (cls(self.config, train, test, self.relation, fold) if social
                                           else cls(self.config, train, test, fold))
        from .dist import BatchParallel
        if BatchParallel.from_env() is not None:
            # one process per GPU (torch.distributed.run): this process already holds a device context and an RCCL
            # communicator, neither of which survives a fork, and every rank must issue the same collectives in the
            # same order -- so the folds run one after another IN this process, each of them data-parallel over the
            # ranks (``-p`` has nothing left to spread: all GPUs work on the current fold).
            results = {order: build(train, test, "[" + str(order) + "]").execute() for order, (train, test) in folds}
        else:
            results = Manager().dict()
            tasks = []
            for order, (train, test) in folds:
                model = build(train, test, "[" + str(order) + "]")     # built in the parent, run in the child
                tasks.append(Process(target=_run_fold, args=(results, model, order, ev.contains("-p"))))
            for p in tasks:
                p.start()
                if not ev.contains("-p"):
                    p.join()
            for p in tasks:
                p.join()
        self.measure = [dict(results)[f] for f in range(1, k + 1)]
        res = []
        for pos, line in enumerate(self.measure[0]):
            if line[:3] == "Top":
                res.append(line)
                continue
            name = line.split(":")[0]
            res.append(name + ":" + str(sum(float(self.measure[f][pos].split(":")[1]) for f in range(k)) / k) + "\n")
        stamp = strftime("%Y-%m-%d %H-%M-%S", localtime(time()))
        outDir = OptionConf(self.config["output.setup"])["-dir"]
        FileIO.writeFile(outDir, self.config["model.name"] + "@" + stamp + "-" + str(k) + "-fold-cv" + ".txt", res)
        print("The result of %d-fold cross validation:\n%s" % (k, "".join(res)))
        return res
