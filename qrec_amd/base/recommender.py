"""The Recommender template-method API -- the boundary the hot path sits behind.

Same lifecycle, hook names, attributes, printed/logged/written artefacts as the
reference's ``Recommender`` (base/recommender.py:14-212): ``execute()`` =
readConfiguration -> initializing_log -> printAlgorConfig -> initModel ->
trainModel | trainModel_tf -> evalRanking | evalRatings [-> saveModel], returning
``self.measure`` (a list of ``"Name:value\\n"`` strings).

No device work happens in ``__init__``: QRec constructs models in the parent process and
forks one child per CV fold (QRec.py:76-89); the HIP context is created lazily by the
first kernel call inside initModel/trainModel.
"""
from __future__ import annotations

import sys
from os.path import abspath
from time import localtime, strftime, time

import numpy as np

from ..data.rating import Rating
from ..util.config import OptionConf
from ..util.io import FileIO
from ..util.log import Log
from ..util.measure import Measure
from ..util.qmath import find_k_largest


class Recommender:
    def __init__(self, conf, trainingSet, testSet, fold="[1]"):
        self.config = conf
        self.isSaveModel = False
        self.isLoadModel = False
        self.isOutput = True
        self.ranking = None
        self.output = None
        self.data = Rating(conf, trainingSet, testSet)
        self.foldInfo = fold
        self.evalSettings = OptionConf(conf["evaluation.setup"])
        self.measure = []
        self.recOutput = []
        self.num_users, self.num_items, self.train_size = self.data.trainingSize()

    # ---- configuration / logging ----------------------------------------------------------
    def initializing_log(self):
        stamp = strftime("%Y-%m-%d %H-%M-%S", localtime(time()))
        self.log = Log(self.modelName, self.modelName + self.foldInfo + " " + stamp)
        self.log.add("### model configuration ###")
        for key in self.config.config:
            self.log.add(key + "=" + self.config[key])

    def readConfiguration(self):
        self.modelName = self.config["model.name"]
        self.output = OptionConf(self.config["output.setup"])
        self.isOutput = self.output.isMainOn()
        self.ranking = OptionConf(self.config["item.ranking"])

    def printAlgorConfig(self):
        print("Model:", self.config["model.name"])
        print("Ratings dataset:", abspath(self.config["ratings"]))
        if self.evalSettings.contains("-testSet"):
            print("Test set:", abspath(self.evalSettings["-testSet"]))
        print("Training set size: (user count: %d, item count %d, record count: %d)" % self.data.trainingSize())
        print("Test set size: (user count: %d, item count %d, record count: %d)" % self.data.testSize())
        print("=" * 80)
        name = self.config["model.name"]
        if self.config.contains(name):
            args = OptionConf(self.config[name])
            print("Specific parameters:", "".join(k[1:] + ":" + args[k] + "  " for k in args.keys()))
            print("=" * 80)

    # ---- hooks ----------------------------------------------------------------------------------
    def initModel(self):
        pass

    def trainModel(self):
        pass

    def trainModel_tf(self):
        """The reference's TensorFlow variant.  There is no TensorFlow here; a model that
        has a device implementation of its TF graph overrides this, otherwise the base
        behaves like the reference on a box without TF (ImportError -> trainModel)."""
        raise ImportError("no trainModel_tf for " + type(self).__name__)

    def saveModel(self):
        pass

    def loadModel(self):
        pass

    def predictForRating(self, u, i):
        pass

    def predictForRanking(self, u):
        pass

    def checkRatingBoundary(self, prediction):
        lo, hi = self.data.rScale[0], self.data.rScale[-1]
        if prediction > hi:
            return hi
        if prediction < lo:
            return lo
        return round(prediction, 3)

    # ---- evaluation -------------------------------------------------------------------------------
    def evalRatings(self):
        lines = ["userId  itemId  original  prediction\n"]
        for pos, (user, item, rating) in enumerate(self.data.testData):
            pred = self.checkRatingBoundary(self.predictForRating(user, item))
            self.data.testData[pos].append(pred)
            lines.append(user + " " + item + " " + str(rating) + " " + str(pred) + "\n")
        stamp = strftime("%Y-%m-%d %H-%M-%S", localtime(time()))
        outDir = self.output["-dir"]
        if self.isOutput:
            FileIO.writeFile(outDir, self.config["model.name"] + "@" + stamp + "-rating-predictions" + self.foldInfo + ".txt", lines)
            print("The result has been output to ", abspath(outDir), ".")
        self.measure = Measure.ratingMeasure(self.data.testData)
        FileIO.writeFile(outDir, self.config["model.name"] + "@" + stamp + "-measure" + self.foldInfo + ".txt", self.measure)
        self.log.add("###Evaluation Results###")
        self.log.add(self.measure)
        print("The result of %s %s:\n%s" % (self.modelName, self.foldInfo, "".join(self.measure)))

    def _top_n_setting(self):
        if not self.ranking.contains("-topN"):
            print("No correct evaluation metric is specified!")
            sys.exit(-1)
        top = [int(x) for x in self.ranking["-topN"].split(",")]
        N = max(top)
        if N > 100 or N < 1:
            print("N can not be larger than 100! It has been reassigned to 10")
            N = 10
        return top, N

    def rank_all_test_users(self, N):
        """recList[user] = [(itemName, score)] * N for every user of testSet_u, by the
        reference's rule: score all items, set the user's rated train items to 0 (not
        -inf, base/recommender.py:147-149), keep the N largest (util/qmath.py:134-146).
        Models with device-resident embeddings override this with the batched kernel."""
        recList = {}
        total = len(self.data.testSet_u)
        for pos, user in enumerate(self.data.testSet_u):
            candidates = self.predictForRanking(user)
            rated, _ = self.data.userRated(user)
            for item in rated:
                candidates[self.data.item[item]] = 0
            ids, scores = find_k_largest(N, candidates)
            recList[user] = [(self.data.id2item[iid], score) for iid, score in zip(ids, scores)]
            if pos % 100 == 0:
                print(self.modelName, self.foldInfo, "progress:" + str(pos) + "/" + str(total))
        return recList

    def rank_measure_all_test_users(self, top, N):
        """Measure strings for all test users without materialising the lists, or None when the model has no
        device-resident tables (the generic host loop then runs)."""
        return None

    def _publish_measure(self, stamp):
        self.log.add("###Evaluation Results###")
        self.log.add(self.measure)
        FileIO.writeFile(self.output["-dir"], self.config["model.name"] + "@" + stamp + "-measure" + self.foldInfo + ".txt", self.measure)
        print("The result of %s %s:\n%s" % (self.modelName, self.foldInfo, "".join(self.measure)))

    def evalRanking(self):
        top, N = self._top_n_setting()
        if not self.isOutput and not self.evalSettings.contains("-predict"):
            # nobody reads the lists (no result file, no -predict): hits and DCG sums come straight from the
            # device-resident top-N lists, the strings are the reference's to the last digit
            fast = self.rank_measure_all_test_users(top, N)
            if fast is not None:
                self.measure = fast
                self._publish_measure(strftime("%Y-%m-%d %H-%M-%S", localtime(time())))
                return
        self.recOutput.append("userId: recommendations in (itemId, ranking score) pairs, * means the item matches.\n")
        recList = self.rank_all_test_users(N)
        for user, recs in recList.items():
            truth = self.data.testSet_u[user]
            self.recOutput.append(user + ":" + "".join(
                " (" + item + "," + str(score) + ")" + ("*" if item in truth else "") for item, score in recs) + "\n")
        stamp = strftime("%Y-%m-%d %H-%M-%S", localtime(time()))
        outDir = self.output["-dir"]
        if self.isOutput:
            FileIO.writeFile(outDir, self.config["model.name"] + "@" + stamp + "-top-" + str(N) + "items" + self.foldInfo + ".txt", self.recOutput)
            print("The result has been output to ", abspath(outDir), ".")
        if self.evalSettings.contains("-predict"):
            sys.exit(0)
        self.measure = Measure.rankingMeasure(self.data.testSet_u, recList, top)
        self._publish_measure(stamp)

    # ---- template method ------------------------------------------------------------------------------
    def execute(self):
        self.readConfiguration()
        self.initializing_log()
        if self.foldInfo == "[1]":
            self.printAlgorConfig()
        if self.isLoadModel:
            print("Loading model %s..." % self.foldInfo)
            self.loadModel()
        else:
            print("Initializing model %s..." % self.foldInfo)
            self.initModel()
            print("Building Model %s..." % self.foldInfo)
            try:
                if self.evalSettings.contains("-tf"):
                    self.trainModel_tf()
                else:
                    self.trainModel()
            except ImportError:
                self.trainModel()
        print("Predicting %s..." % self.foldInfo)
        if self.ranking.isMainOn():
            self.evalRanking()
        else:
            self.evalRatings()
        if self.isSaveModel:
            print("Saving model %s..." % self.foldInfo)
            self.saveModel()
        return self.measure
