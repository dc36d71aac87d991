"""``SocialRecommender`` (base/socialRecommender.py:5-54): an iterative recommender with the ``social`` file's
relations, restricted to users that occur in the training data."""
from __future__ import annotations

from os.path import abspath

from ..data.social import Social
from ..util import config
from .iterativeRecommender import IterativeRecommender


class SocialRecommender(IterativeRecommender):
    def __init__(self, conf, trainingSet, testSet, relation, fold="[1]"):
        super().__init__(conf, trainingSet, testSet, fold)
        self.social = Social(self.config, relation)
        known = self.data.user
        # relations whose two ends are both training users survive, in file order (base/socialRecommender.py:10-41);
        # the list is filtered in place like the reference's ``del`` loop: callers share it across CV folds
        for book in (self.social.followees, self.social.followers):
            for user in [u for u in book if u not in known]:
                del book[user]
            for user in book:
                for other in [o for o in book[user] if o not in known]:
                    del book[user][other]
        self.social.relation[:] = [p for p in self.social.relation if p[0] in known and p[1] in known]

    def readConfiguration(self):
        super().readConfiguration()
        self.regS = float(config.OptionConf(self.config["reg.lambda"])["-s"])

    def printAlgorConfig(self):
        super().printAlgorConfig()
        print("Social dataset:", abspath(self.config["social"]))
        print("Social relation size ", "(User count:", len(self.social.user), "Relation count:" + str(len(self.social.relation)) + ")")
        print("Social Regularization parameter: regS %.3f" % self.regS)
        print("=" * 80)

    def relation_ids(self):
        """(follower, followee) training-user indices of the kept relations, in list order"""
        import numpy as np
        user = self.data.user
        rel = self.social.relation
        return (np.fromiter((user[p[0]] for p in rel), dtype=np.int32, count=len(rel)),
                np.fromiter((user[p[1]] for p in rel), dtype=np.int32, count=len(rel)))
