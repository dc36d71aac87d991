"""Evaluation measures with the reference's definitions (util/measure.py:15-141):
Precision/Recall/F1/NDCG@N over recommendation lists, MAE/RMSE over rating predictions.
Results are ``"Name:value\\n"`` strings because downstream code parses them
(QRec.py:95-101, base/iterativeRecommender.py:135-137)."""
from __future__ import annotations

import math
import sys


class Measure:
    @staticmethod
    def hits(origin, res):
        return {user: len(set(origin[user]).intersection(item for item, _ in res[user]))
                for user in origin}

    @staticmethod
    def precision(hits, N):
        return sum(hits.values()) / (len(hits) * N)

    @staticmethod
    def recall(hits, origin):
        per_user = [hits[user] / len(origin[user]) for user in hits]
        return sum(per_user) / len(per_user)

    @staticmethod
    def F1(prec, recall):
        return 2 * prec * recall / (prec + recall) if (prec + recall) != 0 else 0

    @staticmethod
    def NDCG(origin, res, N):
        # natural-log discounts 1/ln(rank+1), ideal DCG over min(|test items|, N) slots
        total = 0
        for user, recs in res.items():
            truth = origin[user]
            dcg = sum(1.0 / math.log(pos + 2) for pos, (item, _) in enumerate(recs) if item in truth)
            idcg = sum(1.0 / math.log(pos + 2) for pos in range(min(len(truth), N)))
            total += dcg / idcg
        return total / len(res)

    @staticmethod
    def rankingMeasure(origin, res, N):
        out = []
        for n in N:
            cut = {user: recs[:n] for user, recs in res.items()}
            if len(origin) != len(cut):
                print("The Lengths of test set and predicted set are not match!")
                sys.exit(-1)
            hits = Measure.hits(origin, cut)
            prec = Measure.precision(hits, n)
            rec = Measure.recall(hits, origin)
            out.append("Top " + str(n) + "\n")
            out.append("Precision:" + str(prec) + "\n")
            out.append("Recall:" + str(rec) + "\n")
            out.append("F1:" + str(Measure.F1(prec, rec)) + "\n")
            out.append("NDCG:" + str(Measure.NDCG(origin, cut, n)) + "\n")
        return out

    @staticmethod
    def MAE(res):
        err = [abs(entry[2] - entry[3]) for entry in res]
        return sum(err) / len(err) if err else 0

    @staticmethod
    def RMSE(res):
        err = [(entry[2] - entry[3]) ** 2 for entry in res]
        return math.sqrt(sum(err) / len(err)) if err else 0

    @staticmethod
    def ratingMeasure(res):
        return ["MAE:" + str(Measure.MAE(res)) + "\n", "RMSE:" + str(Measure.RMSE(res)) + "\n"]
