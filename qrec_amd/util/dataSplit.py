"""Hold-out and k-fold splitting with the reference's random-number consumption
(util/dataSplit.py:9-44): ``dataSplit`` draws one ``random()`` per row, in row order."""
from __future__ import annotations

import random as _random
from random import random

import numpy as np


def _compact(data):
    from ..data.rows import RatingRows
    return data if isinstance(data, RatingRows) else None


class DataSplit:
    @staticmethod
    def dataSplit(data, test_ratio: float = 0.3, output: bool = False, path: str = "./",
                  order: int = 1, binarized: bool = False):
        if not 0 < test_ratio < 1:
            test_ratio = 0.3
        rows = _compact(data)
        if rows is not None and not output:
            # the same len(data) draws of random(), replayed natively; the Python generator moves in lock-step
            from .. import capi
            state = _random.getstate()
            words = capi.state_from_python(state)
            to_test = capi.mt_data_split(words, len(rows), test_ratio)
            _random.setstate(capi.state_to_python(words, state[2]))
            keep_test = to_test & (rows.rating != 0) if binarized else to_test
            return rows.take(~to_test), rows.take(keep_test)
        train, test = [], []
        for row in data:
            if random() < test_ratio:
                if not binarized or row[2]:
                    test.append(row)
            else:
                train.append(row)
        if output:
            from .io import FileIO
            FileIO.writeFile(path, "testSet[" + str(order) + "]", test)
            FileIO.writeFile(path, "trainingSet[" + str(order) + "]", train)
        return train, test

    @staticmethod
    def crossValidation(data, k: int, output: bool = False, path: str = "./", order: int = 1,
                        binarized: bool = False):
        if k <= 1 or k > 10:
            k = 3
        rows = _compact(data)
        if rows is not None and not output:
            pos = np.arange(len(rows)) % k
            for fold in range(k):
                in_test = pos == fold
                yield rows.take(~in_test), rows.take(in_test & (rows.rating != 0) if binarized else in_test)
            return
        for fold in range(k):
            train = [row[:] for pos, row in enumerate(data) if pos % k != fold]
            test = [row[:] for pos, row in enumerate(data)
                    if pos % k == fold and (not binarized or row[2])]
            yield train, test
