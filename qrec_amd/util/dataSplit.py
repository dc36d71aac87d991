"""Hold-out and k-fold splitting with the reference's random-number consumption
(util/dataSplit.py:9-44): ``dataSplit`` draws one ``random()`` per row, in row order."""
from __future__ import annotations

from random import random


class DataSplit:
    @staticmethod
    def dataSplit(data, test_ratio: float = 0.3, output: bool = False, path: str = "./",
                  order: int = 1, binarized: bool = False):
        if not 0 < test_ratio < 1:
            test_ratio = 0.3
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
        for fold in range(k):
            train = [row[:] for pos, row in enumerate(data) if pos % k != fold]
            test = [row[:] for pos, row in enumerate(data)
                    if pos % k == fold and (not binarized or row[2])]
            yield train, test
