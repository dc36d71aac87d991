"""Dataset text I/O with the reference's on-disk format and options (util/io.py:31-76):
``user item rating`` rows split on space / comma / tab (or ``-delim``), ``-columns`` picks
the fields, ``-header`` skips the first line, and ``-b t`` (binarize) drops rows whose
rating is below ``t`` and sets the rest to 1.
"""
from __future__ import annotations

import os
import re
import sys

from .config import OptionConf


class FileIO:
    @staticmethod
    def writeFile(dir: str, file: str, content, op: str = "w") -> None:
        from ..dist import is_output_rank
        if not is_output_rank():       # multi-GPU run: the ranks hold identical results, rank 0 writes them
            return
        os.makedirs(dir, exist_ok=True)
        with open(dir + file, op) as fh:
            fh.writelines(content)

    @staticmethod
    def deleteFile(filePath: str) -> None:
        if os.path.exists(filePath):
            os.remove(filePath)

    @staticmethod
    def loadDataSet(conf, file: str, bTest: bool = False, binarized: bool = False,
                    threshold: float = 3.0):
        setup = OptionConf(conf["ratings.setup"])
        print("loading test data..." if bTest else "loading training data...")
        rows = FileIO._load_native(setup, file, bTest, binarized, threshold)
        if rows is not None:
            return rows
        with open(file) as fh:
            lines = fh.readlines()
        if setup.contains("-header"):
            lines = lines[1:]
        cols = [int(c) for c in setup["-columns"].strip().split()]
        splitter = re.compile(setup["-delim"] if setup.contains("-delim") else " |,|\t")
        if not bTest and len(cols) < 2:
            print("The rating file is not in a correct format. Error: Line num %d" % 0)
            sys.exit(-1)
        has_rating = len(cols) >= 3
        rows = []
        for line in lines:
            fields = splitter.split(line.strip())
            try:
                user, item = fields[cols[0]], fields[cols[1]]
                rating = fields[cols[2]] if has_rating else 1
                if binarized:
                    if float(fields[cols[2]]) < threshold:
                        continue
                    rating = 1
                rows.append([user, item, float(rating)])
            except ValueError:
                print("Error! Have you added the option -header to the rating.setup?")
                sys.exit(-1)
        return rows

    @staticmethod
    def loadRelationship(conf, filePath: str):
        """``follower followee [weight]`` rows of the ``social`` file (util/io.py:88-111): same splitting, ``-columns``
        and ``-header`` handling as the ratings; weight 1 when no third column is configured."""
        setup = OptionConf(conf["social.setup"])
        print("loading social data...")
        with open(filePath) as fh:
            lines = fh.readlines()
        if setup.contains("-header"):
            lines = lines[1:]
        order = setup["-columns"].strip().split()
        splitter = re.compile(" |,|\t")
        relation = []
        for lineNo, line in enumerate(lines):
            fields = splitter.split(line.strip())
            if len(order) < 2:
                print("The social file is not in a correct format. Error: Line num %d" % lineNo)
                sys.exit(-1)
            weight = 1 if len(order) < 3 else float(fields[int(order[2])])
            relation.append([fields[int(order[0])], fields[int(order[1])], weight])
        return relation

    @staticmethod
    def _load_native(setup, file, bTest, binarized, threshold):
        """The same rows through libqrec_hip's parser (qrec_ratings_load), as a ``RatingRows``; None whenever the
        conf or the file asks for something only the Python path reproduces exactly (regex ``-delim``, non-ASCII
        text, unusual float literals, malformed records, a missing file): the loop below then runs -- and fails --
        as the reference's does."""
        import os
        if os.environ.get("QREC_NATIVE_LOADER", "1") == "0" or not os.path.isfile(file):
            return None
        delims = None
        if setup.contains("-delim"):
            parts = setup["-delim"].split("|")
            if not parts or any(len(p) != 1 or p in "\\^$.*+?()[]{}" for p in parts):
                return None                      # a real regular expression
            delims = "".join(parts)
        try:
            cols = [int(c) for c in setup["-columns"].strip().split()]
        except ValueError:
            return None
        if len(cols) < 2 or any(c < 0 for c in cols):
            return None
        from .. import capi
        from ..data.rows import RatingRows
        got = capi.ratings_load(file, delims, cols[0], cols[1], cols[2] if len(cols) >= 3 else -1,
                                setup.contains("-header"), binarized, threshold)
        return None if got is None else RatingRows(*got)

    @staticmethod
    def loadUserList(filepath: str):
        print("loading user List...")
        with open(filepath) as fh:
            return [line.strip().split()[0] for line in fh]
