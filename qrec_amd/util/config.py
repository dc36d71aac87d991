"""Conf-file and option-string parsing with the semantics of the reference's
util/config.py (ModelConf :2-27, OptionConf :29-66), so stock ``*.conf`` files run
unchanged.  Errors follow the reference: print a message and ``exit(-1)``.
"""
from __future__ import annotations

import os
import sys


def _die(msg: str):
    print(msg)
    sys.exit(-1)


class ModelConf:
    """``key=value`` per line (util/config.py:16-27)."""

    def __init__(self, fileName: str | None = None, **overrides):
        self.config: dict[str, str] = {}
        if fileName is not None:
            self.readConfiguration(fileName)
        self.config.update({k.replace("__", "."): str(v) for k, v in overrides.items()})

    @classmethod
    def from_dict(cls, d: dict) -> "ModelConf":
        c = cls()
        c.config = {str(k): str(v) for k, v in d.items()}
        return c

    def __getitem__(self, key: str) -> str:
        if key not in self.config:
            _die("parameter " + key + " is invalid!")
        return self.config[key]

    def __setitem__(self, key: str, value) -> None:
        self.config[key] = str(value)

    def contains(self, key: str) -> bool:
        return key in self.config

    def readConfiguration(self, file: str) -> None:
        if not os.path.exists(file):
            print("config file is not found!")
            raise IOError(file)
        with open(file) as fh:
            for lineno, raw in enumerate(fh):
                line = raw.strip()
                if not line:
                    continue
                parts = line.split("=")
                if len(parts) != 2:  # the reference unpacks exactly two fields
                    print("config file is not in the correct format! Error Line:%d" % lineno)
                    continue
                self.config[parts[0]] = parts[1]


def _is_flag(tok: str) -> bool:
    # "-topN" is a flag, "-1" / "-0.5"-style negatives are values (util/config.py:39)
    return tok.startswith("-") and not tok[1:].isdigit()


class OptionConf:
    """``"on -topN 10,20 -dir ./results/"``-style option strings.

    The first token ``on``/``off`` sets the main switch; every flag maps to the tokens
    that follow it up to the next flag, joined by single spaces (a flag with nothing
    after it maps to the empty string).
    """

    def __init__(self, content: str):
        self.line = content.strip().split(" ")
        self.options: dict[str, str] = {}
        self.mainOption = self.line[0] == "on"
        n = len(self.line)
        for pos, tok in enumerate(self.line):
            if not _is_flag(tok):
                continue
            end = pos + 1
            while end < n and not _is_flag(self.line[end]):
                end += 1
            self.options[tok] = " ".join(self.line[pos + 1:end])

    def __getitem__(self, key: str):
        if key not in self.options:
            _die("parameter " + key + " is invalid!")
        return self.options[key]

    def keys(self):
        return self.options.keys()

    def isMainOn(self) -> bool:
        return self.mainOption

    def contains(self, key: str) -> bool:
        return key in self.options
