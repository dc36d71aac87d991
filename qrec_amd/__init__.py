
import sys as _sys

# Bit-parity with the reference rests on CPython <= 3.11 behaviour that one native replay hard-codes: random.sample's pool /
# selection-set switch (qrec_amd/csrc/mt_sampler.cpp qrec_mt_sample_range: the -ap data split and SEPT's samples).  choice / shuffle /
# _randbelow, which the BPR and pairwise samplers replay, have not changed; sum()'s plain left-to-right float addition (3.12 made it
# compensated) is EMULATED with numpy in qrec_amd/ranking.py, so the measure strings do not depend on the interpreter either.  The
# reference itself pins 3.x era packages (README.md:47-58).  Refuse to run silently different -- but only where it matters:
# ``qrec_mt_sample_range`` calls ``require_reference_python()``; everything else runs on any interpreter.
PYTHON_MATCHES_REFERENCE = _sys.version_info < (3, 12)


def require_reference_python(what: str):
    if not PYTHON_MATCHES_REFERENCE and not __import__("os").environ.get("QREC_ALLOW_NEW_PYTHON"):
        raise RuntimeError("%s reproduces CPython <= 3.11 `random` / `sum` semantics bit for bit; Python %d.%d changes them "
                           "(qrec_amd/__init__.py; QREC_ALLOW_NEW_PYTHON=1 runs anyway, without the bit-parity claim)"
                           % ((what,) + tuple(_sys.version_info[:2])))


if not PYTHON_MATCHES_REFERENCE:
    import warnings as _warnings
    _warnings.warn("qrec_amd: Python %d.%d -- the replay of random.sample (the -ap split, SEPT's samples) is bit-exact on CPython <= 3.11 "
                   "only and will refuse to run; everything else is unaffected" % _sys.version_info[:2])
