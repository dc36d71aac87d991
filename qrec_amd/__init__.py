
import sys as _sys

# Bit-parity with the reference rests on CPython <= 3.11 behaviour that the native replays and the measure strings
# hard-code: random.sample's pool / selection-set switch (qrec_amd/csrc/mt_sampler.cpp) and sum()'s plain left-to-right
# float addition (3.12 made it compensated; qrec_amd/ranking.py emulates the plain one).  The reference itself pins 3.x
# era packages (README.md:47-58).  Refuse to run silently different -- but only where bit parity is claimed: the exact-mode
# samplers and the measure strings call ``require_reference_python()``; the throughput mode, the graph trainers, bench.py
# and the tools do not depend on either and only get a warning.
PYTHON_MATCHES_REFERENCE = _sys.version_info < (3, 12)


def require_reference_python(what: str):
    if not PYTHON_MATCHES_REFERENCE and not __import__("os").environ.get("QREC_ALLOW_NEW_PYTHON"):
        raise RuntimeError("%s reproduces CPython <= 3.11 `random` / `sum` semantics bit for bit; Python %d.%d changes them "
                           "(qrec_amd/__init__.py; QREC_ALLOW_NEW_PYTHON=1 runs anyway, without the bit-parity claim)"
                           % ((what,) + tuple(_sys.version_info[:2])))


if not PYTHON_MATCHES_REFERENCE:
    import warnings as _warnings
    _warnings.warn("qrec_amd: Python %d.%d -- the exact-mode samplers and the measure strings are bit-exact on CPython <= 3.11 "
                   "only and will refuse to run; throughput mode and the trainers are unaffected" % _sys.version_info[:2])
