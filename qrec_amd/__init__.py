
import sys as _sys

# Bit-parity with the reference rests on CPython <= 3.11 behaviour that the native replays and the measure strings
# hard-code: random.sample's pool / selection-set switch (qrec_amd/csrc/mt_sampler.cpp) and sum()'s plain left-to-right
# float addition (3.12 made it compensated; qrec_amd/ranking.py emulates the plain one).  The reference itself pins 3.x
# era packages (README.md:47-58).  Refuse to run silently different.
if _sys.version_info >= (3, 12):
    raise ImportError("qrec_amd reproduces CPython <= 3.11 `random` / `sum` semantics bit for bit; Python %d.%d changes "
                      "them (see qrec_amd/__init__.py)" % _sys.version_info[:2])
