"""CPU oracle for QRec's embedding-training hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; it is the checker, never the product (see qrec_oracle.c header).

``oracle.c`` : ctypes binding of ``libqrec_oracle.so`` (plain-C restatement; built by
               ``make -C oracle``).
``oracle.tfmodels`` : numpy/scipy restatement of the TF-1.14 graph models (LightGCN, ...).
"""
from . import c  # noqa: F401
