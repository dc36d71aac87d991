#!/usr/bin/env python3
"""Development aid: the HIP SimGCL trainer's tables after the golden run under the recorded sign pattern, written to
gpurun_out/simgcl_parity_tables.npz for analysis on the build host (which coordinates carry the distance to the reference's run)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_tf_golden as T          # noqa: E402
from qrec_amd import capi               # noqa: E402
from qrec_amd.graph import ordered_reductions  # noqa: E402

capi.init(0)
with ordered_reductions():
    out = T._run_simgcl_recorded()
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "simgcl_parity_tables.npz"), **out)
print({k: v.shape for k, v in out.items()})
