"""Bridge to the real scikit-image Lewiner kernel (build container only).

scikit-image is only installed for /opt/conda/bin/python3.9, so the volume is
handed over as .npy and the result read back as .npz.  Used by tests marked
`skimage` (skipped where that interpreter is absent, e.g. on the GPU box) and
by tools/make_mc_golden.py.  TEST INFRASTRUCTURE ONLY.
"""
import os
import subprocess
import tempfile

import numpy as np

CONDA_PY = "/opt/conda/bin/python3.9"

_SCRIPT = r"""
import sys, warnings
warnings.filterwarnings("ignore")
import numpy as np
from skimage.measure import marching_cubes
vol = np.load(sys.argv[1]); level = float(sys.argv[2]); method = sys.argv[3]
try:
    v, f, _, _ = marching_cubes(vol, level, method=method)
    np.savez(sys.argv[4], v=v, f=f, err="")
except Exception as e:
    np.savez(sys.argv[4], v=np.zeros((0,3),np.float32), f=np.zeros((0,3),np.int32),
             err=type(e).__name__ + ":" + str(e))
"""


def available():
    if not os.path.exists(CONDA_PY):
        return False
    try:
        subprocess.check_output([CONDA_PY, "-c", "import skimage"], stderr=subprocess.DEVNULL)
        return True
    except Exception:
        return False


def marching_cubes(volume, level, method="lewiner"):
    with tempfile.TemporaryDirectory() as d:
        a, b = os.path.join(d, "in.npy"), os.path.join(d, "out.npz")
        np.save(a, volume)
        subprocess.check_call([CONDA_PY, "-c", _SCRIPT, a, repr(float(level)), method, b],
                              stderr=subprocess.DEVNULL)
        r = np.load(b)
        err = str(r["err"])
        if err:
            name, msg = err.split(":", 1)
            raise {"ValueError": ValueError, "RuntimeError": RuntimeError}.get(name, Exception)(msg)
        return r["v"], r["f"]
