#!/usr/bin/env python3
"""Generate tests/golden/mc_*.{json,npz} from the REAL scikit-image Lewiner kernel.

Runs only in the build container (needs /opt/conda/bin/python3.9 with
scikit-image 0.18.3; see oracle/mc_skimage.py).  The fixtures pin the oracle
(oracle/mc_lewiner.c) and, through it, the HIP marching-cubes kernels, to what the
reference path calls: skimage.measure.marching_cubes(grid, level,
method="lewiner") (upstream hy3dgen surface_extractors.MCSurfaceExtractor.run,
reached from reference src/2d_to_3d_models/run.py:77-84).

    python tools/make_mc_golden.py                  # (re)write tests/golden/mc_*
    python tools/make_mc_golden.py --check          # regenerate into a scratch directory and DIFF against the committed fixtures
    R3G_SKIMAGE_PYTHON=/path/to/python ...          # another interpreter with another scikit-image (oracle/README.md: the reference
                                                    # asks for >= 0.24.0, the container has 0.18.3): --check then says whether the pin moved
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mc_volumes import golden_volume, small_volumes, cube_zoo  # noqa: E402

CONDA_PY = os.environ.get("R3G_SKIMAGE_PYTHON", "/opt/conda/bin/python3.9")
BATCH = r"""
import sys, warnings
warnings.filterwarnings("ignore")
import numpy as np
from skimage.measure import marching_cubes
d = np.load(sys.argv[1]); out = {}
for k in d.files:
    if k.startswith("level_"): continue
    try:
        v, f, _, _ = marching_cubes(d[k], float(d["level_" + k]), method=sys.argv[3])
        out["v_" + k] = v; out["f_" + k] = f
    except Exception as e:
        out["e_" + k] = np.array(type(e).__name__)
np.savez(sys.argv[2], **out)
"""


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def run_skimage(vols, method="lewiner"):
    with tempfile.TemporaryDirectory() as d:
        a, b = os.path.join(d, "in.npz"), os.path.join(d, "out.npz")
        np.savez(a, **vols)
        subprocess.check_call([CONDA_PY, "-c", BATCH, a, b, method], stderr=subprocess.DEVNULL)
        r = np.load(b)
        return {k: r[k] for k in r.files}


def check():
    """regenerate with the interpreter at hand and compare with tests/golden/ (arrays bit for bit, the sha table entry by entry)"""
    committed = os.path.join(ROOT, "tests", "golden")
    with tempfile.TemporaryDirectory() as scratch:
        main(scratch)
        bad = []
        old, new = (json.load(open(os.path.join(d, "mc_sha.json"))) for d in (committed, scratch))
        print("committed fixtures: scikit-image %s; this interpreter: %s" % (old["_meta"]["skimage"], new["_meta"]["skimage"]))
        for k in sorted(set(old) | set(new)):
            if k != "_meta" and old.get(k) != new.get(k):
                bad.append("mc_sha.json[%s]" % k)
        for name in ("mc_small.npz", "mc_cubes.npz"):
            a, b = np.load(os.path.join(committed, name)), np.load(os.path.join(scratch, name))
            for k in sorted(set(a.files) | set(b.files)):
                if k not in a.files or k not in b.files or a[k].shape != b[k].shape or a[k].tobytes() != b[k].tobytes():
                    bad.append("%s[%s]" % (name, k))
    if bad:
        print("THE PIN MOVED: %d entries differ, e.g. %s" % (len(bad), bad[:8]))
        return 1
    print("identical: every fixture reproduces bit for bit")
    return 0


def main(gold=None):
    gold = gold or os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    # 1. SURVEY.md 4.3 vectors A-D (hashes only: D alone is 6.8 MB of output)
    vols = {}
    for name in "ABCD":
        vol, level = golden_volume(name)
        vols[name] = vol
        vols["level_" + name] = np.float64(level)
    r = run_skimage(vols)
    rc = run_skimage({k: v for k, v in vols.items() if k in ("B", "level_B")}, method="lorensen")
    table = {}
    for name in "ABCD":
        v, f = r["v_" + name], r["f_" + name]
        table[name] = dict(level=float(vols["level_" + name]), in_sha=sha(vols[name]), V=len(v), F=len(f),
                           faces_sha=sha(f.astype(np.int32)), verts_sha=sha(v.astype(np.float32)),
                           f0=f[0].tolist(), v0=[float(x) for x in v[0]])
    table["B_classic"] = dict(level=0.0, in_sha=sha(vols["B"]), V=len(rc["v_B"]), F=len(rc["f_B"]),
                              faces_sha=sha(rc["f_B"].astype(np.int32)),
                              verts_sha=sha(rc["v_B"].astype(np.float32)))
    table["_meta"] = dict(skimage=subprocess.check_output(
        [CONDA_PY, "-c", "import skimage;print(skimage.__version__)"], stderr=subprocess.DEVNULL).decode().strip(),
        note="faces int32 as returned by skimage (before hy3dgen's [:, ::-1]); verts float32 index space")
    with open(os.path.join(gold, "mc_sha.json"), "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    # 2. small volumes with full outputs
    vols = small_volumes()
    r = run_skimage(vols)
    out = dict(vols)
    out.update(r)
    np.savez_compressed(os.path.join(gold, "mc_small.npz"), **out)
    # 3. single-cube zoo: every Lewiner case incl. exact-zero ambiguity tests
    cubes = cube_zoo()
    vols = {}
    for i, c in enumerate(cubes):
        vols["c%05d" % i] = c
        vols["level_c%05d" % i] = np.float64(0.0)
    r = run_skimage(vols)
    nv = np.zeros(len(cubes), np.int32)
    nf = np.zeros(len(cubes), np.int32)
    vs, fs = [], []
    for i in range(len(cubes)):
        k = "c%05d" % i
        if "v_" + k in r:
            nv[i], nf[i] = len(r["v_" + k]), len(r["f_" + k])
            vs.append(r["v_" + k].reshape(-1).astype(np.float32))
            fs.append(r["f_" + k].reshape(-1).astype(np.int32))
    np.savez_compressed(os.path.join(gold, "mc_cubes.npz"), cubes=cubes, nv=nv, nf=nf,
                        v=np.concatenate(vs), f=np.concatenate(fs))
    print("golden written:", sorted(os.listdir(gold)))


if __name__ == "__main__":
    sys.exit(check() if "--check" in sys.argv[1:] else main())
