"""Loaded automatically by the interpreter that runs the reference's stage script in tests/test_reference_script.py
(this directory is first on its PYTHONPATH).  TEST INFRASTRUCTURE, two things:
  * huggingface_hub.snapshot_download (reference src/2d_to_3d_models/run.py:119-120, 201-202; there is no network) is
    replaced by a function that returns the local synthetic snapshot directory named by R3G_TEST_SNAPSHOT;
  * R3G_TEST_CPU_SHIM=1 installs the CPU compute stand-ins of tests/ref_shim.py, which let the UNMODIFIED reference
    script run against the hy3dgen mirror's API on a machine without a GPU (the product itself has no CPU path)."""
import os
import sys

if os.environ.get("R3G_TEST_SNAPSHOT"):
    import huggingface_hub

    def _snapshot_download(repo_id=None, **kwargs):
        path = os.environ["R3G_TEST_SNAPSHOT"]
        print("[test stub] snapshot_download(%r) -> %s" % (repo_id, path))
        return path
    huggingface_hub.snapshot_download = _snapshot_download

for _p in filter(None, os.environ.get("R3G_TEST_EXTRA_PATH", "").split(os.pathsep)):
    if _p not in sys.path:      # (the orchestrator test: reference run.py:72-86 REPLACES PYTHONPATH for the stage it spawns)
        sys.path.append(_p)

if os.environ.get("R3G_TEST_CPU_SHIM") == "1":
    here = os.path.dirname(os.path.realpath(__file__))     # (this file may be reached through a symlink)
    sys.path.insert(0, os.path.dirname(here))
    import ref_shim
    ref_shim.install()

if os.environ.get("R3G_TEST_REPORT_MAPS") == "1":
    # the GPU twin of the reference-script test asks the UNMODIFIED script's interpreter which native libraries it ended up with
    import atexit

    def _report_maps():
        try:
            with open("/proc/self/maps") as f:
                libs = sorted({ln.split("/")[-1].strip() for ln in f if "libr3g" in ln})
        except OSError:
            libs = []
        print("[test stub] native libraries mapped at exit: %s" % (", ".join(libs) or "none"), flush=True)
    atexit.register(_report_maps)
