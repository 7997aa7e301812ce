"""Stand-in for the `trimesh` package where it is not installed (it is not in this image).

The reference stage imports trimesh only for `isinstance(mesh, trimesh.Trimesh)` in its optional remesh step
(src/2d_to_3d_models/run.py:21,36); the mesh object itself comes from hy3dgen.  With this directory on PYTHONPATH
*instead of* a real trimesh, that check sees the mesh class of the MI355X path.  Do NOT put it on the path when the real
trimesh is installed."""
from r3g.mesh import Mesh as Trimesh, load_glb as load  # noqa: F401

__r3g_compat__ = True
__version__ = "0+r3g.compat"
