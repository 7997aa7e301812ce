"""hy3dgen.texgen -- the texture stage's import surface (reference: `from hy3dgen.texgen import Hunyuan3DPaintPipeline`,
src/2d_to_3d_models/run.py:17).  See pipelines.py for what is and is not built on the MI355X path."""
from .pipelines import Hunyuan3DPaintPipeline  # noqa: F401
