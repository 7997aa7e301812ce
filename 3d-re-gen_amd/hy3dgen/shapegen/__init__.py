"""hy3dgen.shapegen -- the names the reference stage imports (src/2d_to_3d_models/run.py:11-16)."""
from .pipelines import Hunyuan3DDiTFlowMatchingPipeline, Hunyuan3DDiTPipeline  # noqa: F401
from .postprocessors import FaceReducer, FloaterRemover, DegenerateFaceRemover  # noqa: F401
from .preprocessors import ImageProcessorV2  # noqa: F401
