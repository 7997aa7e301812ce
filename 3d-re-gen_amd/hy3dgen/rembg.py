"""hy3dgen.rembg -- import surface only.  On the reference path the remover is constructed but never
called (src/2d_to_3d_models/run.py:69-71 converts to RGBA before testing for RGB)."""


class BackgroundRemover:
    def __call__(self, image):
        raise NotImplementedError("background removal is dead code on the reference path (run.py:69-71); "
                                  "crops arrive as RGBA from the segmentation stage")
