"""hy3dgen.texgen -- import surface only.  The texture stage (multiview diffusion UNets + native
rasteriser) is SURVEY.md section 8(f) rank 3, outside this round's hot path: the pipeline object exists
so that the reference stage script runs unmodified, and returns the untextured mesh."""


class Hunyuan3DPaintPipeline:
    implemented = False

    @classmethod
    def from_pretrained(cls, model_path, **kwargs):
        return cls()

    def __call__(self, mesh, image=None, **kwargs):
        print("[hy3dgen.texgen] texture generation is not part of the MI355X hot path yet "
              "(SURVEY.md 8f rank 3): returning the untextured mesh")
        return mesh
