"""hy3dgen -- drop-in mirror of the import surface the reference stage uses
(src/2d_to_3d_models/run.py:10-17), backed by libr3g.so on MI355X."""
