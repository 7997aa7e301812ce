"""CPU oracles for the Hunyuan_2d_to_3d hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (3d-re-gen_amd/) never does.
"""
