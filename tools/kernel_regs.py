#!/usr/bin/env python3
"""List VGPR / AGPR / scratch / LDS of every kernel in a --save-temps gfx950 .s file (metadata section)."""
import re
import sys

txt = open(sys.argv[1]).read()
for blk in txt.split("  - .agpr_count:")[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    name = re.sub(r"_ZN3r3g12_GLOBAL__N_1\d+", "", name)
    print(f"{name[:48]:50s} vgpr {g('vgpr_count'):>4} agpr {g('agpr_count'):>4} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}")
