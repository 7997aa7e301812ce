#!/bin/bash
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_unet_gpu.py tests/test_aekl_gpu.py tests/test_pix2pix_gpu.py tests/test_unet2p5d_gpu.py tests/test_mvpaint_gpu.py tests/test_texgen_gpu.py -m gpu -q -x 2>&1 | tail -8
timeout 600 python tests/tex_stage_time.py > gpurun_out/r04_texture_stage_time_after.json 2>/dev/null
python -c "
import json; r=json.load(open('gpurun_out/r04_texture_stage_time_after.json')); print(r['times_ms'])"
