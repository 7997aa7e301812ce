"""csrc/wave_sum.h (round 6): the 64-lane sum on permlane swaps + DPP must give every lane the BITS of the `__shfl_xor` butterfly it
replaced (same pairings in the same order) -- LayerNorm rows and the geo decoder's chunk statistics go through it, and the bit-identity
claims between launch shapes rest on it.  tools/ubench/wave_sum_check.hip compares the two on 4 M lanes of mixed-magnitude data."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dpp_wave_sum_has_the_bits_of_the_shuffle_butterfly(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc is not on this box")
    exe = str(tmp_path / "wave_sum_check")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-o", exe,
                           os.path.join(ROOT, "tools", "ubench", "wave_sum_check.hip")], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "wave_sum: 0 of" in out.stdout, out.stdout
