"""tools/verify_checkpoint.py end to end on a synthetic snapshot in upstream's layout: keys, per-block deltas
with the residual magnitudes, sampling on both residual-stream types, VAE, grid logits -- the command a real checkpoint gets."""
import pytest

from test_verify_checkpoint_cpu import run_tool, write_snapshot

pytestmark = pytest.mark.gpu


def test_the_tool_passes_a_synthetic_snapshot_and_reports_the_stream_magnitudes(tmp_path):
    import json
    import os
    import subprocess
    import sys
    write_snapshot(tmp_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "verify_checkpoint.py"), str(tmp_path), "--steps", "3", "--blocks", "-1",
                        "--slab", "4096"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    s = rep["shape"]
    assert rep["ok"] and s["keys"]["ok"]
    assert len(s["blocks"]) == 2 + 3 and all(b["delta_rel_l2"] <= 5e-3 for b in s["blocks"])       # tiny config: 2 double + 3 single
    assert 0 < s["max_abs_residual"] < 65504 and s["fp16_stream_has_headroom"]
    assert s["sampling"]["dit_f16_fallbacks"] == 0 and s["sampling"]["fp16_stream"] <= 2e-2 and s["sampling"]["fp32_stream"] <= 2e-2
    assert s["vae_rel_l2"] <= 8e-3 and s["grid_logits_max_err_over_max"] <= 1e-2
    assert rep["texture"]["skipped"]
