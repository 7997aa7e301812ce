"""tools/verify_checkpoint.py, the part that runs without a GPU: the key diff of a snapshot in upstream's layout against the
oracle's module tree (strict) and the product's weight table.  The numeric part is
tests/test_verify_checkpoint_gpu.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_snapshot(tmp_path, mutate=None):
    import yaml
    from safetensors.torch import save_file
    from oracle import hy3d_torch as H
    from parity_support import bf16_round_matrices
    from test_host_cpu import _snapshot_doc
    cfg = H.tiny_config()
    sd = {k: v.contiguous() for k, v in bf16_round_matrices(H.synthetic_state_dict(cfg, seed=5)).items()}
    if mutate:
        mutate(sd)
    sub = tmp_path / "hunyuan3d-dit-v2-0"
    sub.mkdir()
    (sub / "config.yaml").write_text(yaml.safe_dump(_snapshot_doc(cfg)))
    save_file(sd, str(sub / "model.fp16.safetensors"))
    return cfg


def run_tool(tmp_path, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verify_checkpoint.py"), str(tmp_path)] + list(args),
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
    return r, json.loads(r.stdout.strip().splitlines()[-1])


def test_key_diff_of_a_good_snapshot(tmp_path):
    write_snapshot(tmp_path)
    r, rep = run_tool(tmp_path, "--keys-only")
    assert r.returncode == 0, r.stderr[-2000:]
    k = rep["shape"]["keys"]
    assert k["ok"] and not k["oracle_missing"] and not k["product_table_unexpected"] and k["tensors"] > 50


def test_key_diff_names_what_is_wrong(tmp_path):
    def mutate(sd):
        sd["model.double_blocks.0.img_attn.qkv_renamed.weight"] = sd.pop("model.double_blocks.0.img_attn.qkv.weight")
        sd["vae.post_kl.weight"] = sd["vae.post_kl.weight"][:, :-1].contiguous()
    write_snapshot(tmp_path, mutate)
    r, rep = run_tool(tmp_path, "--keys-only")
    assert r.returncode == 1
    k = rep["shape"]["keys"]
    assert "model.double_blocks.0.img_attn.qkv.weight" in k["oracle_missing"]
    assert "model.double_blocks.0.img_attn.qkv_renamed.weight" in k["oracle_unexpected"]
    assert any(m[0] == "vae.post_kl.weight" for m in k["oracle_shape_mismatch"])
    assert "qkv_renamed" in r.stderr
