"""The text conditioning of the delighting model (hy3dgen.texgen.utils.dehighlight_utils.empty_prompt_embedding): upstream calls
its InstructPix2Pix pipeline with the prompt "" every time, so the CLIP text encoder's output is one constant per checkpoint.
A stock checkpoint folder (tokenizer/ + text_encoder/) is enough: the constant is computed at load time with the checkpoint's own
files; a prepared prompt_embeds_empty.safetensors wins when it is there.  Behind reference src/2d_to_3d_models/run.py:97, :126-128."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))


def _tiny_clip_folder(root):
    """a CLIP text tower with random weights and a byte-level tokenizer without merges, in the layout of a diffusers pipeline"""
    from transformers import CLIPTextConfig, CLIPTextModel

    def bytes_to_unicode():       # the byte -> printable character table of byte-level BPE vocabularies (GPT-2, CLIP)
        bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\u00a1"), ord("\u00ac") + 1)) + list(range(ord("\u00ae"), ord("\u00ff") + 1))
        cs = bs[:]
        n = 0
        for b in range(256):
            if b not in bs:
                bs.append(b)
                cs.append(256 + n)
                n += 1
        return dict(zip(bs, [chr(c) for c in cs]))
    tok_dir, enc_dir = os.path.join(root, "tokenizer"), os.path.join(root, "text_encoder")
    os.makedirs(tok_dir)
    chars = list(bytes_to_unicode().values())
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    with open(os.path.join(tok_dir, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(tok_dir, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n")
    with open(os.path.join(tok_dir, "tokenizer_config.json"), "w") as f:
        json.dump({"model_max_length": 13, "tokenizer_class": "CLIPTokenizer"}, f)
    torch.manual_seed(0)
    cfg = CLIPTextConfig(vocab_size=len(vocab), hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                         max_position_embeddings=13, bos_token_id=vocab["<|startoftext|>"], eos_token_id=vocab["<|endoftext|>"])
    model = CLIPTextModel(cfg).eval()
    model.save_pretrained(enc_dir)
    return model, vocab


def test_the_empty_prompt_is_encoded_with_the_checkpoints_own_files(tmp_path):
    pytest.importorskip("transformers")
    from hy3dgen.texgen.utils.dehighlight_utils import empty_prompt_embedding
    model, vocab = _tiny_clip_folder(str(tmp_path))
    emb = empty_prompt_embedding(str(tmp_path))
    assert emb.shape == (1, 13, 32) and emb.dtype == torch.float32
    # what diffusers' encode_prompt feeds the encoder for "": <|startoftext|>, then <|endoftext|> up to the model's length
    ids = torch.tensor([[vocab["<|startoftext|>"]] + [vocab["<|endoftext|>"]] * 12])
    with torch.no_grad():
        want = model(ids)[0]
    assert torch.allclose(emb, want, atol=1e-6)
    # a prepared file wins over the encoder
    from safetensors.torch import save_file
    fixed = torch.randn(1, 13, 32)
    save_file({"prompt_embeds": fixed}, os.path.join(str(tmp_path), "prompt_embeds_empty.safetensors"))
    assert torch.equal(empty_prompt_embedding(str(tmp_path)), fixed)


def test_a_folder_with_neither_is_refused(tmp_path):
    from hy3dgen.texgen.utils.dehighlight_utils import empty_prompt_embedding
    with pytest.raises(FileNotFoundError):
        empty_prompt_embedding(str(tmp_path))
