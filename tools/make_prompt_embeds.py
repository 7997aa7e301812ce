#!/usr/bin/env python3
"""One-time preparation of a delighting checkpoint for this path: the text embedding of the (fixed) prompt.

    python tools/make_prompt_embeds.py <checkpoint dir> [--prompt ""] [--out prompt_embeds_empty.safetensors]

Upstream's Light_Shadow_Remover calls its InstructPix2Pix pipeline with prompt "" every time, so the text encoder's output is one
constant per checkpoint; the CLIP text encoder is not part of the MI355X path.  This tool computes that constant once on the CPU
with the checkpoint's OWN tokenizer/ and text_encoder/ through `transformers` (diffusers is not needed), exactly as diffusers'
`encode_prompt` does -- `text_encoder(tokenizer(prompt, padding="max_length", max_length=model_max_length, truncation=True).input_ids)[0]`,
the last hidden state -- and writes it next to unet/ and vae/ where hy3dgen.texgen.utils.dehighlight_utils.Light_Shadow_Remover.load
looks for it.  Not product code: nothing under 3d-re-gen_amd/ imports it."""
import argparse
import os
import sys


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("checkpoint", help="a diffusers pipeline folder with tokenizer/ and text_encoder/")
    ap.add_argument("--prompt", default="")
    ap.add_argument("--out", default="prompt_embeds_empty.safetensors")
    a = ap.parse_args(argv)
    import torch
    from safetensors.torch import save_file
    from transformers import CLIPTextModel, CLIPTokenizer
    tok = CLIPTokenizer.from_pretrained(os.path.join(a.checkpoint, "tokenizer"))
    enc = CLIPTextModel.from_pretrained(os.path.join(a.checkpoint, "text_encoder")).to(torch.float32).eval()
    ids = tok(a.prompt, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt").input_ids
    with torch.no_grad():
        emb = enc(ids)[0]
    out = os.path.join(a.checkpoint, a.out)
    save_file({"prompt_embeds": emb.contiguous()}, out)
    print("wrote %s: prompt %r -> %s" % (out, a.prompt, tuple(emb.shape)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
