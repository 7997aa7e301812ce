"""Host side of the texture stage's UNet blocks (include/r3g.h "UNet blocks"): diffusers state dict -> device weights in the
layouts the kernels read, and thin methods over the C ABI.  SURVEY.md 8(f) rank 3: the UNet (UnetBlocks) and the SD-family
VAE (AutoencoderKLBlocks) that upstream's texture pipelines are made of -- nothing in the stage uses them yet (hy3dgen.texgen
keeps reporting where its colours come from).

State-dict names are diffusers' (`unet/diffusion_pytorch_model.safetensors` of an SD-2.1-class model):
"down_blocks.0.resnets.0.conv1.weight", "mid_block.attentions.0.transformer_blocks.0.attn2.to_k.weight", ...
Activations cross this boundary as NCHW float tensors (diffusers' convention) and are turned into rows [H*W][C] here.
"""
import ctypes

import torch

from . import ffi as _l


def prepare_weights(sd, device):
    """pure re-layouts: 3x3 conv [O][I][3][3] -> [O][ky][kx][I]; 1x1 conv -> [O][I]; self-attentions (attn1, attn_multiview)
    to_q|to_k|to_v -> to_qkv rows; cross-attentions (attn2, attn_refview) to_k / to_v -> per head (64 k rows, 64 v rows), to_q
    kept; matrices bf16, vectors f32; "class_embedding.weight" stays an f32 table; the learned text embeddings are inputs of
    the model, not weights of the library (skipped)"""
    out = {}
    groups = {}
    for k, t in sd.items():
        if "learned_text_clip" in k:
            continue
        t = t.detach().to(torch.float32)
        if k == "class_embedding.weight":
            out[k] = (t.to(device=device).contiguous(), 0)
            continue
        base, leaf = k.rsplit(".", 1)
        parent, name = base.rsplit(".", 1) if "." in base else ("", base)
        if name in ("to_q", "to_k", "to_v") and leaf == "weight":
            groups.setdefault(parent, {})[name] = t
            if name == "to_q" and not parent.endswith(("attn1", "attn_multiview")):
                out[k] = (t.to(device=device, dtype=torch.bfloat16).contiguous(), 1)
            continue
        if t.ndim == 4:
            if t.shape[-1] == 3:
                t = t.permute(0, 2, 3, 1)                      # [O][ky][kx][I]
                if t.shape[-1] % 64:                           # conv_in: its few input channels are zero-padded to 64
                    pad = torch.zeros(t.shape[:3] + (64 - t.shape[-1] % 64,), dtype=t.dtype)
                    t = torch.cat([t, pad], dim=-1)
                t = t.reshape(t.shape[0], -1)
            else:
                t = t.reshape(t.shape[0], t.shape[1])
        if t.ndim == 2:
            out[k] = (t.to(device=device, dtype=torch.bfloat16).contiguous(), 1)
        else:
            out[k] = (t.reshape(1, -1).to(device=device).contiguous(), 0)
    for parent, g in groups.items():
        if parent.endswith(("attn1", "attn_multiview")):
            w = torch.cat([g["to_q"], g["to_k"], g["to_v"]], dim=0)
            out[parent + ".to_qkv.weight"] = (w.to(device=device, dtype=torch.bfloat16).contiguous(), 1)
        else:
            k_, v_ = g["to_k"], g["to_v"]
            heads = k_.shape[0] // 64
            w = torch.stack([k_.view(heads, 64, -1), v_.view(heads, 64, -1)], dim=1).reshape(2 * k_.shape[0], -1)
            out[parent + ".to_kv.weight"] = (w.to(device=device, dtype=torch.bfloat16).contiguous(), 1)
    return out


def split_2p5d_state_dict(sd):
    """state dict of upstream's UNet2p5DConditionModel ([UPSTREAM-RECALLED] names: "unet.*" with the wrapped blocks'
    parameters under "...transformer_blocks.0.transformer.*", "unet_dual.*" likewise) -> (generator weights, reference-copy
    weights, {"learned_text_clip_gen", "learned_text_clip_ref"}) under the plain diffusers names the library registers"""
    gen, ref, extra = {}, {}, {}
    for k, v in sd.items():
        if k.startswith("unet_dual."):
            dst, name = ref, k[len("unet_dual."):]
        elif k.startswith("unet."):
            dst, name = gen, k[len("unet."):]
        else:
            continue
        if name.startswith("learned_text_clip"):
            extra[name] = v
            continue
        dst[name.replace(".transformer_blocks.0.transformer.", ".transformer_blocks.0.")] = v
    return gen, ref, extra


def transformer_prefixes(n_levels, layers_per_block=2):
    """names of the Transformer2DModels of the SD-2.1 layout (the keys of the reference pass's kept states)"""
    out = []
    for i in range(n_levels - 1):
        out += ["down_blocks.%d.attentions.%d" % (i, j) for j in range(layers_per_block)]
    out.append("mid_block.attentions.0")
    for i in range(1, n_levels):
        out += ["up_blocks.%d.attentions.%d" % (i, j) for j in range(layers_per_block + 1)]
    return out


def to_rows(x):
    """NCHW [n, C, H, W] -> rows f32 [n*H*W, C] (sample after sample)"""
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).to(torch.float32).contiguous()


def from_rows(r, h, w, n=1):
    return r.reshape(n, h, w, -1).permute(0, 3, 1, 2).contiguous()


class _OwnsContext:
    """the wrapper owns one library context (its arena and weight table): close() -- or garbage collection -- gives the HBM back.
    A generator that reads another instance's kept states (set_condition) must not outlive that instance's close()."""
    ctx = None

    def close(self):
        ctx, self.ctx = self.ctx, None
        if ctx is None:
            return
        try:
            with torch.cuda.device(self.device):
                torch.cuda.synchronize()          # nothing in flight reads the arena or the weights any more
                self.L.r3g_destroy(ctx)
        except Exception:          # interpreter shutdown (modules already torn down): the process's memory goes with it
            pass
        self._w = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class UnetBlocks(_OwnsContext):
    def __init__(self, state_dict, max_hw, max_channels, temb_dim, ctx_dim, ctx_tokens, groups=32, resnet_eps=1e-5, device=0,
                 block_out_channels=(), layers_per_block=2, in_channels=4, out_channels=4):
        """block_out_channels: the level structure for forward() (empty: building blocks only); max_channels must then cover
        the widest cat(hidden, skip) of the up path (2 x the widest level)"""
        if not torch.cuda.is_available():
            raise RuntimeError("r3g.unet needs an MI355X: libr3g has no CPU path")
        self.device = torch.device("cuda", device)
        self.ctx = _l.new_context(device)          # a context of its own: the shape model keeps the shared one
        self.L = _l.lib()
        c = _l.UnetConfig(int(max_hw), int(max_channels), int(temb_dim), int(ctx_dim), int(ctx_tokens), int(groups),
                          float(resnet_eps), len(block_out_channels), int(layers_per_block), int(in_channels), int(out_channels),
                          (ctypes.c_int32 * 4)(*(list(block_out_channels) + [0] * (4 - len(block_out_channels)))))
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_unet_create(self.ctx, ctypes.byref(c)))
            self._w = prepare_weights(state_dict, self.device)
            for name, (t, code) in self._w.items():
                _l.check(self.L.r3g_unet_set_tensor(self.ctx, name.encode(), t.data_ptr(), code, t.shape[0], t.shape[1]))
            torch.cuda.synchronize()

    def _s(self):
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _in(self, x, temb=None, ctx=None):
        rows = to_rows(x.to(self.device))
        t = None if temb is None else temb.reshape(-1).to(self.device, torch.float32).contiguous()
        c = None if ctx is None else ctx[0].to(self.device, torch.bfloat16).contiguous()
        return rows, t, c

    def resnet(self, prefix, x, temb, c_out):
        _, cin, h, w = x.shape
        rows, t, _ = self._in(x, temb)
        out = torch.empty((h * w, c_out), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_unet_resnet(self.ctx, prefix.encode(), rows.data_ptr(), h, w, cin, c_out, t.data_ptr(),
                                            out.data_ptr(), self._s()))
        return from_rows(out, h, w)

    def transformer(self, prefix, x, ctx):
        _, c, h, w = x.shape
        rows, _, cx = self._in(x, None, ctx)
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_unet_transformer(self.ctx, prefix.encode(), rows.data_ptr(), h, w, c, cx.data_ptr(),
                                                 cx.shape[0], self._s()))
        return from_rows(rows, h, w)

    def downsample(self, prefix, x):
        _, c, h, w = x.shape
        rows, _, _ = self._in(x)
        ho, wo = (h + 1) // 2, (w + 1) // 2
        out = torch.empty((ho * wo, c), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_unet_downsample(self.ctx, prefix.encode(), rows.data_ptr(), h, w, c, out.data_ptr(), self._s()))
        return from_rows(out, ho, wo)

    def down_block(self, prefix, x, temb, ctx, c_out, layers=2, add_downsample=True):
        """-> (output, [hidden state of every layer (+ the downsampled output)]) as diffusers' CrossAttnDownBlock2D"""
        _, cin, h, w = x.shape
        rows, t, cx = self._in(x, temb, ctx)
        states = torch.empty((layers, h * w, c_out), dtype=torch.float32, device=self.device)
        ho, wo = (h + 1) // 2, (w + 1) // 2
        out = torch.empty((ho * wo, c_out), dtype=torch.float32, device=self.device) if add_downsample else None
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_unet_down_block(self.ctx, prefix.encode(), rows.data_ptr(), h, w, cin, c_out, t.data_ptr(),
                                                cx.data_ptr(), cx.shape[0], layers, int(bool(add_downsample)),
                                                states.data_ptr(), out.data_ptr() if add_downsample else None, self._s()))
        st = [from_rows(states[i], h, w) for i in range(layers)]
        if add_downsample:
            o = from_rows(out, ho, wo)
            return o, st + [o]
        return st[-1], st

    def forward(self, sample, timestep, ctx):
        """UNet2DConditionModel.forward: sample NCHW [1, in_channels, H, W], scalar timestep, ctx [1, tokens, ctx_dim] ->
        NCHW [1, out_channels, H, W]"""
        _, cin, h, w = sample.shape
        rows, _, cx = self._in(sample, None, ctx)
        out = torch.empty((h * w, self.out_channels), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_unet_forward(self.ctx, rows.data_ptr(), h, w, ctypes.c_float(float(timestep)), cx.data_ptr(),
                                             cx.shape[0], out.data_ptr(), self._s()))
        return from_rows(out, h, w)

    def forward_rows(self, rows, h, w, timestep, ctx_rows, out=None):
        """the same on rows already in HBM (a sampling loop keeps them there): rows f32 [h*w][in_channels], ctx_rows bf16
        [tokens][ctx_dim] -> f32 [h*w][out_channels]"""
        if out is None:
            out = torch.empty((h * w, self.out_channels), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_unet_forward(self.ctx, rows.data_ptr(), h, w, ctypes.c_float(float(timestep)), ctx_rows.data_ptr(),
                                             ctx_rows.shape[0], out.data_ptr(), self._s()))
        return out

    # ---- several samples per call, 2.5D blocks (include/r3g.h "several samples per call ...")
    def forward_mv(self, sample, timestep, ctx, class_labels=None, flags=0, mva_scale=1.0, ref_scale=1.0):
        """sample NCHW [n, in_channels, H, W] (the n views of ONE object) -> NCHW [n, out_channels, H, W]; ctx [1, tokens, dim]
        shared by the views; class_labels: n camera indices or None; flags 1 = keep the states for reference attention
        (the reference pass), 2 = use the registered ones (set_condition)"""
        if int(flags) & 4:
            raise ValueError("flag 4 (the guidance pair) needs both contexts stacked: use forward_mv_rows")
        n, cin, h, w = sample.shape
        rows, _, cx = self._in(sample, None, ctx)
        out = torch.empty((n * h * w, self.out_channels), dtype=torch.float32, device=self.device)
        lab = None
        if class_labels is not None:
            lab = (ctypes.c_int32 * n)(*[int(v) for v in class_labels])
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_unet_forward_mv(self.ctx, rows.data_ptr(), h, w, ctypes.c_float(float(timestep)), cx.data_ptr(),
                                                cx.shape[0], n, lab, int(flags), ctypes.c_float(float(mva_scale)),
                                                ctypes.c_float(float(ref_scale)), out.data_ptr(), self._s()))
        return from_rows(out, h, w, n)

    def forward_mv_rows(self, rows, n, h, w, timestep, ctx_rows, class_labels=None, flags=0, mva_scale=1.0, ref_scale=1.0, out=None):
        """the same on rows already in HBM: rows f32 [n*h*w][in_channels], ctx_rows bf16 [tokens][ctx_dim].  flags & 4: the n samples
        are a classifier-free-guidance pair (n / 2 conditional, then n / 2 unconditional views) and ctx_rows holds both contexts,
        [2 tokens][ctx_dim] (include/r3g.h: r3g_unet_forward_mv)"""
        if out is None:
            out = torch.empty((n * h * w, self.out_channels), dtype=torch.float32, device=self.device)
        lab = None if class_labels is None else (ctypes.c_int32 * n)(*[int(v) for v in class_labels])
        if (int(flags) & 4) and (ctx_rows.shape[0] % 2 or n % 2):
            raise ValueError("flag 4 (the guidance pair): an even number of samples and two stacked contexts [2 tokens][ctx_dim]")
        tokens = ctx_rows.shape[0] // 2 if (int(flags) & 4) else ctx_rows.shape[0]
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_unet_forward_mv(self.ctx, rows.data_ptr(), h, w, ctypes.c_float(float(timestep)), ctx_rows.data_ptr(),
                                                tokens, n, lab, int(flags), ctypes.c_float(float(mva_scale)),
                                                ctypes.c_float(float(ref_scale)), out.data_ptr(), self._s()))
        return out

    def transformer_mv(self, prefix, x, ctx, flags=0, mva_scale=1.0, ref_scale=1.0):
        n, c, h, w = x.shape
        rows, _, cx = self._in(x, None, ctx)
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_unet_transformer_mv(self.ctx, prefix.encode(), rows.data_ptr(), h, w, c, cx.data_ptr(), cx.shape[0],
                                                    n, int(flags), ctypes.c_float(float(mva_scale)),
                                                    ctypes.c_float(float(ref_scale)), self._s()))
        return from_rows(rows, h, w, n)

    def condition(self, prefix):
        """(device pointer, rows, cols) of what the last pass with flag 1 kept for the transformer `prefix`"""
        p, r, c = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
        _l.check(self.L.r3g_unet_condition(self.ctx, prefix.encode(), ctypes.byref(p), ctypes.byref(r), ctypes.byref(c)))
        return p.value, r.value, c.value

    def set_condition(self, prefix, source):
        """register the states `source` (another UnetBlocks: the reference copy) kept for `prefix` as this model's "cond:"
        tensor; the memory stays owned by `source` and is overwritten by its next reference pass (same stream: ordered)"""
        p, r, c = source.condition(prefix)
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_unet_set_tensor(self.ctx, ("cond:" + prefix).encode(), ctypes.c_void_p(p), 1, r, c))

    def mid_block(self, prefix, x, temb, ctx):
        _, c, h, w = x.shape
        rows, t, cx = self._in(x, temb, ctx)
        out = torch.empty((h * w, c), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_unet_mid_block(self.ctx, prefix.encode(), rows.data_ptr(), h, w, c, t.data_ptr(), cx.data_ptr(),
                                               cx.shape[0], out.data_ptr(), self._s()))
        return from_rows(out, h, w)


def prepare_aekl_weights(sd, device):
    """AutoencoderKL state dict -> the layouts r3g_aekl_encode / r3g_aekl_decode read (pure re-layouts and zero padding):
    3x3 conv [O][I][3][3] -> [O][ky][kx][I] with I zero-padded to a multiple of 64 (conv_in); 1x1 conv -> [O][I], I padded to
    64 (quant_conv, post_quant_conv); O (and the bias) zero-padded to a multiple of 4 (conv_out: 3 image channels); linear
    layers of the mid block's attention as they are; matrices bf16, vectors f32"""
    out = {}
    # checkpoints written before diffusers 0.18 name the mid block's attention layers query / key / value / proj_attn
    # (diffusers' own loader renames them the same way); their weights may also be stored as 1x1 convolutions [C][C][1][1]
    legacy = {".query.": ".to_q.", ".key.": ".to_k.", ".value.": ".to_v.", ".proj_attn.": ".to_out.0."}
    for k, t in sd.items():
        if ".attentions." in k:
            for old_name, new_name in legacy.items():
                k = k.replace(old_name, new_name)
        t = t.detach().to(torch.float32)
        if t.ndim == 4:
            if t.shape[-1] == 3:
                t = t.permute(0, 2, 3, 1)
                if t.shape[-1] % 64:
                    t = torch.cat([t, torch.zeros(t.shape[:3] + (64 - t.shape[-1] % 64,), dtype=t.dtype)], dim=-1)
                t = t.reshape(t.shape[0], -1)
            else:
                t = t.reshape(t.shape[0], t.shape[1])
                if t.shape[1] % 64:
                    t = torch.cat([t, torch.zeros((t.shape[0], 64 - t.shape[1] % 64), dtype=t.dtype)], dim=1)
        if t.ndim == 2:
            if t.shape[0] % 4:
                t = torch.cat([t, torch.zeros((4 - t.shape[0] % 4, t.shape[1]), dtype=t.dtype)], dim=0)
            out[k] = (t.to(device=device, dtype=torch.bfloat16).contiguous(), 1)
        else:
            t = t.reshape(-1)
            if t.numel() % 4:
                t = torch.cat([t, torch.zeros(4 - t.numel() % 4, dtype=t.dtype)])
            out[k] = (t.reshape(1, -1).to(device=device).contiguous(), 0)
    return out


class AutoencoderKLBlocks(_OwnsContext):
    """diffusers AutoencoderKL (SD family) on the HIP blocks: encode(image) -> moments, decode(latent) -> image"""

    def __init__(self, state_dict, block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4,
                 image_channels=3, groups=32, max_image_hw=512 * 512, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("r3g.unet needs an MI355X: libr3g has no CPU path")
        self.device = torch.device("cuda", device)
        self.ctx = _l.new_context(device)
        self.L = _l.lib()
        self.factor = 2 ** (len(block_out_channels) - 1)
        self.latent_channels, self.image_channels = int(latent_channels), int(image_channels)
        c = _l.UnetConfig(int(max_image_hw), int(max(block_out_channels)), 8, 64, 1, int(groups), 1e-6, len(block_out_channels),
                          int(layers_per_block), int(latent_channels), int(image_channels),
                          (ctypes.c_int32 * 4)(*(list(block_out_channels) + [0] * (4 - len(block_out_channels)))))
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_unet_create(self.ctx, ctypes.byref(c)))
            self._w = prepare_aekl_weights(state_dict, self.device)
            for name, (t, code) in self._w.items():
                _l.check(self.L.r3g_unet_set_tensor(self.ctx, name.encode(), t.data_ptr(), code, t.shape[0], t.shape[1]))
            torch.cuda.synchronize()

    def _s(self):
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def decode(self, z):
        """AutoencoderKL.decode(z).sample: z NCHW [1, latent, h, w] -> NCHW [1, image channels, f h, f w]"""
        _, zc, h, w = z.shape
        if zc != self.latent_channels:
            raise ValueError("latent has %d channels, the model %d" % (zc, self.latent_channels))
        rows = to_rows(z.to(self.device))
        f = self.factor
        out = torch.empty((f * h * f * w, (self.image_channels + 3) // 4 * 4), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_aekl_decode(self.ctx, rows.data_ptr(), h, w, out.data_ptr(), self._s()))
        return from_rows(out[:, :self.image_channels], f * h, f * w)

    def decode_rows(self, rows, h, w):
        """latent rows f32 [h*w][latent] -> image rows f32 [(f h)(f w)][rup(image channels, 4)]"""
        f = self.factor
        out = torch.empty((f * h * f * w, (self.image_channels + 3) // 4 * 4), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_aekl_decode(self.ctx, rows.data_ptr(), h, w, out.data_ptr(), self._s()))
        return out

    def encode(self, x):
        """parameters of AutoencoderKL.encode(x).latent_dist: x NCHW [1, image channels, H, W] -> NCHW [1, 2 latent, H/f, W/f]
        (mean | log-variance); latent_dist.mode() is the first half"""
        _, c, h, w = x.shape
        if c != self.image_channels:
            raise ValueError("image has %d channels, the model %d" % (c, self.image_channels))
        f = self.factor
        if h % f or w % f:
            raise ValueError("image size must be divisible by %d" % f)
        rows = to_rows(x.to(self.device))
        zc2 = 2 * self.latent_channels
        out = torch.empty(((h // f) * (w // f), (zc2 + 3) // 4 * 4), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_aekl_encode(self.ctx, rows.data_ptr(), h, w, out.data_ptr(), self._s()))
        return from_rows(out[:, :zc2], h // f, w // f)
