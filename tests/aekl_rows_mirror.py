"""Test support: a torch-CPU mirror of the DATA FLOW of 3d-re-gen_amd/csrc/unet.cpp's AutoencoderKL path (vae_encode /
vae_decode / vae_attention) on rows [H*W][C], reading the weights in exactly the layouts r3g.unet.prepare_aekl_weights hands
to the library (re-laid 3x3 kernels, K zero-padded to 64, rows zero-padded to 4).  It does not call the library and is not
product code: it lets the CPU suite check the host-side re-layouts and the algebra of the HIP path (im2col with the one-sided
padding, V produced transposed, v's bias moved behind the softmax) against the oracle without a GPU, and -- with bf16=True,
which rounds every GEMM operand to bf16 as the MFMA path does -- it says what difference from the fp32 oracle to expect."""
import torch
import torch.nn.functional as F


def _r(t, bf16):
    return t.to(torch.bfloat16).to(torch.float32) if bf16 else t


class Mirror:
    def __init__(self, weights, cfg, bf16=False):
        """weights: name -> (tensor, code) from prepare_aekl_weights(state_dict, "cpu")"""
        self.w = {k: v[0].to(torch.float32) for k, v in weights.items()}
        self.cfg = cfg
        self.bf16 = bf16
        self.eps = 1e-6

    # ---- the kernels' semantics
    def gemm(self, a, w, bias=None):
        out = _r(a, self.bf16) @ _r(w, self.bf16).t()
        return out if bias is None else out + bias.reshape(1, -1)

    def lin(self, a, name):
        return self.gemm(a, self.w[name + ".weight"], self.w.get(name + ".bias"))

    @staticmethod
    def im2col(x, H, W, C, stride, pad):
        xp = F.pad(x.reshape(H, W, C), (0, 0, pad, 1, pad, 1))
        Ho, Wo = (H + pad - 2) // stride + 1, (W + pad - 2) // stride + 1
        cols = [xp[ky:ky + stride * (Ho - 1) + 1:stride, kx:kx + stride * (Wo - 1) + 1:stride] for ky in range(3) for kx in range(3)]
        return torch.cat(cols, dim=-1).reshape(Ho * Wo, 9 * C), Ho, Wo

    def conv3x3(self, x, H, W, C, name, stride=1, pad=1):
        col, Ho, Wo = self.im2col(x, H, W, C, stride, pad)
        return self.lin(col, name), Ho, Wo

    def group_norm(self, x, name, silu):
        y = F.group_norm(x.t()[None], self.cfg["groups"], self.w[name + ".weight"].reshape(-1), self.w[name + ".bias"].reshape(-1),
                         self.eps)[0].t()
        return F.silu(y) if silu else y

    @staticmethod
    def cast_pad(x, C, Cpad):
        return torch.cat([x[:, :C], torch.zeros((x.shape[0], Cpad - C), dtype=x.dtype)], dim=1)

    # ---- unet.cpp
    def resnet(self, pre, x, H, W, cin, cout):
        h, _, _ = self.conv3x3(self.group_norm(x, pre + ".norm1", True), H, W, cin, pre + ".conv1")
        sc = self.lin(x, pre + ".conv_shortcut") if (pre + ".conv_shortcut.weight") in self.w else x
        o, _, _ = self.conv3x3(self.group_norm(h, pre + ".norm2", True), H, W, cout, pre + ".conv2")
        return sc + o

    def attention(self, pre, x, C):
        xn = self.group_norm(x, pre + ".group_norm", False)
        q, k = self.lin(xn, pre + ".to_q"), self.lin(xn, pre + ".to_k")
        vt = self.gemm(self.w[pre + ".to_v.weight"], xn)                       # V^T [C][hw], no bias
        s = self.gemm(_r(q, self.bf16), _r(k, self.bf16))
        p = torch.softmax(s * (float(C) ** -0.5), dim=-1)
        o = self.gemm(p, _r(vt, self.bf16), self.w[pre + ".to_v.bias"])       # + b_v behind the softmax
        return x + self.lin(_r(o, self.bf16), pre + ".to_out.0")

    def mid(self, pre, x, H, W, C):
        x = self.resnet(pre + ".resnets.0", x, H, W, C, C)
        x = self.attention(pre + ".attentions.0", x, C)
        return self.resnet(pre + ".resnets.1", x, H, W, C, C)

    def decode(self, z_rows, h, w):
        c = self.cfg
        ch, L, zc = c["block_out_channels"], c["layers_per_block"], c["latent_channels"]
        n = len(ch)
        zp = (zc + 3) // 4 * 4
        t = self.lin(self.cast_pad(z_rows, zc, 64), "post_quant_conv")         # [hw][zp]
        assert t.shape[1] == zp
        x, _, _ = self.conv3x3(self.cast_pad(t, zc, 64), h, w, 64, "decoder.conv_in")
        x = self.mid("decoder.mid_block", x, h, w, ch[-1])
        cc = ch[-1]
        for i in range(n):
            cout = ch[n - 1 - i]
            for j in range(L + 1):
                x = self.resnet("decoder.up_blocks.%d.resnets.%d" % (i, j), x, h, w, cc, cout)
                cc = cout
            if i < n - 1:
                x = x.reshape(h, w, cc).repeat_interleave(2, 0).repeat_interleave(2, 1).reshape(4 * h * w, cc)
                h, w = 2 * h, 2 * w
                x, _, _ = self.conv3x3(x, h, w, cc, "decoder.up_blocks.%d.upsamplers.0.conv" % i)
        out, _, _ = self.conv3x3(self.group_norm(x, "decoder.conv_norm_out", True), h, w, cc, "decoder.conv_out")
        return out, h, w                                                          # [hw][rup(image channels, 4)]

    def encode(self, img_rows, H, W):
        c = self.cfg
        ch, L, zc2 = c["block_out_channels"], c["layers_per_block"], 2 * c["latent_channels"]
        n = len(ch)
        x, _, _ = self.conv3x3(self.cast_pad(img_rows, c["image_channels"], 64), H, W, 64, "encoder.conv_in")
        cc = ch[0]
        for i in range(n):
            for j in range(L):
                x = self.resnet("encoder.down_blocks.%d.resnets.%d" % (i, j), x, H, W, cc, ch[i])
                cc = ch[i]
            if i < n - 1:
                x, H, W = self.conv3x3(x, H, W, cc, "encoder.down_blocks.%d.downsamplers.0.conv" % i, stride=2, pad=0)
        x = self.mid("encoder.mid_block", x, H, W, cc)
        t, _, _ = self.conv3x3(self.group_norm(x, "encoder.conv_norm_out", True), H, W, cc, "encoder.conv_out")
        return self.lin(self.cast_pad(t, zc2, 64), "quant_conv"), H, W          # [hw][rup(2 latent, 4)]
