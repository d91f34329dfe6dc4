"""DINOv2 vision transformer, forward only (reference: depth/metric_depth/depth_anything_v2/dinov2.py and
dinov2_layers/{attention,block,mlp,layer_scale,patch_embed}.py).

Tokens live as one (B*N, C) bf16 matrix from the patch embedding to the last block: the linears are hipBLASLt GEMMs on
cached bf16 weight copies (ops.linear), attention is the MFMA flash kernel on the packed qkv output, LayerNorm /
GELU / LayerScale+residual are one kernel each.  Parameters keep the reference's names and float32 storage, so its
checkpoints load with load_state_dict.  Not built: training (drop path, masks), register tokens, the SwiGLU FFN of
ViT-g."""
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hipops as ops


class PatchEmbed(nn.Module):
    """patch_embed.py:26-81: Conv2d(kernel = stride = patch) == one GEMM over the flattened patches"""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        B, C, H, W = x.shape
        p = self.patch_size[0]
        assert H % p == 0 and W % p == 0, f"Input image size {H}x{W} is not a multiple of the patch size {p}"
        ph, pw = H // p, W // p
        cols = x.reshape(B, C, ph, p, pw, p).permute(0, 2, 4, 1, 3, 5).reshape(B * ph * pw, C * p * p)
        K = cols.shape[1]
        Kp = (K + 7) // 8 * 8                       # row length of the GEMM operands: multiple of 16 bytes
        w = self.proj.weight.reshape(self.embed_dim, K)
        if Kp != K:
            cols, w = F.pad(cols, (0, Kp - K)), F.pad(w, (0, Kp - K))
        y = torch.addmm(self.proj.bias.to(torch.bfloat16), cols.to(torch.bfloat16), w.to(torch.bfloat16).t())
        return y, ph, pw                             # (B*ph*pw, D)


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class Attention(nn.Module):
    """attention.py:27-62"""

    def __init__(self, dim, num_heads=8, qkv_bias=False, proj_bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)

    def forward(self, x, B, N):
        qkv = ops.linear(x, self.qkv.weight, self.qkv.bias)                        # (B*N, 3*H*D) == (B,N,3,H,D)
        a = ops.attention(qkv, B, N, self.num_heads, self.head_dim, self.scale)
        return ops.linear(a, self.proj.weight, self.proj.bias)


MemEffAttention = Attention


class Mlp(nn.Module):
    """mlp.py:17-40"""

    def __init__(self, in_features, hidden_features=None, out_features=None, bias=True):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features, bias=bias)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features, bias=bias)

    def forward(self, x):
        return ops.linear(ops.gelu_(ops.linear(x, self.fc1.weight, self.fc1.bias)), self.fc2.weight, self.fc2.bias)


class Block(nn.Module):
    """block.py:37-110 in eval mode: x += ls1(attn(norm1 x)); x += ls2(mlp(norm2 x))"""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, proj_bias=True, ffn_bias=True, init_values=None,
                 norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, proj_bias=proj_bias)
        self.ls1 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), bias=ffn_bias)
        self.ls2 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()

    def forward(self, x, B, N, h=None, next_norm=None):
        """x: residual stream; h: LayerNorm1(x) if the previous block already produced it.  Returns (x_out, h_next):
        h_next = next_norm(x_out) when `next_norm` is given -- every "x + ls * branch" is fused with the LayerNorm that
        follows it (the second one of this block, or the first of the next block)."""
        g1 = self.ls1.gamma if isinstance(self.ls1, LayerScale) else None
        g2 = self.ls2.gamma if isinstance(self.ls2, LayerScale) else None
        if h is None:
            h = ops.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        x, h = ops.scale_residual_layernorm(x, self.attn(h, B, N), g1, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        m = self.mlp(h)
        if next_norm is None:
            return ops.scale_residual(x, m, g2), None
        return ops.scale_residual_layernorm(x, m, g2, next_norm.weight, next_norm.bias, next_norm.eps)


class DinoVisionTransformer(nn.Module):
    """dinov2.py:39-330 (block_chunks = 0, no register tokens, MLP FFN)"""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 qkv_bias=True, ffn_bias=True, proj_bias=True, init_values=None, ffn_layer="mlp", block_chunks=0,
                 num_register_tokens=0, interpolate_antialias=False, interpolate_offset=0.1):
        super().__init__()
        if ffn_layer != "mlp" or block_chunks != 0 or num_register_tokens != 0:
            raise NotImplementedError("only the MLP FFN, unchunked blocks and no register tokens are built (vits / vitb / vitl)")
        if embed_dim // num_heads != 64:
            raise NotImplementedError("the attention kernel is built for head dimension 64")
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 1
        self.n_blocks = depth
        self.num_heads = num_heads
        self.patch_size = patch_size
        self.num_register_tokens = 0
        self.interpolate_antialias = interpolate_antialias
        self.interpolate_offset = interpolate_offset
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.register_tokens = None
        self.chunked_blocks = False
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias, proj_bias, ffn_bias, init_values,
                                           norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Identity()
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))
        self._pos_cache = {}
        self.init_weights()

    def init_weights(self):
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def interpolate_pos_encoding(self, npatch, w, h):
        """dinov2.py:179-209 (bicubic resize of the patch position table, cached per input size and weight version)"""
        N = self.pos_embed.shape[1] - 1
        if npatch == N and w == h:
            return self.pos_embed
        key = (w, h, self.pos_embed._version, self.pos_embed.data_ptr())
        hit = self._pos_cache.get("k") == key
        if not hit:
            pos = self.pos_embed.float()
            dim = pos.shape[-1]
            w0, h0 = w // self.patch_size + self.interpolate_offset, h // self.patch_size + self.interpolate_offset
            sqrt_N = math.sqrt(N)
            grid = F.interpolate(pos[:, 1:].reshape(1, int(sqrt_N), int(sqrt_N), dim).permute(0, 3, 1, 2),
                                 scale_factor=(float(w0) / sqrt_N, float(h0) / sqrt_N), mode="bicubic",
                                 antialias=self.interpolate_antialias)
            assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
            self._pos_cache = {"k": key, "v": torch.cat((pos[:, :1], grid.permute(0, 2, 3, 1).view(1, -1, dim)), dim=1)}
        return self._pos_cache["v"]

    def prepare_tokens(self, x):
        """dinov2.py:211-231 without masks: (B,3,H,W) float -> ((B*N, C) bf16, B, N).  The image's first spatial axis is
        called `w` in the reference as well."""
        B, _, w, h = x.shape
        tok, ph, pw = self.patch_embed(x)
        D = self.embed_dim
        tok = torch.cat((self.cls_token.to(tok.dtype).expand(B, 1, D), tok.view(B, ph * pw, D)), dim=1)
        tok = tok + self.interpolate_pos_encoding(ph * pw, w, h).to(tok.dtype)
        return tok.reshape(B * (ph * pw + 1), D).contiguous(), B, ph * pw + 1

    def get_intermediate_layers(self, x, n=1, reshape=False, return_class_token=False, norm=True):
        """dinov2.py:284-308: outputs of the blocks listed in `n` (or the last n), normalised, split into patch tokens and
        the class token; tensors come back as (B, N-1, C) / (B, C) bf16."""
        if not x.is_cuda:
            raise RuntimeError("3dod_amd.depth_anything_v2 runs on the GPU only (no CPU path)")
        with torch.no_grad():
            tok, B, N = self.prepare_tokens(x)
            take = range(len(self.blocks) - n, len(self.blocks)) if isinstance(n, int) else n
            outs = []
            h = None
            for i, blk in enumerate(self.blocks):
                nxt = self.blocks[i + 1].norm1 if i + 1 < len(self.blocks) else None
                tok, h = blk(tok, B, N, h, nxt)
                if i in take:
                    outs.append(tok)
            assert len(outs) == len(take), f"only {len(outs)} / {len(take)} blocks found"
            if norm:
                outs = [ops.layernorm(o, self.norm.weight, self.norm.bias, self.norm.eps) for o in outs]
            outs = [o.view(B, N, -1) for o in outs]
            cls = [o[:, 0] for o in outs]
            outs = [o[:, 1:] for o in outs]
            if reshape:
                _, _, w, h = x.shape
                outs = [o.reshape(B, w // self.patch_size, h // self.patch_size, -1).permute(0, 3, 1, 2).contiguous() for o in outs]
            return tuple(zip(outs, cls)) if return_class_token else tuple(outs)

    def forward(self, x):
        (_, cls), = self.get_intermediate_layers(x, 1, return_class_token=True)
        return cls


def DINOv2(model_name):
    """dinov2.py:397-414"""
    dims = {"vits": (384, 12, 6), "vitb": (768, 12, 12), "vitl": (1024, 24, 16)}
    if model_name not in dims:
        raise NotImplementedError(f"encoder {model_name} is not built (vits, vitb, vitl)")
    d, depth, heads = dims[model_name]
    return DinoVisionTransformer(img_size=518, patch_size=14, embed_dim=d, depth=depth, num_heads=heads, mlp_ratio=4,
                                 init_values=1.0, ffn_layer="mlp", block_chunks=0, num_register_tokens=0,
                                 interpolate_antialias=False, interpolate_offset=0.1)
