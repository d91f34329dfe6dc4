"""TEST INFRASTRUCTURE -- CPU restatement of the Depth-Anything-V2 forward (DINOv2 ViT encoder + DPT head), float32, plain
torch functions on a state dict with the reference's keys.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may use this module; the product (3dod_amd.depth_anything_v2) never does.

Follows the reference (paths into /root/reference/depth/metric_depth/depth_anything_v2):
    dinov2.py:179-231        position-table resize (bicubic, offset 0.1) and token preparation
    dinov2.py:284-308        get_intermediate_layers: tapped block outputs, final LayerNorm, class token split off
    dinov2_layers/attention.py:50-65   plain softmax attention (xformers absent: MemEffAttention falls back to it)
    dinov2_layers/block.py:85-110      x + ls1(attn(norm1 x)); x + ls2(mlp(norm2 x))
    dinov2_layers/mlp.py, layer_scale.py, patch_embed.py
    dpt.py:116-151           DPT head forward;  dpt.py:180-189  DepthAnythingV2.forward (sigmoid head x max_depth)
    util/blocks.py:32-148    ResidualConvUnit, FeatureFusionBlock (bilinear, align_corners=True)
Pinned by tests/test_depth_oracle.py against tests/golden/depth_anything_vits.npz, the output of the reference's own model
(tests/golden/make_golden_depth.py)."""
import math

import torch
import torch.nn.functional as F

LAYER_IDX = {'vits': [2, 5, 8, 11], 'vitb': [2, 5, 8, 11], 'vitl': [4, 11, 17, 23]}
ENCODERS = {'vits': (384, 12, 6), 'vitb': (768, 12, 12), 'vitl': (1024, 24, 16)}      # embed dim, depth, heads


def _pos_table(sd, npatch, w, h, patch=14, offset=0.1):
    pos = sd["pretrained.pos_embed"].float()
    N = pos.shape[1] - 1
    if npatch == N and w == h:
        return pos
    dim = pos.shape[-1]
    w0, h0 = w // patch + offset, h // patch + offset
    sq = math.sqrt(N)
    grid = F.interpolate(pos[:, 1:].reshape(1, int(sq), int(sq), dim).permute(0, 3, 1, 2),
                         scale_factor=(float(w0) / sq, float(h0) / sq), mode="bicubic", antialias=False)
    assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
    return torch.cat((pos[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)


def encoder_features(sd, x, encoder):
    """the four tapped (patch tokens (B, N, C), class token (B, C)) pairs, final norm applied"""
    dim, depth, heads = ENCODERS[encoder]
    B, _, w, h = x.shape
    t = F.conv2d(x, sd["pretrained.patch_embed.proj.weight"], sd["pretrained.patch_embed.proj.bias"], stride=14)
    ph, pw = t.shape[-2:]
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat((sd["pretrained.cls_token"].expand(B, -1, -1), t), dim=1) + _pos_table(sd, ph * pw, w, h)
    N, hd = t.shape[1], dim // heads
    outs = []
    for i in range(depth):
        p = f"pretrained.blocks.{i}."
        y = F.layer_norm(t, (dim,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
        qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
        a = ((qkv[0] * hd ** -0.5) @ qkv[1].transpose(-2, -1)).softmax(dim=-1)
        y = (a @ qkv[2]).transpose(1, 2).reshape(B, N, dim)
        t = t + sd[p + "ls1.gamma"] * F.linear(y, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        y = F.layer_norm(t, (dim,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
        y = F.linear(F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        t = t + sd[p + "ls2.gamma"] * y
        if i in LAYER_IDX[encoder]:
            o = F.layer_norm(t, (dim,), sd["pretrained.norm.weight"], sd["pretrained.norm.bias"], 1e-6)
            outs.append((o[:, 1:], o[:, 0]))
    return outs, ph, pw


def _conv(sd, key, x, stride=1, padding=0):
    return F.conv2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=stride, padding=padding)


def _rcu(sd, key, x):
    y = _conv(sd, key + ".conv1", F.relu(x), padding=1)
    return _conv(sd, key + ".conv2", F.relu(y), padding=1) + x


def _fusion(sd, key, *xs, size=None):
    out = xs[0]
    if len(xs) == 2:
        out = out + _rcu(sd, key + ".resConfUnit1", xs[1])
    out = _rcu(sd, key + ".resConfUnit2", out)
    kw = {"scale_factor": 2} if size is None else {"size": tuple(size)}
    out = F.interpolate(out, **kw, mode="bilinear", align_corners=True)
    return _conv(sd, key + ".out_conv", out)


def forward(sd, x, encoder="vitl", max_depth=20.0):
    """sd: state dict with the reference's keys (float32 CPU tensors); x (B,3,H,W) float32, H and W multiples of 14 ->
    (B,H,W) float32 metric depth"""
    sd = {k: v.float() for k, v in sd.items()}
    feats, ph, pw = encoder_features(sd, x, encoder)
    h = "depth_head."
    layers = []
    for i, (tok, _) in enumerate(feats):
        t = tok.permute(0, 2, 1).reshape(tok.shape[0], tok.shape[-1], ph, pw)
        t = _conv(sd, f"{h}projects.{i}", t)
        if i == 0:
            t = F.conv_transpose2d(t, sd[h + "resize_layers.0.weight"], sd[h + "resize_layers.0.bias"], stride=4)
        elif i == 1:
            t = F.conv_transpose2d(t, sd[h + "resize_layers.1.weight"], sd[h + "resize_layers.1.bias"], stride=2)
        elif i == 3:
            t = _conv(sd, h + "resize_layers.3", t, stride=2, padding=1)
        layers.append(_conv(sd, f"{h}scratch.layer{i + 1}_rn", t, padding=1))
    l1, l2, l3, l4 = layers
    p4 = _fusion(sd, h + "scratch.refinenet4", l4, size=l3.shape[2:])
    p3 = _fusion(sd, h + "scratch.refinenet3", p4, l3, size=l2.shape[2:])
    p2 = _fusion(sd, h + "scratch.refinenet2", p3, l2, size=l1.shape[2:])
    p1 = _fusion(sd, h + "scratch.refinenet1", p2, l1)
    o = _conv(sd, h + "scratch.output_conv1", p1, padding=1)
    o = F.interpolate(o, (int(ph * 14), int(pw * 14)), mode="bilinear", align_corners=True)
    o = F.relu(_conv(sd, h + "scratch.output_conv2.0", o, padding=1))
    o = torch.sigmoid(_conv(sd, h + "scratch.output_conv2.2", o))
    return (o * max_depth).squeeze(1)
