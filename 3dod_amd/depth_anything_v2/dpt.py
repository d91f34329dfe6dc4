"""DPT head and the DepthAnythingV2 model, forward only (reference: depth/metric_depth/depth_anything_v2/dpt.py).
Activations are NHWC bf16: the encoder's (B, N, C) patch tokens ARE the (B, ph, pw, C) feature map, so the reference's
permute to NCHW disappears; ConvTranspose2d with kernel == stride is one GEMM plus a pixel shuffle."""
import torch
import torch.nn as nn

from .. import hipops as ops
from .dinov2 import DINOv2
from .util.blocks import FeatureFusionBlock, _make_scratch, conv


def _make_fusion_block(features, use_bn, size=None):
    return FeatureFusionBlock(features, nn.ReLU(False), deconv=False, bn=use_bn, expand=False, align_corners=True, size=size)


def deconv_ks(x, m):
    """nn.ConvTranspose2d with kernel_size == stride, padding 0, on NHWC bf16: every input pixel writes its own k x k
    output patch -> (B*h*w, Cin) @ (Cin, k*k*Cout), then (B,h,w,k,k,Co) -> (B,h*k,w*k,Co)."""
    k = m.kernel_size[0]
    assert m.stride[0] == k and m.padding[0] == 0
    B, h, w, Cin = x.shape
    Cout = m.weight.shape[1]
    # the GEMM operand (k*k*Cout, Cin) and the bias repeated per patch position are cached on the module per weight version
    ent = getattr(m, "_cr_deconv", None)
    tag = (m.weight._version, m.weight.data_ptr(), None if m.bias is None else m.bias._version)
    if ent is None or ent[0] != tag:
        wm = m.weight.detach().permute(2, 3, 1, 0).reshape(k * k * Cout, Cin).to(torch.bfloat16).contiguous()
        bias = None if m.bias is None else m.bias.detach().float().repeat(k * k).contiguous()
        ent = m._cr_deconv = (tag, wm, bias)
    y = ops.linear_fwd_raw(x.reshape(B * h * w, Cin), ent[1], ent[2]).view(B, h, w, k, k, Cout)      # cr_linear_fwd
    return y.permute(0, 1, 3, 2, 4, 5).reshape(B, h * k, w * k, Cout).contiguous()


class DPTHead(nn.Module):
    """dpt.py:38-156"""

    def __init__(self, in_channels, features=256, use_bn=False, out_channels=(256, 512, 1024, 1024), use_clstoken=False):
        super().__init__()
        if use_clstoken:
            raise NotImplementedError("use_clstoken is False in every Depth-Anything-V2 configuration")
        out_channels = list(out_channels)
        self.use_clstoken = use_clstoken
        self.projects = nn.ModuleList([nn.Conv2d(in_channels, oc, kernel_size=1, stride=1, padding=0) for oc in out_channels])
        self.resize_layers = nn.ModuleList([
            nn.ConvTranspose2d(out_channels[0], out_channels[0], kernel_size=4, stride=4, padding=0),
            nn.ConvTranspose2d(out_channels[1], out_channels[1], kernel_size=2, stride=2, padding=0),
            nn.Identity(),
            nn.Conv2d(out_channels[3], out_channels[3], kernel_size=3, stride=2, padding=1)])
        self.scratch = _make_scratch(out_channels, features, groups=1, expand=False)
        self.scratch.stem_transpose = None
        self.scratch.refinenet1 = _make_fusion_block(features, use_bn)
        self.scratch.refinenet2 = _make_fusion_block(features, use_bn)
        self.scratch.refinenet3 = _make_fusion_block(features, use_bn)
        self.scratch.refinenet4 = _make_fusion_block(features, use_bn)
        head_features_2 = 32
        self.scratch.output_conv1 = nn.Conv2d(features, features // 2, kernel_size=3, stride=1, padding=1)
        self.scratch.output_conv2 = nn.Sequential(
            nn.Conv2d(features // 2, head_features_2, kernel_size=3, stride=1, padding=1), nn.ReLU(True),
            nn.Conv2d(head_features_2, 1, kernel_size=1, stride=1, padding=0), nn.Sigmoid())

    def _final_1x1(self, x):
        """32 -> 1 channel: the conv kernel's narrowest tile is 16 output channels, so the single filter sits in row 0 of a
        zero-padded 16-row weight; float32 output"""
        m = self.scratch.output_conv2[2]
        w = torch.zeros((16, m.weight.shape[1]), dtype=torch.bfloat16, device=x.device)
        w[0] = m.weight.detach().reshape(-1).to(torch.bfloat16)
        b = torch.zeros(16, dtype=torch.float32, device=x.device)
        b[0] = m.bias.detach().float()[0]
        return ops.conv_fwd_raw(x, w, 16, 1, 1, 0, bias=b, out_f32=True)[..., 0]

    def forward(self, out_features, patch_h, patch_w):
        out = []
        for i, x in enumerate(out_features):
            x = x[0]                                              # patch tokens (B, ph*pw, C); the class token is unused
            x = x.reshape(x.shape[0], patch_h, patch_w, x.shape[-1]).contiguous()
            x = conv(x, self.projects[i])
            r = self.resize_layers[i]
            if isinstance(r, nn.ConvTranspose2d):
                x = deconv_ks(x, r)
            elif isinstance(r, nn.Conv2d):
                x = conv(x, r)
            out.append(x)
        l1, l2, l3, l4 = out
        s = self.scratch
        l1, l2, l3, l4 = conv(l1, s.layer1_rn), conv(l2, s.layer2_rn), conv(l3, s.layer3_rn), conv(l4, s.layer4_rn)
        p4 = s.refinenet4(l4, size=l3.shape[1:3])
        p3 = s.refinenet3(p4, l3, size=l2.shape[1:3])
        p2 = s.refinenet2(p3, l2, size=l1.shape[1:3])
        p1 = s.refinenet1(p2, l1)
        o = conv(p1, s.output_conv1)
        o = ops.resize_bilinear_ac(o, (int(patch_h * 14), int(patch_w * 14)))
        o = conv(o, s.output_conv2[0], relu=True)
        return torch.sigmoid(self._final_1x1(o)).unsqueeze(1)     # (B,1,H,W) float32


class DepthAnythingV2(nn.Module):
    """dpt.py:159-222.  forward: (B,3,H,W) float image (ImageNet-normalised RGB, H and W multiples of 14) on the GPU ->
    (B,H,W) float32 metric depth.  Image reading / resizing of `infer_image` (cv2) stays with the caller."""

    intermediate_layer_idx = {'vits': [2, 5, 8, 11], 'vitb': [2, 5, 8, 11], 'vitl': [4, 11, 17, 23]}

    def __init__(self, encoder='vitl', features=256, out_channels=(256, 512, 1024, 1024), use_bn=False, use_clstoken=False,
                 max_depth=20.0):
        super().__init__()
        self.max_depth = max_depth
        self.encoder = encoder
        self.pretrained = DINOv2(model_name=encoder)
        self.depth_head = DPTHead(self.pretrained.embed_dim, features, use_bn, out_channels=out_channels,
                                  use_clstoken=use_clstoken)
        self.channels_last_()
        self.register_load_state_dict_post_hook(lambda module, incompatible: ops.bump_weight_epoch())

    def channels_last_(self):
        """conv weights in the kernels' [Cout][kh][kw][Cin] order (same logical shape, so checkpoints load unchanged)"""
        for m in self.depth_head.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        return self

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.channels_last_()
        ops.bump_weight_epoch()
        return out

    @torch.no_grad()
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("3dod_amd.depth_anything_v2 runs on the GPU only (no CPU path)")
        patch_h, patch_w = x.shape[-2] // 14, x.shape[-1] // 14
        feats = self.pretrained.get_intermediate_layers(x, self.intermediate_layer_idx[self.encoder], return_class_token=True)
        depth = self.depth_head(feats, patch_h, patch_w) * self.max_depth
        return depth.squeeze(1)


    # ------------------------------------------------------------------ raw images (dpt.py:191-222)
    @staticmethod
    def _net_size(h, w, input_size=518, multiple=14):
        """util/transform.py Resize(resize_method='lower_bound', keep_aspect_ratio=True, ensure_multiple_of=14): scale so
        that both sides are at least `input_size`, then round each side to a multiple of 14 (never below input_size)"""
        import numpy as np
        scale = max(input_size / h, input_size / w)

        def constrain(x):
            y = int(np.round(x / multiple) * multiple)
            if y < input_size:
                y = int(np.ceil(x / multiple) * multiple)
            return y
        return constrain(scale * h), constrain(scale * w)

    def image2tensor(self, raw_image, input_size=518):
        """raw_image: (H,W,3) uint8 BGR (numpy or tensor) -> ((1,3,h',w') normalised RGB on the model's device, (H,W)).
        The reference resizes with cv2.INTER_CUBIC on the host; here the resize is torch's bicubic on the GPU
        [cv2 absent: parity unpinned for the resampling]."""
        import torch.nn.functional as F
        dev = next(self.parameters()).device
        img = torch.as_tensor(raw_image).to(dev)
        h, w = img.shape[:2]
        x = img.flip(-1).permute(2, 0, 1).float()[None] / 255.0
        nh, nw = self._net_size(h, w, input_size)
        x = F.interpolate(x, (nh, nw), mode="bicubic", align_corners=False)
        mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)
        return (x - mean) / std, (h, w)

    @torch.no_grad()
    def infer_image(self, raw_image, input_size=518):
        """(H,W,3) uint8 BGR -> (H,W) float32 numpy depth in metres, like the reference's infer_image"""
        import torch.nn.functional as F
        image, (h, w) = self.image2tensor(raw_image, input_size)
        depth = self.forward(image)
        depth = F.interpolate(depth[:, None], (h, w), mode="bilinear", align_corners=True)[0, 0]
        return depth.cpu().numpy()
