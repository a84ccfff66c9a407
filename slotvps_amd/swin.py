"""Swin Transformer backbone (hierarchical shifted-window attention) for the Swin-L Slot-VPS configuration
(configs/cityscapes/swinL_fpn_slotvps.py; reference module mmdet/models/backbones/swin_transformer.py:449-631).
PyTorch-ROCm like the other backbones (north star), inference only, with the reference's parameter names
(patch_embed.proj / norm, layers.{i}.blocks.{j}.{norm1, attn.{relative_position_bias_table, qkv, proj}, norm2,
mlp.{fc1, fc2}}, layers.{i}.downsample.{norm, reduction}, norm{i}) so its checkpoints load unchanged.

Written for the fused attention kernel: one `scaled_dot_product_attention` call per block over all windows,
with the relative-position bias and the shifted-window mask folded into one additive bias [nW, heads, N, N]
that is built once per (block, feature size) and cached - instead of an explicit softmax(q k^T + bias + mask) v.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .registry import BACKBONES


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


def _relative_position_index(ws):
    """[ws*ws, ws*ws] index into the (2ws-1)^2 bias table for every (query, key) pair of a window."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)   # [2, N]
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0) + (ws - 1)                         # [N, N, 2] >= 0
    return rel[:, :, 0] * (2 * ws - 1) + rel[:, :, 1]


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        self.register_buffer("relative_position_index", _relative_position_index(window_size))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)

    def bias(self, shift_mask):
        """Additive attention bias [1 or nW, heads, N, N]: relative-position bias (+ the 0 / -100 shift mask)."""
        N = self.window_size ** 2
        b = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(N, N, -1).permute(2, 0, 1)
        return b.unsqueeze(0) if shift_mask is None else b.unsqueeze(0) + shift_mask.unsqueeze(1)

    def forward(self, x, bias):
        """x [B * nW, N, C] windows (window index fastest inside a batch element), bias from `bias()`."""
        Bn, N, C = x.shape
        nW = bias.shape[0]
        qkv = self.qkv(x).view(Bn // nW, nW, N, 3, self.num_heads, C // self.num_heads)
        q, k, v = (qkv[:, :, :, i].transpose(2, 3) for i in range(3))                  # [B, nW, heads, N, hd] views
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=bias.unsqueeze(0), scale=self.scale)
        return self.proj(o.transpose(2, 3).reshape(Bn, N, C))


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True, qk_scale=None):
        super().__init__()
        assert 0 <= shift_size < window_size, "shift_size must in 0-window_size"
        self.window_size, self.shift_size = window_size, shift_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size, num_heads, qkv_bias, qk_scale)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self._bias_cache = {}

    def _bias(self, Hp, Wp, device):
        key = (Hp, Wp, str(device), self.attn.relative_position_bias_table._version)
        if key not in self._bias_cache:
            mask = None
            if self.shift_size > 0:
                ws, ss = self.window_size, self.shift_size
                img = torch.zeros((Hp, Wp), device=device)
                cnt = 0
                for hs in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):          # 9 regions of the shifted image
                    for wsl in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
                        img[hs, wsl] = cnt
                        cnt += 1
                win = img.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)   # [nW, N]
                diff = win.unsqueeze(1) - win.unsqueeze(2)
                mask = torch.zeros_like(diff).masked_fill(diff != 0, -100.0)           # tokens of different regions do not mix
            self._bias_cache = {key: self.attn.bias(mask).contiguous()}
        return self._bias_cache[key]

    def forward(self, x, H, W):
        B, L, C = x.shape
        ws = self.window_size
        shortcut = x
        x = self.norm1(x).view(B, H, W, C)
        pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
        if pad_r or pad_b:
            x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))                                   # after the norm, zeros (as the reference)
        Hp, Wp = H + pad_b, W + pad_r
        if self.shift_size > 0:
            x = torch.roll(x, shifts=(-self.shift_size, -self.shift_size), dims=(1, 2))
        xw = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
        xw = self.attn(xw, self._bias(Hp, Wp, x.device))
        x = xw.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
        if self.shift_size > 0:
            x = torch.roll(x, shifts=(self.shift_size, self.shift_size), dims=(1, 2))
        if pad_r or pad_b:
            x = x[:, :H, :W, :]
        x = shortcut + x.reshape(B, H * W, C)                                          # drop_path is the identity at inference
        return x + self.mlp(self.norm2(x))


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x, H, W):
        B, L, C = x.shape
        x = x.view(B, H, W, C)
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
        return self.reduction(self.norm(x.view(B, -1, 4 * C)))


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio, qkv_bias, qk_scale, downsample):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio, qkv_bias, qk_scale)
            for i in range(depth)])
        self.downsample = PatchMerging(dim) if downsample else None

    def forward(self, x, H, W):
        for blk in self.blocks:
            x = blk(x, H, W)
        if self.downsample is None:
            return x, H, W, x, H, W
        return x, H, W, self.downsample(x, H, W), (H + 1) // 2, (W + 1) // 2


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, patch_norm=True):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dim) if patch_norm else None

    def forward(self, x):
        _, _, H, W = x.shape
        ph, pw = self.patch_size
        if W % pw or H % ph:
            x = F.pad(x, (0, (pw - W % pw) % pw, 0, (ph - H % ph) % ph))
        x = self.proj(x)
        if self.norm is not None:
            B, C, Wh, Ww = x.shape
            x = self.norm(x.flatten(2).transpose(1, 2)).transpose(1, 2).reshape(B, C, Wh, Ww)
        return x


@BACKBONES.register_module
class SwinTransformer(nn.Module):
    def __init__(self, pretrain_img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0.2, norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 out_indices=(0, 1, 2, 3), frozen_stages=-1, use_checkpoint=False):
        super().__init__()
        if norm_layer is not nn.LayerNorm:
            raise NotImplementedError("norm_layer other than LayerNorm")
        self.num_layers = len(depths)
        self.embed_dim, self.ape, self.out_indices, self.frozen_stages = embed_dim, ape, tuple(out_indices), frozen_stages
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim, patch_norm)
        if ape:
            n = pretrain_img_size if isinstance(pretrain_img_size, (tuple, list)) else (pretrain_img_size, pretrain_img_size)
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, embed_dim, n[0] // patch_size, n[1] // patch_size))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=.02)
        # drop_rate / attn_drop_rate / drop_path_rate only matter in training (the reference ships no training code)
        self.layers = nn.ModuleList([
            BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i], window_size, mlp_ratio, qkv_bias, qk_scale,
                       downsample=i < self.num_layers - 1) for i in range(self.num_layers)])
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in self.out_indices:
            self.add_module(f"norm{i}", nn.LayerNorm(self.num_features[i]))

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str):
            raise NotImplementedError("load a checkpoint with load_state_dict; there is no model zoo access here")
        for m in self.modules():                    # swin_transformer.py:582-590
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

    def forward(self, x):
        x = self.patch_embed(x)
        Wh, Ww = x.shape[2], x.shape[3]
        if self.ape:
            x = x + F.interpolate(self.absolute_pos_embed, size=(Wh, Ww), mode="bicubic")
        x = x.flatten(2).transpose(1, 2)
        outs = []
        for i, layer in enumerate(self.layers):
            x_out, H, W, x, Wh, Ww = layer(x, Wh, Ww)
            if i in self.out_indices:
                x_out = getattr(self, f"norm{i}")(x_out)
                outs.append(x_out.view(-1, H, W, self.num_features[i]).permute(0, 3, 1, 2).contiguous())
        return tuple(outs)
