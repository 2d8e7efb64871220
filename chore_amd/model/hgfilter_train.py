"""Differentiable forward of the stacked-hourglass encoder (training path).

Inference runs the whole encoder as one launch program with aggressive fusion and buffer reuse
(HGFilter.forward -> chore_encode_fwd).  Training needs every intermediate tensor and a backward per layer, so here
the same network (/root/reference/model/HGFilters.py:144-185, HourGlass :26-50, ConvBlock net_util.py:374-396) is
composed from autograd nodes whose forward AND backward are HIP kernels (chore_amd/ops.py):
  every GroupNorm -> ReLU -> conv3x3 / conv1x1 layer   = ops.conv_gn   (150 of the 151 convolutions)
  bn_end -> ReLU                                        = ops.gn_relu
  up1 + bicubic_up2(low3)                               = ops.upadd     (gather-form transpose in the backward)
  7x7 stem, 2x2 average pooling                         = ops.stem, ops.avgpool2
Only the glue that moves no FLOPs -- concat and residual adds -- is torch; tensors are NHWC throughout and no library
convolution / GEMM is on the path (results are bit-reproducible run to run).
"""
import os

import torch

from .. import ops


def _nchw(x):          # NHWC tensor -> (B,C,H,W) channels-last view
    return x.permute(0, 3, 1, 2)


def conv_block(m, x, x_stats=None):
    """ConvBlock.forward (net_util.py:374-396) -> (y, GroupNorm statistics of y): one fused operator (csrc/convblock.hip);
    CHORE_TRAIN_LAYERWISE=1 composes it from per-layer nodes instead (the older path, kept for A/B runs)"""
    if not os.environ.get("CHORE_TRAIN_LAYERWISE"):
        return ops.conv_block(x, m, x_stats)
    return _conv_block_layerwise(m, x), None


def _conv_block_layerwise(m, x):
    sx = ops.gn_stats(x)                  # bn1 and bn4 normalise the same tensor: one statistics pass
    o1, s1 = ops.conv_gn(x, m.conv1.weight, None, m.bn1.weight, m.bn1.bias, x_stats=sx, want_stats=True)
    o2, s2 = ops.conv_gn(o1, m.conv2.weight, None, m.bn2.weight, m.bn2.bias, x_stats=s1, want_stats=True)
    o3 = ops.conv_gn(o2, m.conv3.weight, None, m.bn3.weight, m.bn3.bias, x_stats=s2)
    out = torch.cat((o1, o2, o3), dim=3)
    res = x if m.downsample is None else ops.conv_gn(x, m.downsample[2].weight, None, m.bn4.weight, m.bn4.bias, x_stats=sx)
    return out + res


def hourglass(m, level, x, xs=None):
    """HourGlass._forward (HGFilters.py:26-50); xs: statistics of x when its producer made them"""
    up1, _ = conv_block(getattr(m, f"b1_{level}"), x, xs)
    low1, sl = ops.avgpool2(x, True)              # the pooling / upsampling kernels produce their output's statistics too
    low1, s1 = conv_block(getattr(m, f"b2_{level}"), low1, sl)
    low2, s2 = hourglass(m, level - 1, low1, s1) if level > 1 else conv_block(getattr(m, f"b2_plus_{level}"), low1, s1)
    low3, _ = conv_block(getattr(m, f"b3_{level}"), low2, s2)
    return ops.upadd(up1, low3, True)


def forward_train(enc, images, tdt, on_stack=None, x3=False, cuts=None):
    """enc: chore_amd.model.hgfilter.HGFilter (parameter tree); images (B,C,H,W) fp32; tdt: activation dtype; x3 (with fp32
    activations): the convolutions and their gradients on the fp16 matrix cores with split operands (ops.x3_convs).
    Returns (outputs, tmpx, normx) as (B,C,H,W) channels-last views like HGFilter.forward; outputs carry grad.
    on_stack(i, output_i, tmpx): called as soon as stack i's output exists (CHORE.forward launches that stack's field query and
    loss on a second stream from it, so they run beside the next stack's encoder -- forward and backward).
    cuts (a list, filled here): the tensor that enters each stack -- all that stack i and everything before it share; what
    chore_amd.parallel.backward_in_segments cuts the backward at."""
    with ops.zero_arena(images.device), ops.x3_convs(x3 and tdt == torch.float32):
        return _forward_train(enc, images, tdt, on_stack, cuts)


def _forward_train(enc, images, tdt, on_stack=None, cuts=None):
    x = ops.stem(images, enc.conv1.weight, enc.conv1.bias, tdt)
    x = ops.gn_relu(x, enc.bn1.weight, enc.bn1.bias)
    tmpx = x
    x, sn = ops.avgpool2(conv_block(enc.conv2, x)[0], True)
    normx = x
    x, sx = conv_block(enc.conv3, x, sn)
    previous, sp = conv_block(enc.conv4, x, sx)
    outputs = []
    n = enc.num_modules
    for i in range(n):
        if cuts is not None:
            cuts.append(previous)
        hg, sh = hourglass(getattr(enc, f"m{i}"), enc.opt.num_hourglass, previous, sp)
        sp = None                  # `previous` is re-formed by torch adds below: its statistics are recomputed
        ll, _ = conv_block(getattr(enc, f"top_m_{i}"), hg, sh)
        cl, be = getattr(enc, f"conv_last{i}"), getattr(enc, f"bn_end{i}")
        ll, sl = ops.conv_gn(ll, cl.weight, cl.bias, want_stats=True)
        ll = ops.gn_relu(ll, be.weight, be.bias, x_stats=sl)
        li = getattr(enc, f"l{i}")
        tmp_out = ops.conv_gn(ll, li.weight, li.bias)
        outputs.append(tmp_out)
        if on_stack is not None:
            on_stack(i, _nchw(tmp_out), _nchw(tmpx.detach()))
        if i < n - 1:
            bl, al = getattr(enc, f"bl{i}"), getattr(enc, f"al{i}")
            previous = previous + ops.conv_gn(ll, bl.weight, bl.bias) + ops.conv_gn(tmp_out, al.weight, al.bias)
    return [_nchw(o) for o in outputs], _nchw(tmpx.detach()), _nchw(normx)
