"""Oracle: ContentLossVGG (CPU, plain torch).  TEST INFRASTRUCTURE ONLY.

Restates reference loss.py:11-70: ImageNet normalisation (mean .485/.456/.406, std .229/.224/.225), torchvision VGG19
``features[0..35]`` (16 conv3x3+ReLU, 4 MaxPool2d(2) before index 36), taps at features.17 / .26 / .35 (ReLU outputs
relu3_4 / relu4_4 / relu5_4) weighted 1/8, 1/4, 1/2 (config.py:60-64), criterion MSE.
PARITY UNPINNED: the reference fetches torchvision's IMAGENET1K_V1 weights from the network (loss.py:46); torchvision
and the weights are both unavailable, so this restatement cannot be checked against reference outputs.  It follows
torchvision's published VGG19 layer list; weights are seeded random (torchvision's own VGG init: kaiming_normal
fan_out / zero bias) unless a state dict with keys ``features.N.weight|bias`` is supplied."""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn

VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def layer_list(upto=35):
    """[(index, kind, cin, cout)] of torchvision vgg19().features up to and including `upto`."""
    out, idx, cin = [], 0, 3
    for v in VGG19_CFG:
        if v == "M":
            out.append((idx, "pool", cin, cin))
            idx += 1
        else:
            out.append((idx, "conv", cin, v))
            out.append((idx + 1, "relu", v, v))
            idx += 2
            cin = v
    return [l for l in out if l[0] <= upto]


def init_vgg_state(seed=0, upto=35):
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for idx, kind, cin, cout in layer_list(upto):
        if kind == "conv":
            w = torch.empty(cout, cin, 3, 3)
            std = (2.0 / (cout * 9)) ** 0.5                   # kaiming_normal_(mode="fan_out", nonlinearity="relu")
            sd[f"features.{idx}.weight"] = w.normal_(0, std, generator=g)
            sd[f"features.{idx}.bias"] = torch.zeros(cout)
    return sd


def vgg_features(sd, x, taps=(17, 26, 35)):
    mean = torch.tensor(MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(STD, dtype=x.dtype).view(1, 3, 1, 1)
    h = (x - mean) / std
    feats = {}
    for idx, kind, cin, cout in layer_list(max(taps)):
        if kind == "conv":
            h = F.conv2d(h, sd[f"features.{idx}.weight"].to(x.dtype), sd[f"features.{idx}.bias"].to(x.dtype), 1, 1)
        elif kind == "relu":
            h = F.relu(h)
        else:
            h = F.max_pool2d(h, 2)
        if idx in taps:
            feats[idx] = h
    return feats


def content_loss(sd, x, gt, layers=None):
    layers = layers or {"features.17": 1 / 8, "features.26": 1 / 4, "features.35": 1 / 2}
    taps = tuple(int(k.split(".")[1]) for k in layers)
    fx, fg = vgg_features(sd, x, taps), vgg_features(sd, gt, taps)
    loss = x.new_zeros(())
    for name, w in layers.items():
        i = int(name.split(".")[1])
        loss = loss + w * F.mse_loss(fx[i], fg[i])
    return loss
