"""Oracle: SRResNet generator / VGG-style discriminator (CPU, plain torch).
TEST INFRASTRUCTURE ONLY.

Functional restatement over a *state dict* whose keys are the reference's
(``conv1.0.weight``, ``trunk.{i}.rcb.{0..4}.*``, ``conv2.{0,1}.*``,
``upsampling.{j}.upsample_block.{0,2}.*``, ``conv3.*``, ``features.{n}.*``,
``classifier.{0,2}.*``):

  * generator_forward       reference model.py:138-152 (+ 155-184 for the blocks)
  * discriminator_forward   reference model.py:67-71 (+ 30-65 for the layer list)
  * init_generator_state / init_discriminator_state
                            reference model.py:79-136 / 15-65: same layer
                            construction order and the same Kaiming-normal re-init
                            loop, so the torch CPU RNG is consumed identically
                            ("identical seeds" => identical parameters).

Train-mode BatchNorm follows torch.nn.BatchNorm2d defaults: eps 1e-5, momentum 0.1,
biased variance for normalisation, unbiased for running_var, num_batches_tracked += 1.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------- init
def _g_skeleton(in_ch=3, out_ch=3, ch=64, n_rcb=16, upscale=4) -> nn.Module:
    """A bare container reproducing the construction order of model.py:100-127."""
    m = nn.Module()
    m.conv1 = nn.Sequential(nn.Conv2d(in_ch, ch, 9, 1, 4), nn.PReLU())
    blocks = []
    for _ in range(n_rcb):
        b = nn.Module()
        b.rcb = nn.Sequential(nn.Conv2d(ch, ch, 3, 1, 1, bias=False), nn.BatchNorm2d(ch), nn.PReLU(),
                              nn.Conv2d(ch, ch, 3, 1, 1, bias=False), nn.BatchNorm2d(ch))
        blocks.append(b)
    m.trunk = nn.Sequential(*blocks)
    m.conv2 = nn.Sequential(nn.Conv2d(ch, ch, 3, 1, 1, bias=False), nn.BatchNorm2d(ch))
    ups = []
    if upscale in (2, 4, 8):
        for _ in range(int(math.log(upscale, 2))):
            u = nn.Module()
            u.upsample_block = nn.Sequential(nn.Conv2d(ch, ch * 4, 3, 1, 1), nn.PixelShuffle(2), nn.PReLU())
            ups.append(u)
    else:
        raise NotImplementedError("only x2/x4/x8 are in scope (model.py:119-124)")
    m.upsampling = nn.Sequential(*ups)
    m.conv3 = nn.Conv2d(ch, out_ch, 9, 1, 4)
    for mod in m.modules():                       # model.py:130-136
        if isinstance(mod, nn.Conv2d):
            nn.init.kaiming_normal_(mod.weight)
            if mod.bias is not None:
                nn.init.constant_(mod.bias, 0)
        elif isinstance(mod, nn.BatchNorm2d):
            nn.init.constant_(mod.weight, 1)
    return m


def init_generator_state(in_ch=3, out_ch=3, ch=64, n_rcb=16, upscale=4) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k, v.detach().clone()) for k, v in
                       _g_skeleton(in_ch, out_ch, ch, n_rcb, upscale).state_dict().items())


D_PLAN = [  # (conv idx in features, bn idx or None, cin mult, cout mult, stride)   model.py:30-59
    (0, None, 0, 1, 1), (2, 3, 1, 1, 2), (5, 6, 1, 2, 1), (8, 9, 2, 2, 2),
    (11, 12, 2, 4, 1), (14, 15, 4, 4, 2), (17, 18, 4, 8, 1), (20, 21, 8, 8, 2),
]


def _d_skeleton(in_ch=3, ch=64, out_ch=1, image_size=96) -> nn.Module:
    m = nn.Module()
    layers = []
    for ci, bi, cm, om, s in D_PLAN:
        cin = in_ch if cm == 0 else cm * ch
        layers.append(nn.Conv2d(cin, om * ch, 3, s, 1, bias=(bi is None)))
        if bi is not None:
            layers.append(nn.BatchNorm2d(om * ch))
        layers.append(nn.LeakyReLU(0.2, True))
    m.features = nn.Sequential(*layers)
    fs = image_size // 16
    m.classifier = nn.Sequential(nn.Linear(8 * ch * fs * fs, 1024), nn.LeakyReLU(0.2, True),
                                 nn.Linear(1024, out_ch))
    return m


def init_discriminator_state(in_ch=3, ch=64, out_ch=1, image_size=96):
    return OrderedDict((k, v.detach().clone()) for k, v in
                       _d_skeleton(in_ch, ch, out_ch, image_size).state_dict().items())


# ----------------------------------------------------------------------------- forward
def _bn(x, sd, pre, training, new_buffers):
    w, b = sd[pre + ".weight"], sd[pre + ".bias"]
    rm, rv = sd[pre + ".running_mean"], sd[pre + ".running_var"]
    if training:
        rm2, rv2 = rm.detach().clone(), rv.detach().clone()
        y = F.batch_norm(x, rm2, rv2, w, b, True, BN_MOMENTUM, BN_EPS)
        if new_buffers is not None:
            new_buffers[pre + ".running_mean"] = rm2
            new_buffers[pre + ".running_var"] = rv2
            new_buffers[pre + ".num_batches_tracked"] = sd[pre + ".num_batches_tracked"] + 1
        return y
    return F.batch_norm(x, rm, rv, w, b, False, BN_MOMENTUM, BN_EPS)


def generator_forward(sd, x, training=True, new_buffers=None, n_rcb=None, n_up=None):
    """model.py:142-152.  ``new_buffers`` (dict) receives the updated BN buffers."""
    if n_rcb is None:
        n_rcb = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("trunk."))
    if n_up is None:
        n_up = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("upsampling."))
    c1 = F.prelu(F.conv2d(x, sd["conv1.0.weight"], sd["conv1.0.bias"], 1, 4), sd["conv1.1.weight"])
    h = c1
    for i in range(n_rcb):
        p = f"trunk.{i}.rcb"
        t = F.conv2d(h, sd[p + ".0.weight"], None, 1, 1)
        t = _bn(t, sd, p + ".1", training, new_buffers)
        t = F.prelu(t, sd[p + ".2.weight"])
        t = F.conv2d(t, sd[p + ".3.weight"], None, 1, 1)
        t = _bn(t, sd, p + ".4", training, new_buffers)
        h = t + h
    h = F.conv2d(h, sd["conv2.0.weight"], None, 1, 1)
    h = _bn(h, sd, "conv2.1", training, new_buffers)
    h = h + c1
    for j in range(n_up):
        p = f"upsampling.{j}.upsample_block"
        h = F.conv2d(h, sd[p + ".0.weight"], sd[p + ".0.bias"], 1, 1)
        h = F.pixel_shuffle(h, 2)
        h = F.prelu(h, sd[p + ".2.weight"])
    h = F.conv2d(h, sd["conv3.weight"], sd["conv3.bias"], 1, 4)
    return torch.clamp(h, 0.0, 1.0)


def discriminator_forward(sd, x, training=True, new_buffers=None):
    """model.py:67-71 -> logits [B,1]."""
    h = x
    for ci, bi, cm, om, s in D_PLAN:
        h = F.conv2d(h, sd[f"features.{ci}.weight"], sd.get(f"features.{ci}.bias"), s, 1)
        if bi is not None:
            h = _bn(h, sd, f"features.{bi}", training, new_buffers)
        h = F.leaky_relu(h, 0.2)
    h = torch.flatten(h, 1)
    h = F.leaky_relu(F.linear(h, sd["classifier.0.weight"], sd["classifier.0.bias"]), 0.2)
    return F.linear(h, sd["classifier.2.weight"], sd["classifier.2.bias"])


def param_keys(sd):
    """Keys that are nn.Parameters (everything except BN buffers), in state-dict order."""
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]


# ---- ContentLossDiscriminator (reference loss.py:231-289): feature taps of the discriminator in eval mode
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def discriminator_features(sd, x, taps):
    """Outputs of the modules `features.<idx>` (idx in taps; LeakyReLU outputs) of the discriminator in eval mode
    (loss.py:264-277: create_feature_extractor(...).eval()) -> {idx: tensor}."""
    out, h = {}, x
    for ci, bi, cm, om, s in D_PLAN:
        h = F.conv2d(h, sd[f"features.{ci}.weight"], sd.get(f"features.{ci}.bias"), s, 1)
        act_idx = ci + 1
        if bi is not None:
            h = _bn(h, sd, f"features.{bi}", False, None)
            act_idx = bi + 1
        h = F.leaky_relu(h, 0.2)
        if act_idx in taps:
            out[act_idx] = h
        if act_idx >= max(taps):
            break
    return out


def disc_content_loss(sd, x, gt, layers, criterion="mse"):
    """loss.py:279-289: sum_layer weight * criterion(feat(normalize(x)), feat(normalize(gt)))."""
    mean = torch.tensor(IMAGENET_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=x.dtype).view(1, 3, 1, 1)
    taps = {int(k.split(".")[1]): float(v) for k, v in layers.items()}
    fx = discriminator_features(sd, (x - mean) / std, set(taps))
    fg = discriminator_features(sd, (gt - mean) / std, set(taps))
    crit = F.mse_loss if criterion in ("mse", "l2") else F.l1_loss
    loss = 0.0
    for t, w in taps.items():
        loss = loss + w * crit(fx[t], fg[t])
    return loss
