"""Oracle: one warm-up step / one SRGAN train step (CPU, plain torch).
TEST INFRASTRUCTURE ONLY.

  * OracleTrainer.warmup_step   reference warmup.py:74-96
  * OracleTrainer.train_step    reference train.py:116-164
        (G update with D frozen but in train() mode => D's BN running stats move in the
         G step too, train.py:110,136; then D update on gt and sr.detach() every
         D_UPDATE_INTERVAL-th batch, train.py:149-164)
  * Adam = torch.optim.Adam(lr 1e-4, betas (.9,.999), eps 1e-4, wd 0)  config.py:99-114
  * criterions: Pixel = MSE (config.py:88-90) or L1; ST = loss.py:380-413 x 1/3 (config.py:85);
    Adversarial = BCEWithLogits vs 0.9 x 1e-3 (config.py:24,72,78; train.py:113,136)
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import model as om
from . import st as ost


def pixel_loss(sr, gt, kind="mse"):
    return F.mse_loss(sr, gt) if kind == "mse" else F.l1_loss(sr, gt)


class OracleTrainer:
    """Holds G (and optionally D) state dicts as leaf tensors + their Adam optimizers."""

    def __init__(self, g_state, d_state=None, lr=1e-4, betas=(0.9, 0.999), eps=1e-4,
                 criterions=(("Pixel", 1.0), ("ST", 1.0 / 3.0)), pixel_kind="mse",
                 label_smoothing=0.1, d_update_interval=1, vgg=None):
        self.g = OrderedDict((k, v.detach().clone()) for k, v in g_state.items())
        self.g_keys = om.param_keys(self.g)
        for k in self.g_keys:
            self.g[k].requires_grad_(True)
        self.g_opt = torch.optim.Adam([self.g[k] for k in self.g_keys], lr=lr, betas=betas, eps=eps)
        self.d = None
        if d_state is not None:
            self.d = OrderedDict((k, v.detach().clone()) for k, v in d_state.items())
            self.d_keys = om.param_keys(self.d)
            for k in self.d_keys:
                self.d[k].requires_grad_(True)
            self.d_opt = torch.optim.Adam([self.d[k] for k in self.d_keys], lr=lr, betas=betas, eps=eps)
        self.criterions = list(criterions)
        self.pixel_kind = pixel_kind
        self.real = 1.0 - label_smoothing
        self.d_update_interval = d_update_interval
        self.vgg = vgg
        self.batch_num = 0

    # -- helpers
    def _commit(self, sd, new_buffers):
        for k, v in new_buffers.items():
            sd[k] = v.detach()

    def _g_losses(self, sr, gt):
        out = OrderedDict()
        for name, w in self.criterions:
            if name == "Pixel":
                l = pixel_loss(sr, gt, self.pixel_kind)
            elif name == "ST":
                l = ost.st_loss(sr, gt)
            elif name == "Adversarial":
                for k in self.d_keys:                      # train.py:125-126
                    self.d[k].requires_grad_(False)
                nb = {}
                logits = om.discriminator_forward(self.d, sr, True, nb)
                self._commit(self.d, nb)                   # BN side effect, train.py:110,136
                l = F.binary_cross_entropy_with_logits(logits, torch.full_like(logits, self.real))
            elif name == "ContentVGG":
                l = self.vgg(sr, gt)
            else:
                raise KeyError(name)
            out[name] = l * w
        return out

    # -- steps
    def warmup_step(self, gt, lr):
        self.g_opt.zero_grad(set_to_none=True)
        nb = {}
        sr = om.generator_forward(self.g, lr, True, nb)
        losses = self._g_losses(sr, gt)
        total = sum(losses.values())
        total.backward()
        self.g_opt.step()
        self._commit(self.g, nb)
        return sr.detach(), OrderedDict((k, v.detach()) for k, v in losses.items())

    def train_step(self, gt, lr):
        sr, losses = self.warmup_step(gt, lr)              # G update: train.py:125-144
        d_loss = None
        if self.batch_num % self.d_update_interval == 0:   # train.py:149-164
            for k in self.d_keys:
                self.d[k].requires_grad_(True)
            self.d_opt.zero_grad(set_to_none=True)
            nb = {}
            pred_gt = om.discriminator_forward(self.d, gt, True, nb)
            self._commit(self.d, nb)
            loss_real = F.binary_cross_entropy_with_logits(pred_gt, torch.full_like(pred_gt, self.real))
            nb = {}
            pred_sr = om.discriminator_forward(self.d, sr.detach().clone(), True, nb)
            self._commit(self.d, nb)
            loss_fake = F.binary_cross_entropy_with_logits(pred_sr, torch.zeros_like(pred_sr))
            d_loss = loss_real + loss_fake
            d_loss.backward()
            self.d_opt.step()
            d_loss = d_loss.detach()
        self.batch_num += 1
        return sr, losses, d_loss

    def g_grads(self):
        return OrderedDict((k, self.g[k].grad.detach().clone()) for k in self.g_keys)

    def d_grads(self):
        return OrderedDict((k, self.d[k].grad.detach().clone()) for k in self.d_keys)
