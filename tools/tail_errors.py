#!/usr/bin/env python3
"""Dev tool: where does the conv3.bias / conv3.weight gradient error of the HIP path come from at the bench size?  Compares the
pre-clamp output, the clamp mask, the pixel-loss gradient and the bias gradient of the HIP generator graph against the fp64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "srgan-st_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from conftest import rel_err
from oracle import model as om
from srganst import gen_graph, ops
from srganst.config import Config
from srganst.model import Generator

cfg = Config()
torch.manual_seed(41)
from srganst.model import Discriminator
D = Discriminator(cfg)      # same RNG stream as grad_errors_iter.py
G = Generator(cfg)
g0 = {k: v.clone() for k, v in G.state_dict().items()}
gen = torch.Generator().manual_seed(42)
gt, lr = torch.rand(16, 3, 96, 96, generator=gen), torch.rand(16, 3, 24, 24, generator=gen)

cap = {}
_clamp = torch.clamp
def spy(x, *a, **k):
    cap["pre"] = x
    return _clamp(x, *a, **k)

def oracle(dtype):
    sd = {k: (v.to(dtype) if v.is_floating_point() else v).clone() for k, v in g0.items()}
    for k in om.param_keys(sd):
        sd[k].requires_grad_(True)
    torch.clamp = spy
    try:
        sr = om.generator_forward(sd, lr.to(dtype), True, {})
    finally:
        torch.clamp = _clamp
    pre = cap["pre"]
    pre.retain_grad()
    loss = torch.nn.functional.mse_loss(sr, gt.to(dtype))
    loss.backward()
    return sr.detach(), pre.detach(), pre.grad.detach(), {k: sd[k].grad for k in om.param_keys(sd)}

sr64, pre64, gpre64, g64 = oracle(torch.float64)
sr32, pre32, gpre32, g32 = oracle(torch.float32)
G.cuda().train()
params = [p.detach() for p in G.parameters()]
names = [n for n, _ in G.named_parameters()]
with torch.no_grad():
    sr, sv = gen_graph.forward(G, lr.cuda(), params, True)
    u, slope, sr_pre = sv["last"]
    n = sr.numel()
    dsr = (2.0 / n) * (sr - gt.cuda())
    grads, _ = gen_graph.backward(G, params, sv, dsr)
    torch.cuda.synchronize()
hp = sr_pre.cpu().double()
print("pre-clamp output: hip vs fp64", rel_err(hp, pre64), " oracle fp32 vs fp64", rel_err(pre32.double(), pre64))
print("   mean signed error hip", float((hp - pre64).mean()), " oracle fp32", float((pre32.double() - pre64).mean()),
      " mean |error| hip", float((hp - pre64).abs().mean()), " oracle fp32", float((pre32.double() - pre64).abs().mean()))
m64 = (pre64 >= 0) & (pre64 <= 1)
mh = (hp >= 0) & (hp <= 1)
m32 = (pre32 >= 0) & (pre32 <= 1)
print("mask flips vs fp64: hip", int((mh != m64).sum()), " oracle fp32", int((m32 != m64).sum()), " of", m64.numel(), " unmasked", int(m64.sum()))
dh = dsr.cpu().double()
true_dsr = (2.0 / n) * (sr64 - gt.double())
print("dsr: hip vs fp64", rel_err(dh, true_dsr))
bias_true = g64["conv3.bias"]
bias_hip = grads[names.index("conv3.bias")].cpu().double()
bias_from_hip_inputs = (dh * mh).sum((0, 2, 3))
print("conv3.bias grad: fp64", bias_true.tolist())
print("   hip", bias_hip.tolist(), " rel", rel_err(bias_hip, bias_true))
print("   fp64 sum of hip's own masked dsr", bias_from_hip_inputs.tolist(), " rel vs truth", rel_err(bias_from_hip_inputs, bias_true),
      " hip kernel vs that", rel_err(bias_hip, bias_from_hip_inputs))
print("   oracle fp32 rel", rel_err(g32["conv3.bias"].double(), bias_true))
print("   sum |terms| per channel", (true_dsr * m64).abs().sum((0, 2, 3)).tolist())
for k in ("conv3.weight", "upsampling.1.upsample_block.0.bias", "upsampling.1.upsample_block.2.weight"):
    print(k, " hip", rel_err(grads[names.index(k)].cpu().double(), g64[k]), " oracle fp32", rel_err(g32[k].double(), g64[k]))
