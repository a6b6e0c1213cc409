#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, CPU torch).  The reference
never travels: what is committed is data (inputs + the reference's outputs) and this
script.  Re-run:  python tests/golden/make_golden.py

How the reference is imported:
  * config.py, model.py, bicubic.py import unmodified.
  * utils.py / loss.py have module-level imports of cv2 and torchvision, which are not
    installed and cannot be.  None of the hot-path functions use them, except
    StructureTensorLoss.st_loss -> torchvision.transforms.Grayscale.  Inert placeholder
    modules satisfy the import statements; Grayscale is restated from torchvision's
    documented ITU-R 601 weights (0.2989, 0.587, 0.114), so the RGB->gray step of the
    ST-loss fixtures is pinned by documentation, everything downstream (gray -> loss)
    by the reference's own code.  Fixtures ``st_ops`` use gray inputs and are pure
    reference.
  * utils.get_gaussian_kernel hard-codes .cuda() (utils.py:206,208); Tensor.cuda is
    made the identity for this process because there is no GPU here.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _placeholder(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def import_reference():
    sys.path.insert(0, REF)
    _placeholder("cv2")
    tv = _placeholder("torchvision")
    tv.utils = _placeholder("torchvision.utils")
    tv.utils.make_grid = None
    tv.models = _placeholder("torchvision.models")
    fe = _placeholder("torchvision.models.feature_extraction")
    fe.create_feature_extractor = None
    tv.models.feature_extraction = fe
    tv.transforms = _placeholder("torchvision.transforms")

    class Grayscale:  # torchvision.transforms.Grayscale, documented weights
        def __call__(self, x):
            r, g, b = x.unbind(dim=-3)
            return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)

    tv.transforms.Grayscale = Grayscale
    tv.transforms.Normalize = None
    torch.Tensor.cuda = lambda self, *a, **k: self
    import config as rconfig
    import model as rmodel
    import bicubic as rbicubic
    import utils as rutils
    import loss as rloss
    return rconfig, rmodel, rbicubic, rutils, rloss


def npd(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


BIG = 150_000


def save(name, **arrs):
    """Arrays above BIG elements are stored as their first 4 rows (key#head4) plus the
    float64 L2 norm and sum of the whole array (key#norm, key#sum): the fixtures stay small;
    the full tensors are reproducible from the recorded seeds (checked against the heads)."""
    small = {}
    for k, v in arrs.items():
        v = np.asarray(v)
        if v.size > BIG and v.dtype.kind == "f" and k.split("/")[0] not in ("x", "gt", "lr", "sr"):
            small[k + "#head4"] = v[:4].copy()
            small[k + "#norm"] = np.array(np.sqrt((v.astype(np.float64) ** 2).sum()))
            small[k + "#sum"] = np.array(v.astype(np.float64).sum())
        else:
            small[k] = v
    arrs = small
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}.npz  {os.path.getsize(path)/1024:.1f} KB  ({len(arrs)} arrays)")


def lowfreq(gen, B, H):
    """Low-frequency image batch: bicubic-upsampled 12x12 noise + small noise, on the 1/255 grid."""
    base = torch.rand(B, 3, 12, 12, generator=gen)
    im = torch.nn.functional.interpolate(base, size=(H, H), mode="bicubic", align_corners=False)
    im = im + 0.05 * torch.randn(B, 3, H, H, generator=gen)
    return torch.round(im.clamp(0, 1) * 255) / 255


def main():
    rconfig, rmodel, rbicubic, rutils, rloss = import_reference()
    torch.set_num_threads(8)
    gen = torch.Generator().manual_seed(1234)

    # ------------------------------------------------------------------ 1. ST building blocks (pure reference)
    g5, dg5 = rutils.get_gaussian_kernel(0.5, also_dg=True)
    k17 = rutils.get_gaussian_kernel(2.0)
    gray = torch.rand(1, 32, 32, generator=gen)
    gray2 = lowfreq(gen, 1, 32)[0, :1]
    S1 = rutils.structure_tensor(gray, sigma=0.5, rho=2.0)
    S2 = rutils.structure_tensor(gray2, sigma=0.5, rho=2.0)
    N1 = rutils.normalize(S1)
    M = rutils.compute_invS1xS2(S1, S2, True)
    L = rutils.compute_eigenvalues(M)
    d = rutils.compute_distance(L)
    save("st_ops", g5=g5.numpy(), dg5=dg5.numpy(), k17=k17.numpy(), gray1=gray.numpy(), gray2=gray2.numpy(),
         S1=S1.numpy(), S2=S2.numpy(), N1=N1.numpy(), M=M.numpy(), L=L.numpy(), d=d.numpy())

    # ------------------------------------------------------------------ 2. full ST loss fwd + d/dx
    crit = rloss.StructureTensorLoss()
    cases = {}
    u8 = torch.randint(0, 256, (2, 3, 32, 32), generator=gen, dtype=torch.uint8)
    u8b = torch.randint(0, 256, (2, 3, 32, 32), generator=gen, dtype=torch.uint8)
    cases["noise32"] = (u8.float() / 255, u8b.float() / 255)
    lf = lowfreq(gen, 2, 32)
    cases["lowfreq32"] = ((lf + 0.03 * torch.randn(lf.shape, generator=gen)).clamp(0, 1), lf)
    flat = torch.full((2, 3, 32, 32), 0.5)
    flat2 = flat.clone()
    flat2[:, :, 8:24, 8:24] = lowfreq(gen, 2, 16)
    cases["flat32"] = (flat2, flat)                 # constant patches: pins the epsilon paths
    a96 = torch.randint(0, 256, (2, 3, 96, 96), generator=gen, dtype=torch.uint8)
    lf96 = lowfreq(gen, 2, 96)
    cases["mixed96"] = ((lf96 + 0.1 * (a96.float() / 255 - 0.5)).clamp(0, 1), lf96)
    arrs = {}
    for name, (x, gt) in cases.items():
        x = x.clone().requires_grad_(True)
        loss = crit(x, gt)
        (gx,) = torch.autograd.grad(loss, x)
        # fp64 "truth" through the same reference code
        # (default dtype fp64 so that the reference builds fp64 filter taps too)
        torch.set_default_dtype(torch.float64)
        x64 = x.detach().double().requires_grad_(True)
        loss64 = crit(x64, gt.double())
        (gx64,) = torch.autograd.grad(loss64, x64)
        torch.set_default_dtype(torch.float32)
        arrs[name + "_x"] = x.detach().numpy()
        arrs[name + "_gt"] = gt.numpy()
        arrs[name + "_loss"] = loss.detach().numpy()
        arrs[name + "_grad"] = gx.numpy()
        arrs[name + "_loss64"] = loss64.detach().numpy()
        arrs[name + "_grad64"] = gx64.numpy().astype(np.float32)
        print(f"  st {name}: loss={loss.item():.6f} loss64={loss64.item():.6f} |g|={gx.norm():.4e}")
    save("st_loss", **arrs)

    # ------------------------------------------------------------------ 3. reduced generator: fwd, grads, two Adam steps
    cfg = rconfig.Config()
    cfg.MODEL.G_N_CHANNEL = 8
    cfg.MODEL.G_N_RCB = 2
    torch.manual_seed(0)
    G = rmodel.Generator(cfg)
    G.train()
    # perturb BN affine / PReLU away from their trivial init so gradients are informative
    with torch.no_grad():
        for n, p in G.named_parameters():
            if p.dim() == 1:        # conv biases, BN gamma/beta, PReLU slopes
                p.add_(0.1 * torch.randn(p.shape, generator=gen))
    state0 = {k: v.clone() for k, v in G.state_dict().items()}
    gt = lowfreq(gen, 2, 32)
    lr = rbicubic.Bicubic()(gt, scale=0.25)
    opt = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-4)
    mse = torch.nn.MSELoss()
    out = {"gt": gt.numpy(), "lr": lr.numpy()}
    for step in range(2):
        G.zero_grad()
        sr = G(lr)
        l_pix = mse(sr, gt) * 1.0
        l_st = crit(sr, gt) * (1 / 3)
        (l_pix + l_st).backward()
        if step == 0:
            out["sr0"] = sr.detach().numpy()
            out["loss_pixel0"] = l_pix.detach().numpy()
            out["loss_st0"] = l_st.detach().numpy()
            for n, p in G.named_parameters():
                out["grad0/" + n] = p.grad.numpy().copy()
        opt.step()
        for k, v in G.state_dict().items():
            out[f"state{step+1}/" + k] = v.numpy().copy()
    for k, v in state0.items():
        out["state0/" + k] = v.numpy()
    save("g_small_step", **out)

    # ------------------------------------------------------------------ 4. full-size generator, seed-pinned
    cfg.MODEL.G_N_CHANNEL = 64
    cfg.MODEL.G_N_RCB = 16
    torch.manual_seed(0)
    Gf = rmodel.Generator(cfg)
    Gf.train()
    gtf = lowfreq(gen, 2, 96)
    lrf = rbicubic.Bicubic()(gtf, scale=0.25)
    srf = Gf(lrf)
    lossf = mse(srf, gtf)
    lossf.backward()
    out = {"gt": gtf.numpy(), "lr": lrf.numpy(), "sr": srf.detach().numpy(), "loss": lossf.detach().numpy()}
    sd = Gf.state_dict()
    # weights are reproducible from seed 0; store a few for a direct RNG-order check + all grad norms
    for k in ["conv1.0.weight", "trunk.0.rcb.0.weight", "trunk.15.rcb.3.weight", "conv3.weight"]:
        out["w/" + k] = sd[k].numpy()
    out["grad_names"] = np.array([n for n, _ in Gf.named_parameters()])
    out["grad_norms"] = np.array([p.grad.norm().item() for _, p in Gf.named_parameters()], dtype=np.float64)
    for k in ["conv1.0.weight", "trunk.0.rcb.0.weight", "trunk.7.rcb.1.weight", "trunk.15.rcb.3.weight",
              "upsampling.1.upsample_block.0.bias", "conv3.weight", "conv1.1.weight"]:
        out["g/" + k] = dict(Gf.named_parameters())[k].grad.numpy()
    out["bn/trunk.0.rcb.1.running_mean"] = sd["trunk.0.rcb.1.running_mean"].numpy()
    out["bn/trunk.0.rcb.1.running_var"] = sd["trunk.0.rcb.1.running_var"].numpy()
    out["n_params"] = np.array(sum(p.numel() for p in Gf.parameters()))
    save("g_full_seed0", **out)

    # ------------------------------------------------------------------ 5. reduced discriminator + one full train iteration
    cfg.MODEL.G_N_CHANNEL = 8
    cfg.MODEL.G_N_RCB = 2
    cfg.MODEL.D_N_CHANNEL = 4
    torch.manual_seed(0)
    D = rmodel.Discriminator(cfg)
    G = rmodel.Generator(cfg)
    D.train()
    G.train()
    gt = lowfreq(gen, 4, 96)
    lr = rbicubic.Bicubic()(gt, scale=0.25)
    out = {"gt": gt.numpy(), "lr": lr.numpy()}
    for k, v in D.state_dict().items():
        out["d_state0/" + k] = v.numpy().copy()
    for k, v in G.state_dict().items():
        out["g_state0/" + k] = v.numpy().copy()
    d_opt = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-4)
    g_opt = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-4)
    bce = torch.nn.BCEWithLogitsLoss()
    real = torch.full([4, 1], 0.9)
    fake = torch.zeros([4, 1])
    crits = {"Adversarial": (bce, 0.001), "Pixel": (mse, 1.0), "ST": (crit, 1 / 3)}
    # --- train.py:125-144
    for p in D.parameters():
        p.requires_grad = False
    G.zero_grad()
    sr = G(lr)
    g_loss = torch.tensor(0.0)
    for name, (c, w) in crits.items():
        l = c(D(sr), real) if name == "Adversarial" else c(sr, gt)
        g_loss = g_loss + l * w
        out["g_loss/" + name] = (l * w).detach().numpy()
    g_loss.backward()
    for n, p in G.named_parameters():
        out["g_grad/" + n] = p.grad.numpy().copy()
    g_opt.step()
    for k, v in D.state_dict().items():
        if "running" in k or "tracked" in k:
            out["d_bn_after_g/" + k] = v.numpy().copy()
    # --- train.py:149-164
    for p in D.parameters():
        p.requires_grad = True
    D.zero_grad()
    pred_gt = D(gt)
    loss_real = bce(pred_gt, real)
    pred_sr = D(sr.detach().clone())
    loss_fake = bce(pred_sr, fake)
    d_loss = loss_real + loss_fake
    d_loss.backward()
    for n, p in D.named_parameters():
        out["d_grad/" + n] = p.grad.numpy().copy()
    d_opt.step()
    out["sr"] = sr.detach().numpy()
    out["pred_gt"] = pred_gt.detach().numpy()
    out["pred_sr"] = pred_sr.detach().numpy()
    out["d_loss"] = d_loss.detach().numpy()
    for k, v in D.state_dict().items():
        out["d_state1/" + k] = v.numpy().copy()
    for k, v in G.state_dict().items():
        out["g_state1/" + k] = v.numpy().copy()
    save("gan_small_iter", **out)

    # ------------------------------------------------------------------ 6. full-size discriminator, seed-pinned
    cfg.MODEL.D_N_CHANNEL = 64
    torch.manual_seed(0)
    Df = rmodel.Discriminator(cfg)
    Df.train()
    x = lowfreq(gen, 2, 96).requires_grad_(True)
    logit = Df(x)
    l = bce(logit, torch.full([2, 1], 0.9))
    l.backward()
    out = {"x": x.detach().numpy(), "logit": logit.detach().numpy(), "loss": l.detach().numpy(),
           "dx": x.grad.numpy(),
           "grad_names": np.array([n for n, _ in Df.named_parameters()]),
           "grad_norms": np.array([p.grad.norm().item() for _, p in Df.named_parameters()], dtype=np.float64),
           "n_params": np.array(sum(p.numel() for p in Df.parameters()))}
    for k in ["features.0.weight", "features.2.weight", "features.20.weight", "classifier.2.weight"]:
        out["g/" + k] = dict(Df.named_parameters())[k].grad.numpy()
    out["w/features.0.weight"] = Df.state_dict()["features.0.weight"].numpy()
    out["w/classifier.2.weight"] = Df.state_dict()["classifier.2.weight"].numpy()
    save("d_full_seed0", **out)

    # ------------------------------------------------------------------ 7. bicubic x1/4 (dataset.py:27-28)
    hr = torch.randint(0, 256, (1, 3, 96, 96), generator=gen, dtype=torch.uint8)
    step = torch.zeros(1, 3, 96, 96)
    step[..., 48:] = 1.0
    bic = rbicubic.Bicubic()
    save("bicubic", hr_u8=hr.numpy(), lr=bic(hr.float() / 255, scale=0.25).numpy(),
         step_lr=bic(step, scale=0.25).numpy())

    # ------------------------------------------------------------------ 8. metrics helpers that do not need cv2 (utils.py:62-102,132-154)
    img = torch.rand(1, 3, 24, 20, generator=gen)
    t2i = rutils.tensor2img(img)
    f = t2i.astype(np.float32) / 255.0
    y = rutils.bgr2ycbcr(f.copy(), only_y=True)
    t2i_b = rutils.tensor2img((img + 0.05 * torch.randn(img.shape, generator=gen)).clamp(0, 1))
    yb = rutils.bgr2ycbcr(t2i_b.astype(np.float32) / 255.0, only_y=True)
    save("metrics", img=img.numpy(), tensor2img=t2i, y=y, psnr=np.array(rutils.PSNR(y * 255, yb * 255)),
         img_b_u8=t2i_b, yb=yb)


if __name__ == "__main__":
    main()
