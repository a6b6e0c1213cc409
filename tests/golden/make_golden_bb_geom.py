#!/usr/bin/env python3
"""Golden fixture for the patch losses OFF the reference's default configuration, FROM THE REFERENCE: BestBuddyLoss with other patch
geometries (loss.py:86: ksize / pad / stride) and both matching distances (utils.py:157-191 dist_norm 'l1' / 'l2'), GramLoss and
PatchwiseStructureTensorLoss with dist_norm 'l1'.  Inputs, loss, d(loss)/d(sr), the selected candidate per patch and the margin to the
runner-up.  Build container only (needs /root/reference).  Re-run:  python tests/golden/make_golden_bb_geom.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import import_reference, lowfreq, save  # noqa: E402

# name -> (B, H, ksize, pad, stride, dist_norm, criterion)
BB_CASES = {
    "k3p1s2_l2": (2, 24, 3, 1, 2, "l2", "l1"),       # overlapping patches with padding
    "k5p0s5_l1": (2, 40, 5, 0, 5, "l1", "l2"),       # 5 x 5 patches, L1 matching distance, L2 criterion
    "k3p0s3_l1": (2, 48, 3, 0, 3, "l1", "l1"),       # the default geometry with the L1 matching distance
    "k4p2s3_l2": (1, 36, 4, 2, 3, "l2", "l2"),       # even patch size, stride < ksize, pad 2
}


def main():
    _, _, _, rutils, rloss = import_reference()
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(9753)
    arrs = {}
    for name, (B, H, k, pad, stride, dn, crit) in BB_CASES.items():
        gt = lowfreq(gen, B, H)
        x = (gt + 0.08 * torch.randn(gt.shape, generator=gen)).clamp(0, 1).requires_grad_(True)
        mod = rloss.BestBuddyLoss(ksize=k, pad=pad, stride=stride, dist_norm=dn, criterion=crit)
        loss = mod(x, gt)
        (gx,) = torch.autograd.grad(loss, x)
        with torch.no_grad():
            unf = lambda t: F.unfold(t, kernel_size=k, padding=pad, stride=stride).permute(0, 2, 1).contiguous()
            p1, p2 = unf(x), unf(gt)
            gt2 = F.interpolate(gt, scale_factor=0.5, mode="bicubic", align_corners=False)
            gt4 = F.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False)
            cat = torch.cat([p2, unf(gt2), unf(gt4)], 1)
            score = rutils.batch_pairwise_distance(p1, cat, dn) + rutils.batch_pairwise_distance(p2, cat, dn)
            top2 = torch.topk(score, 2, dim=2, largest=False)
        arrs[f"bb/{name}/x"], arrs[f"bb/{name}/gt"] = x.detach().numpy(), gt.numpy()
        arrs[f"bb/{name}/loss"], arrs[f"bb/{name}/grad"] = loss.detach().numpy(), gx.numpy()
        arrs[f"bb/{name}/ind"] = top2.indices[..., 0].numpy().astype(np.int32)
        arrs[f"bb/{name}/margin"] = (top2.values[..., 1] - top2.values[..., 0]).numpy()
        arrs[f"bb/{name}/score_scale"] = top2.values[..., 0].abs().mean().numpy()
    # GramLoss / PatchwiseStructureTensorLoss with the L1 matching distance
    gt = lowfreq(gen, 2, 48)
    x = (gt + 0.08 * torch.randn(gt.shape, generator=gen)).clamp(0, 1).requires_grad_(True)
    arrs["l1/x"], arrs["l1/gt"] = x.detach().numpy(), gt.numpy()
    for tag, cls in (("gram", rloss.GramLoss), ("pst", rloss.PatchwiseStructureTensorLoss)):
        for crit in ("l1", "l2"):
            m = cls(dist_norm="l1", criterion=crit)
            l = m(x, gt)
            (g,) = torch.autograd.grad(l, x)
            arrs[f"l1/{tag}/{crit}/loss"], arrs[f"l1/{tag}/{crit}/grad"] = l.detach().numpy(), g.numpy()
        with torch.no_grad():
            m = cls(dist_norm="l1")
            gt2 = F.interpolate(gt, scale_factor=0.5, mode="bicubic", align_corners=False)
            gt4 = F.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False)
            q1, q2 = m.compute_patches(x), m.compute_patches(gt)
            qcat = torch.cat([q2, m.compute_patches(gt2), m.compute_patches(gt4)], 1)
            sc = rutils.batch_pairwise_distance(q1, qcat, "l1") + rutils.batch_pairwise_distance(q2, qcat, "l1")
            top2 = torch.topk(sc, 2, dim=2, largest=False)
        arrs[f"l1/{tag}/ind"] = top2.indices[..., 0].numpy().astype(np.int32)
        arrs[f"l1/{tag}/margin"] = (top2.values[..., 1] - top2.values[..., 0]).numpy()
    save("bestbuddy_geom", **arrs)


if __name__ == "__main__":
    main()
