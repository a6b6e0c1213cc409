#!/usr/bin/env python3
"""Golden fixture for BestBuddyLoss FROM THE REFERENCE (loss.py:78-142): inputs, loss, d(loss)/d(sr), the selected candidate
per patch and the margin to the runner-up (so that tests can tell genuine mismatches from fp32 near-ties).
Build container only (needs /root/reference).  Re-run:  python tests/golden/make_golden_bb.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import import_reference, lowfreq, save  # noqa: E402


def main():
    _, _, _, rutils, rloss = import_reference()
    gen = torch.Generator().manual_seed(4321)
    arrs = {}
    for name, (B, H) in {"lf48": (2, 48), "lf96": (1, 96)}.items():
        gt = lowfreq(gen, B, H)
        x = (gt + 0.08 * torch.randn(gt.shape, generator=gen)).clamp(0, 1).requires_grad_(True)
        for crit in ("l1", "l2"):
            mod = rloss.BestBuddyLoss(criterion=crit)
            loss = mod(x, gt)
            (gx,) = torch.autograd.grad(loss, x)
            arrs[f"{name}/{crit}/loss"] = loss.detach().numpy()
            arrs[f"{name}/{crit}/grad"] = gx.numpy()
        # matching details from the reference's own helper (same calls as loss.py:116-134)
        import torch.nn.functional as F
        with torch.no_grad():
            unf = lambda t: F.unfold(t, kernel_size=3, padding=0, stride=3).permute(0, 2, 1).contiguous()
            p1, p2 = unf(x), unf(gt)
            gt2 = F.interpolate(gt, scale_factor=0.5, mode="bicubic", align_corners=False)
            gt4 = F.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False)
            cat = torch.cat([p2, unf(gt2), unf(gt4)], 1)
            score = rutils.batch_pairwise_distance(p1, cat, "l2") + rutils.batch_pairwise_distance(p2, cat, "l2")
            top2 = torch.topk(score, 2, dim=2, largest=False)
        arrs[f"{name}/x"], arrs[f"{name}/gt"] = x.detach().numpy(), gt.numpy()
        arrs[f"{name}/gt2"], arrs[f"{name}/gt4"] = gt2.numpy(), gt4.numpy()
        arrs[f"{name}/ind"] = top2.indices[..., 0].numpy().astype(np.int32)
        arrs[f"{name}/margin"] = (top2.values[..., 1] - top2.values[..., 0]).numpy()
        # GramLoss (loss.py:145-228) on the same pair
        for crit in ("l1", "l2"):
            gm = rloss.GramLoss(criterion=crit)
            gl = gm(x, gt)
            (ggx,) = torch.autograd.grad(gl, x)
            arrs[f"{name}/gram/{crit}/loss"] = gl.detach().numpy()
            arrs[f"{name}/gram/{crit}/grad"] = ggx.numpy()
        with torch.no_grad():
            gm = rloss.GramLoss()
            q1, q2 = gm.compute_patches(x), gm.compute_patches(gt)
            qcat = torch.cat([q2, gm.compute_patches(gt2), gm.compute_patches(gt4)], 1)
            gs = rutils.batch_pairwise_distance(q1, qcat, "l2") + rutils.batch_pairwise_distance(q2, qcat, "l2")
            gtop = torch.topk(gs, 2, dim=2, largest=False)
        arrs[f"{name}/gram/ind"] = gtop.indices[..., 0].numpy().astype(np.int32)
        arrs[f"{name}/gram/margin"] = (gtop.values[..., 1] - gtop.values[..., 0]).numpy()
        arrs[f"{name}/gram/p1"] = q1.numpy()
        # PatchwiseStructureTensorLoss (loss.py:292-375) on the same pair
        for crit in ("l1", "l2"):
            pm = rloss.PatchwiseStructureTensorLoss(criterion=crit)
            pl = pm(x, gt)
            (pgx,) = torch.autograd.grad(pl, x)
            arrs[f"{name}/pst/{crit}/loss"] = pl.detach().numpy()
            arrs[f"{name}/pst/{crit}/grad"] = pgx.numpy()
        with torch.no_grad():
            pm = rloss.PatchwiseStructureTensorLoss()
            s1, s2 = pm.compute_patches(x), pm.compute_patches(gt)
            scat = torch.cat([s2, pm.compute_patches(gt2), pm.compute_patches(gt4)], 1)
            ss = rutils.batch_pairwise_distance(s1, scat, "l2") + rutils.batch_pairwise_distance(s2, scat, "l2")
            stop = torch.topk(ss, 2, dim=2, largest=False)
        arrs[f"{name}/pst/ind"] = stop.indices[..., 0].numpy().astype(np.int32)
        arrs[f"{name}/pst/margin"] = (stop.values[..., 1] - stop.values[..., 0]).numpy()
        arrs[f"{name}/pst/p1"] = s1.numpy()
    save("bestbuddy", **arrs)


if __name__ == "__main__":
    main()
