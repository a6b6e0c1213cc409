"""CPU restatement of the reference's BestBuddyLoss (loss.py:78-142, utils.py:157-191) - TEST INFRASTRUCTURE ONLY.

Best-buddy matching: the SR image and the GT image are cut into non-overlapping k x k patches (unfold, channel-major
C*k*k vectors); the candidate set is the GT patches at full, 1/2 and 1/4 resolution (torch bicubic, align_corners=False,
no antialiasing); every SR patch i is paired with the candidate j minimising
    alpha * ||p_sr[i] - c[j]||^2 + beta * ||p_gt[i] - c[j]||^2      (squared L2 via  |x|^2 + |y|^2 - 2 x.y, clamped at 0)
and the loss is the L1 (or L2) criterion between the SR patches and their buddies.  Only the final criterion is
differentiated (argmin indices carry no gradient)."""
import torch
import torch.nn.functional as F


def pairwise_sq_l2(x, y):
    """utils.py:173-187: dist[b,i,j] = |x_i|^2 + |y_j|^2 - 2 x_i.y_j, clamped to [0, inf)."""
    xn = (x ** 2).sum(dim=2).unsqueeze(2)
    yn = (y ** 2).sum(dim=2).unsqueeze(1)
    return torch.clamp(xn + yn - 2.0 * torch.bmm(x, y.transpose(1, 2)), 0.0)


def pairwise_l1(x, y):
    return (x.unsqueeze(2) - y.unsqueeze(1)).abs().sum(dim=3)


def patches(img, ksize=3, pad=0, stride=3):
    """loss.py:116-118: F.unfold -> [B, n_patches, C*k*k]."""
    return F.unfold(img, kernel_size=ksize, padding=pad, stride=stride).permute(0, 2, 1).contiguous()


def candidates(gt, ksize=3, pad=0, stride=3):
    """loss.py:119-129: GT patches at scales 1, 1/2, 1/4 concatenated."""
    gt2 = F.interpolate(gt, scale_factor=0.5, mode="bicubic", align_corners=False)
    gt4 = F.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False)
    return torch.cat([patches(gt, ksize, pad, stride), patches(gt2, ksize, pad, stride), patches(gt4, ksize, pad, stride)], 1), gt2, gt4


def best_buddy_loss(x, gt, alpha=1.0, beta=1.0, ksize=3, pad=0, stride=3, dist_norm="l2", criterion="l1"):
    """-> (loss, ind [B, n_patches], score [B, n_patches, n_candidates])."""
    p1 = patches(x, ksize, pad, stride)
    p2 = patches(gt, ksize, pad, stride)
    cat, _, _ = candidates(gt, ksize, pad, stride)
    dist = pairwise_sq_l2 if dist_norm == "l2" else pairwise_l1
    score = alpha * dist(p1, cat) + beta * dist(p2, cat)
    ind = torch.min(score, dim=2)[1]
    sel = torch.gather(cat, 1, ind.unsqueeze(-1).expand(-1, -1, p1.shape[2]))
    loss = F.l1_loss(p1, sel) if criterion == "l1" else F.mse_loss(p1, sel)
    return loss, ind, score


# ---- GramLoss (reference loss.py:145-228): the same matching on the 3x3 gram matrix of every 3x3 patch
def gram_patches(img, ksize=3):
    """loss.py:182-201: non-overlapping k x k patches of a [B,3,H,W] image -> F = patch as [3, k*k]; G = F F^T / (3*k*k)
    flattened row-major -> [B, n_patches, 9] (k = 3)."""
    B = img.shape[0]
    p = F.unfold(img, kernel_size=ksize, stride=ksize).permute(0, 2, 1)          # [B, nP, 3*k*k] in (c, ky, kx) order
    Fm = p.reshape(B, -1, 3, ksize * ksize)
    G = torch.matmul(Fm, Fm.transpose(2, 3)) / float(3 * ksize * ksize)
    return G.reshape(B, -1, 9)


def gram_loss(x, gt, alpha=1.0, beta=1.0, ksize=3, dist_norm="l2", criterion="l1"):
    p1, p2 = gram_patches(x, ksize), gram_patches(gt, ksize)
    gt2 = F.interpolate(gt, scale_factor=0.5, mode="bicubic", align_corners=False)
    gt4 = F.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False)
    cat = torch.cat([p2, gram_patches(gt2, ksize), gram_patches(gt4, ksize)], 1)
    dist = pairwise_sq_l2 if dist_norm == "l2" else pairwise_l1
    score = alpha * dist(p1, cat) + beta * dist(p2, cat)
    ind = torch.min(score, dim=2)[1]
    sel = torch.gather(cat, 1, ind.unsqueeze(-1).expand(-1, -1, p1.shape[2]))
    loss = F.l1_loss(p1, sel) if criterion == "l1" else F.mse_loss(p1, sel)
    return loss, ind, score


# ---- PatchwiseStructureTensorLoss (reference loss.py:292-375): the same matching on the normalised structure tensor of every
# 3x3 patch (gray -> structure_tensor(sigma, rho) with zero 'same' padding inside the patch -> normalize; 3 x 9 = 27 features)
def st_patches(img, sigma=0.5, rho=2.0, ksize=3):
    from . import st as ost
    B = img.shape[0]
    p = F.unfold(img, kernel_size=ksize, stride=ksize).permute(0, 2, 1).reshape(-1, 3, ksize, ksize)     # [B*nP, 3, k, k]
    S = ost.structure_tensor(ost.grayscale(p), sigma, rho)                                              # [B*nP, 3, k, k]
    return ost.normalize(S).reshape(B, -1, 3 * ksize * ksize)


def patchwise_st_loss(x, gt, sigma=0.5, rho=2.0, alpha=1.0, beta=1.0, ksize=3, dist_norm="l2", criterion="l1"):
    p1, p2 = st_patches(x, sigma, rho, ksize), st_patches(gt, sigma, rho, ksize)
    gt2 = F.interpolate(gt, scale_factor=0.5, mode="bicubic", align_corners=False)
    gt4 = F.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False)
    cat = torch.cat([p2, st_patches(gt2, sigma, rho, ksize), st_patches(gt4, sigma, rho, ksize)], 1)
    dist = pairwise_sq_l2 if dist_norm == "l2" else pairwise_l1
    score = alpha * dist(p1, cat) + beta * dist(p2, cat)
    ind = torch.min(score, dim=2)[1]
    sel = torch.gather(cat, 1, ind.unsqueeze(-1).expand(-1, -1, p1.shape[2]))
    loss = F.l1_loss(p1, sel) if criterion == "l1" else F.mse_loss(p1, sel)
    return loss, ind, score
