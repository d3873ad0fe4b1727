import torch


def to_cl(x):
    """(N, C, H, W) -> CL rows (N*H*W, C)."""
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous()


def from_cl(rows, n, h, w):
    return rows.reshape(n, h, w, rows.shape[1]).permute(0, 3, 1, 2).contiguous()


def unet_to_cl(x):
    """(B, C, T, H, W) -> (B*T*H*W, C)."""
    b, c, t, h, w = x.shape
    return x.permute(0, 2, 3, 4, 1).reshape(b * t * h * w, c).contiguous()


def unet_from_cl(rows, b, t, h, w):
    return rows.reshape(b, t, h, w, rows.shape[1]).permute(0, 4, 1, 2, 3).contiguous()


def rel_err(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def assert_close(a, b, tol, what=""):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = float((a - b).abs().max())
    scale = float(b.abs().max()) + 1e-12
    assert err <= tol * max(1.0, scale), "%s: max abs err %.3e (ref scale %.3e, tol %.1e)" % (what, err, scale, tol)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale
