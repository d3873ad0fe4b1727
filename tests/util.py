import json
import os

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


def record_margin(what, err, scale, tol):
    """LFDM_PARITY_LOG=<file>: every comparison appends its ACHIEVED error (one JSON object per line) - the pass/fail bars alone
    do not say how much of a tolerance a kernel change has used up (tools/parity_margins.py folds the log into profiles/)."""
    path = os.environ.get("LFDM_PARITY_LOG")
    if not path:
        return
    bar = tol * max(1.0, scale)
    with open(path, "a") as f:
        f.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], "what": what, "max_abs_err": err,
                            "ref_scale": scale, "tol": tol, "fraction_of_bar": err / bar if bar > 0 else None}) + "\n")


def assert_close(a, b, tol, what=""):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = float((a - b).abs().max())
    scale = float(b.abs().max()) + 1e-12
    record_margin(what, err, scale, tol)
    assert err <= tol * max(1.0, scale), "%s: max abs err %.3e (ref scale %.3e, tol %.1e)" % (what, err, scale, tol)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale
