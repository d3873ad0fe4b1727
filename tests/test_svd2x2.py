"""The device-side closed form of the region predictor's 2x2 SVD (cvpr23_lfdm_amd.lfae_predictors.svd2x2_sym_lapack)
against the call the reference makes: torch.svd on the host = LAPACK xGESDD (LFAE/modules/region_predictor.py:16-25).
U is sign-ambiguous mathematically; the reference uses U * sqrt(S), so LAPACK's convention is part of the contract."""
import torch

from cvpr23_lfdm_amd.lfae_predictors import svd2x2_sym_lapack


def _cov(t):
    return torch.stack((torch.stack((t[:, 0], t[:, 1]), -1), torch.stack((t[:, 1], t[:, 2]), -1)), -2)


def _affine(u, s):
    return u @ torch.diag_embed(s.sqrt())


def test_random_covariances_match_lapack_signs():
    g = torch.Generator().manual_seed(1)
    for scale, n in [(0.1, 200000), (1.0, 100000), (1e-3, 100000), (30.0, 50000)]:
        m = torch.randn(n, 2, 2, generator=g) * scale
        cov = m @ m.transpose(1, 2)
        u, s, _ = torch.svd(cov)
        uu, ss = svd2x2_sym_lapack(cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1])
        assert float((uu - u).abs().max()) < 5e-5, "sign convention or value of U differs from LAPACK"
        assert float(((ss - s).abs() / s[:, :1]).max()) < 2e-6


def test_edge_cases_match_lapack():
    cases = []
    vals = [1e-8, 1e-4, 0.01, 0.5, 1.0, 3.0]
    for a in vals:
        for c in vals:
            for b in [0.0, 1e-9, -1e-9, 1e-5, -1e-5, 1e-3, -1e-3]:       # diagonal, negligible and small off-diagonals
                if b * b <= a * c:
                    cases.append((a, b, c))
    for x, y in [(1.0, 0.5), (0.3, -0.7), (1e-3, 2e-3), (1.0, 1e-4), (-0.4, -0.4), (0.6, 0.6)]:   # rank one and nearly
        cases.append((x * x, x * y, y * y))
        cases.append((x * x + 1e-7, x * y, y * y + 1e-7))
    for b in [0.1, -0.1, 0.5, -0.5, 1e-4, 1e-7, -1e-7, 3e-8]:            # equal diagonal
        cases.append((0.5, b, 0.5))
    t = torch.tensor(cases, dtype=torch.float32)
    u, s, _ = torch.svd(_cov(t))
    uu, ss = svd2x2_sym_lapack(t[:, 0], t[:, 1], t[:, 2])
    assert not torch.isnan(uu).any() and not torch.isnan(ss).any()
    ref, got = _affine(u, s), _affine(uu, ss)          # what the predictor uses; blind to U columns of a zero singular value
    err = (got - ref).abs().amax(dim=(1, 2)) / (ref.abs().amax(dim=(1, 2)) + 1e-20)
    assert float(err.max()) < 1e-3, cases[int(err.argmax())]        # rank-one inputs: sqrt of a rounding-noise singular value


def _edge_cases():
    cases = []
    vals = [1e-8, 1e-4, 0.01, 0.5, 1.0, 3.0]
    for a in vals:
        for c in vals:
            for b in [0.0, 1e-9, -1e-9, 1e-5, -1e-5, 1e-3, -1e-3]:
                if b * b <= a * c:
                    cases.append((a, b, c))
    for x, y in [(1.0, 0.5), (0.3, -0.7), (1e-3, 2e-3), (1.0, 1e-4), (-0.4, -0.4), (0.6, 0.6)]:
        cases.append((x * x, x * y, y * y))
        cases.append((x * x + 1e-7, x * y, y * y + 1e-7))
    for b in [0.1, -0.1, 0.5, -0.5, 1e-4, 1e-7, -1e-7, 3e-8]:
        cases.append((0.5, b, 0.5))
    return torch.tensor(cases, dtype=torch.float32)


def test_library_kernel_is_the_same_closed_form(backend):
    """lfdm_svd2x2_sym_f32 (the scalar port inside lfdm_lfae_region_stats_f32) takes the same branches as the element-wise torch
    formulation above - on random covariances at four scales and on the edge cases - and therefore LAPACK's signs."""
    from cvpr23_lfdm_amd import ops
    dev = backend
    g = torch.Generator().manual_seed(2)
    batches = [_edge_cases()]
    for scale, n in [(0.1, 20000), (1.0, 20000), (1e-3, 20000), (30.0, 20000)]:
        m = torch.randn(n, 2, 2, generator=g) * scale
        cov = m @ m.transpose(1, 2)
        batches.append(torch.stack((cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]), -1))
    for t in batches:
        uu, ss = svd2x2_sym_lapack(t[:, 0], t[:, 1], t[:, 2])
        u, s = ops.svd2x2_sym(t[:, 0].to(dev), t[:, 1].to(dev), t[:, 2].to(dev))
        assert not torch.isnan(u).any() and not torch.isnan(s).any()
        ref, got = _affine(uu, ss), _affine(u.cpu(), s.cpu())
        err = (got - ref).abs().amax(dim=(1, 2)) / (ref.abs().amax(dim=(1, 2)) + 1e-20)
        # (same bars as against LAPACK above: near-equal diagonals amplify the last-bit differences between the device's and the
        #  host's divisions / square roots by 1 / gap)
        assert float(err.max()) < (1e-3 if t is batches[0] else 5e-5), (float(err.max()), t[int(err.argmax())])
