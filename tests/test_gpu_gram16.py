"""GPU tests of the fp16-split Gram kernel (csrc/gram16.hip): G = Z'Z of an fp32 panel - what stands behind `svd!(Z)`
(src/robustPCA.jl:194) and the two opnorm calls (:177, :225) for panels of more than 2048 columns - on the fp16 MFMA from two
fp16 planes and three products, against float64 on the same fp32 data and against the fp32-MFMA kernel it replaces."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    torch.zeros(1, device="cuda")
    return torch


@pytest.fixture(scope="module")
def eng(torch_mod):
    import tlsq_amd
    e = tlsq_amd.Engine(0)
    yield e
    e.close()


def _gram(eng, torch, dZ, M, N):
    G = torch.full((N, N), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    assert eng.lib.tlsq_k_gram_f32(eng.h, dZ.data_ptr(), M, N, M, G.data_ptr(), N, 1) == 0, eng.lib.tlsq_last_error(eng.h)
    eng.synchronize()
    return G


@pytest.mark.parametrize("M,N,kind", [
    (8192, 1024, "lowrank"),        # the smallest shape the kernel takes
    (16384, 2304, "scaled"),        # columns spread over e^(+-2): entries down to 1e-4 of the largest
    (65536, 4096, "lowrank"),       # BASELINE config 5
    (4096, 1152, "tiny"),           # 1e-25: the scale is a large power of two
    (4096, 1152, "huge"),           # 1e+25
])
def test_gram_fp16_split(eng, torch_mod, M, N, kind):
    import tlsq_amd
    torch = torch_mod
    g = torch.Generator(device="cuda").manual_seed(M + N)
    f32 = dict(device="cuda", dtype=torch.float32, generator=g)
    if kind == "scaled":
        Zt = torch.randn(N, M, **f32) * torch.exp(2.0 * torch.randn(N, 1, **f32)) + 0.5
    else:
        Zt = torch.randn(N, 16, **f32) @ torch.randn(16, M, **f32) + 0.3 * torch.randn(N, M, **f32)
        Zt = Zt + 4.0 * torch.randn(N, M, **f32) * (torch.rand(N, M, **f32) < 0.02)
    if kind == "tiny":
        Zt = Zt * 1e-25
    if kind == "huge":
        Zt = Zt * 1e25
    Zt = Zt.contiguous()                                   # (N x M row-major = M x N column-major)
    Zd = Zt.double()
    ref = Zd @ Zd.T
    d = torch.sqrt(torch.diagonal(ref))
    scale = d[:, None] * d[None, :]
    G = _gram(eng, torch, Zt, M, N)
    with tlsq_amd.dev_switches(GRAM_H3=0):
        G32 = _gram(eng, torch, Zt, M, N)
    assert torch.equal(G, G.T)                             # one writer per entry pair
    err = float(((G - ref).abs() / scale).max())
    err32 = float(((G32 - ref).abs() / scale).max())
    # entry-wise against sqrt(G_ii G_jj): the two-plane split keeps 22 bits of every entry, the fp32 partial sums of 64 rows are
    # folded in fp64 - the same order of magnitude as the fp32-MFMA kernel (whose products are exact), far below plain fp32
    assert err < 2e-7, (err, err32)
    if np.isfinite(err32):                                 # (1e25: the squares overflow the fp32-MFMA kernel's partial sums, not the scaled planes)
        assert err < 8 * err32 + 2e-8, (err, err32)
    # the quantity the solver takes from it: sigma_max^2 to 1e-7
    lam = float(torch.linalg.eigvalsh(G)[-1])
    lam_ref = float(torch.linalg.eigvalsh(ref)[-1])
    assert abs(lam - lam_ref) <= 1e-7 * lam_ref


def test_gram_fp16_split_zero_and_nonfinite(eng, torch_mod):
    torch = torch_mod
    M, N = 4096, 1024
    Z = torch.zeros(N, M, dtype=torch.float32, device="cuda")
    assert float(_gram(eng, torch, Z, M, N).abs().max()) == 0.0
    Z[5, 7] = float("nan")
    G = _gram(eng, torch, Z, M, N)
    assert bool(torch.isnan(G[5, 5])) and float(G[6, 6]) == 0.0
