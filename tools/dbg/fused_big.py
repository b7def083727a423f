import ctypes as C, sys
sys.path.insert(0, "/root/repo")
import torch, tlsq_amd
torch.zeros(1, device="cuda")
eng = tlsq_amd.Engine(0); lib, h = eng.lib, eng.h
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for (M, N, r) in [(131072, 256, 3), (131072, 256, 8), (262144, 256, 16), (524288, 256, 3), (1000000, 256, 3), (1000000, 256, 8), (200000, 512, 16)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda *s: torch.randn(s, dtype=torch.float64, device="cuda", generator=g)
    D, Y, Z = rnd(N, M), rnd(N, M), rnd(N, M)
    Tm, Vs = rnd(max(r, 1), M), rnd(max(r, 1), N)
    mu, mu_n, lam = 0.27, 0.405, 0.1
    A = (Tm[:r].T @ Vs[:r]).T.contiguous() if r else torch.zeros_like(D)   # N x M layout
    Yexp = mu * (Z - A)
    Y1, Z1 = torch.empty_like(Y), Z.clone()
    torch.cuda.synchronize()   # (the library runs on its own stream)
    ss1 = torch.zeros(72, dtype=torch.float64, device="cuda")
    st1 = lib.tlsq_k_zsweep_f64(h, p(D), p(Tm), p(Vs), None, p(Y), p(Y1), p(Z1), None, M, N, r, mu, 1 / mu, 0, 1 / mu_n, lam / mu_n, 0, p(ss1))
    eng.synchronize()
    Y2, Z2 = torch.empty_like(Y), torch.empty_like(Y)
    G2 = torch.empty((N, N), dtype=torch.float64, device="cuda")
    ss2 = torch.zeros(72, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    st2 = lib.tlsq_k_zsweep_gram_f64(h, p(D), p(Tm), p(Vs), p(Y), p(Y2), p(Z), p(Z2), None, M, N, r, mu, 1 / mu, 0, 1 / mu_n, lam / mu_n, 0, p(ss2), None, 0, p(G2), N)
    eng.synchronize()
    e1 = (Y1 - Yexp).abs().max().item(); e2 = (Y2 - Yexp).abs().max().item()
    bad = (Y1 != Y2)
    nb = int(bad.sum().item())
    where = ""
    if nb:
        idx = bad.nonzero()
        where = f" first bad (col,row) {idx[0].tolist()} last {idx[-1].tolist()} cols {idx[:,0].unique().numel()} rows {idx[:,1].unique().numel()}"
    print(M, N, r, "status", st1, st2, "ref err", e1, "fused err", e2, "nbad", nb, where, "ss", ss1[:64].sum().item(), ss2[:64].sum().item(), flush=True)
