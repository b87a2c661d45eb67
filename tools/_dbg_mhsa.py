import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd._lib import call
from legommenders_amd.kernels import _ptr, _stream, _drop
dev = torch.device("cuda:0")
for hd, heads, lens in ((32, 2, [5, 3, 32, 7]), (16, 4, [33, 1, 7, 20, 12]), (8, 8, [33, 1, 7, 20, 12]), (64, 2, [5, 40, 3])):
    D = hd * heads
    n, Lmax, R = len(lens), 64, sum(lens)
    seg = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=dev)
    torch.manual_seed(0)
    qkv = torch.randn(R, 3 * D, device=dev) * 0.7
    out = torch.full((R + 8, D), 7.0, device=dev)
    lse = torch.full((R, heads), float("nan"), device=dev)
    probs = torch.zeros(R, heads, Lmax, device=dev)
    call("lego_mhsa_core_fwd", _ptr(qkv), 3 * D, _ptr(seg), n, None, D, heads, _ptr(out), D, _ptr(lse), _ptr(probs), Lmax, None, R, 0, None, None, _stream())
    torch.cuda.synchronize()
    q = qkv.double().cpu().view(R, 3, heads, hd)
    print("hd", hd, "heads", heads, "guard rows untouched:", bool((out[R:] == 7.0).all()))
    for s, L in enumerate(lens):
        b = int(seg[s])
        sc = torch.einsum("ihd,jhd->hij", q[b:b + L, 0], q[b:b + L, 1]) / hd ** 0.5
        o = torch.einsum("hij,jhd->ihd", torch.softmax(sc, 2), q[b:b + L, 2]).reshape(L, D)
        pr = probs.cpu().double().reshape(-1)
        pe = max(float((pr[(b * heads + h * L) * Lmax:][: L * L].reshape(L, L).t() - torch.softmax(sc, 2)[h]).abs().max()) for h in range(heads))
        le = float((lse[b:b + L].double().cpu() - torch.logsumexp(sc, 2).t()).abs().max())
        print("  probs err", pe, "lse err", le)
        e = (out[b:b + L].double().cpu() - o).abs()
        print("  seg", s, "L", L, "max err", float(e.max()), "per head", [round(float(e[:, h * hd:(h + 1) * hd].max()), 4) for h in range(heads)],
              "rows bad", (e.max(1).values > 1e-4).nonzero().view(-1).tolist()[:10])
