"""NRMS (GloVe) planned training step with the in-projection per distinct key + sparse Dropout correction against the row-by-row form:
intermediates side by side (debug aid for csrc/dropcorr_ops.hip's engine wiring).
    python tools/dropcorr_debug.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd import engine as E  # noqa: E402
from legommenders_amd.synthetic import glove_like, init_nrms_params, make_world  # noqa: E402

dev = torch.device("cuda:0")
D, B, C, S, V = 128, 16, 5, 50, 3000
w = make_world(seed=9, n_items=700, n_users=300, n_rows=400, V=V)
P = init_nrms_params(D=D, V=V, n_cat=w["n_cat"], heads=8, glove=glove_like(V, 300, seed=4, device=dev), seed=6)
for k in P:
    if k.endswith("bias"):
        P[k] = torch.randn_like(P[k]) * 0.1
Pd = {k: v.to(dev).contiguous() for k, v in P.items()}
tb = E.ItemTables(w["title_tok"], w["title_len"], w["cat"], dev)
rs = np.random.RandomState(3)
users = rs.randint(0, 300, size=B)
ids = [torch.tensor(np.ascontiguousarray(a)).int().to(dev).contiguous() for a in
       (rs.randint(0, w["n_items"], size=(B, C)), w["user_hist"][users], np.maximum(w["user_hist_len"][users], 1))]
snap = {}
for form in ("0", "1"):
    os.environ["LEGO_NRMS_DROPCORR"] = form
    for p_override in (None, 0.0):
        eng = E.NrmsEngine(Pd, tb, B, C, S, heads=8, glove=True, seed=77)
        p_proj = eng.p_proj if p_override is None else p_proj
        if p_override is not None:
            eng.p_proj = 0.0
        G = eng.grads_like()
        eng.enable_plan_slots()
        eng.plan_on(torch.cuda.current_stream(), 0, *ids)
        eng.prefetch_masks(torch.cuda.current_stream(), 0)
        eng.use_slot(0)
        scores, loss = eng.forward(*ids, training=True, planned=True)
        torch.cuda.synchronize()
        R = int(eng.counters[0])
        U = int(eng.counters[6])
        s = dict(scores=scores.clone(), loss=float(loss), qkv=eng.item_ws["qkv"][:R].clone(), E=eng.E[:R].clone(), items=eng.items.clone(),
                 inv=eng.inv[:R].clone(), tokinfo=eng.tokinfo[:R].clone(), R=R, U=U, active=eng._dc_active,
                 Hu=eng.Hu[:U].clone(), mask=eng.mask_proj.clone())
        if eng._dc_active:
            s.update(Eu=eng.Eu[:U].clone(), QKVu=eng.QKVu[:U].clone(), WinT=eng.WinT.clone())
        eng.backward(G)
        torch.cuda.synchronize()
        s["G"] = {k: v.clone() for k, v in G.items()}
        snap[(form, p_override)] = s
        print(f"form {form} p_override {p_override}: R {R} U {U} dc_active {eng._dc_active} loss {float(loss):.6f}", flush=True)
W_in, b_in = Pd["item_op.multi_head_attention.in_proj_weight"], Pd["item_op.multi_head_attention.in_proj_bias"]
for po in (None, 0.0):
    a, b = snap[("0", po)], snap[("1", po)]
    print(f"--- p_override {po}")
    print(" inv equal", bool((a["inv"] == b["inv"]).all()), " tokinfo equal", bool((a["tokinfo"] == b["tokinfo"]).all()), " mask equal", bool((a["mask"] == b["mask"]).all()))
    live = ((a["tokinfo"] & 4) != 0)
    print(" Hu rows diff (live rows)", float(((a["Hu"][a["inv"].long()] - b["Hu"][b["inv"].long()]) * live.float()[:, None]).abs().max()))
    print(" E rows diff", float((a["E"] - b["E"]).abs().max()), "scale", float(a["E"].abs().max()))
    ref = a["E"].double() @ W_in.double().T + b_in.double()
    print(" qkv(row form) vs torch", float((a["qkv"].double() - ref).abs().max()))
    d = (b["qkv"].double() - ref).abs()
    print(" qkv(dropcorr) vs torch", float(d.max()), "rows off", int((d.max(1).values > 1e-4).sum()), "of", a["R"],
          "| of those live", int(((d.max(1).values > 1e-4) & ((a["tokinfo"] & 4) != 0)).sum()))
    bad = torch.nonzero(d.max(1).values > 1e-4).flatten()[:8].tolist()
    print("   first bad rows", bad, "cols of first", torch.nonzero(d[bad[0]] > 1e-4).flatten()[:10].tolist() if bad else None)
    if "Eu" in b:
        # the row the dropcorr kernel should produce, from ITS inputs, in torch: keep bits from the mask bytes
        R = a["R"]
        r = torch.arange(R, device=dev)
        mk = b["mask"][: ((R + 3) // 4) * D].view(-1, D)[(r >> 2)]
        keep = ((mk.int() >> (r & 3).int()[:, None]) & 1).double()
        keep = torch.where(live[:, None], keep, torch.ones_like(keep))
        sc = torch.where(live, torch.full((R,), 1.0 / (1.0 - (0.0 if po == 0.0 else p_proj)), device=dev, dtype=torch.double), torch.ones(R, device=dev, dtype=torch.double))
        Erow = b["Eu"][b["inv"].long()].double() * keep * sc[:, None]
        print(" E rows rebuilt from Eu/inv/mask vs E of the row form", float((Erow - a["E"].double()).abs().max()))
        print(" E rows (side expansion of dropcorr form) vs E of the row form", float((b["E"] - a["E"]).abs().max()))
        ref2 = Erow @ W_in.double().T + b_in.double()
        d2 = (b["qkv"].double() - ref2).abs()
        print(" qkv(dropcorr) vs torch from its own inputs", float(d2.max()))
        print(" QKVu vs Eu W^T", float((b["QKVu"].double() - b["Eu"].double() @ W_in.double().T).abs().max()))
        print(" WinT vs W^T", float((b["WinT"] - W_in.T).abs().max()))
    print(" items diff", float((a["items"] - b["items"]).abs().max()), " scores diff", float((a["scores"] - b["scores"]).abs().max()))
    for k in a["G"]:
        dd = float((a["G"][k] - b["G"][k]).abs().max())
        if dd > 2e-5 * max(1.0, float(a["G"][k].abs().max())):
            print("  grad", k, "diff", dd, "scale", float(a["G"][k].abs().max()))

# K1 called directly on the engine's captured inputs
import ctypes
from legommenders_amd._lib import call, LegoDropout


def Pp(t):
    return ctypes.c_void_p(t.data_ptr())


b = snap[("1", None)]
R, U = b["R"], b["U"]
live = ((b["tokinfo"] & 4) != 0)
r = torch.arange(R, device=dev)
mk = b["mask"][: ((R + 3) // 4) * D].view(-1, D)[(r >> 2)]
keep = ((mk.int() >> (r & 3).int()[:, None]) & 1).double()
keep = torch.where(live[:, None], keep, torch.ones_like(keep))
cnt = torch.tensor([R], dtype=torch.int32, device=dev)
for p in (0.1, 0.2, 0.5):
    for cap in (R, R + 1, 4 * R):
        sc = torch.where(live, torch.full((R,), 1.0 / (1.0 - p), device=dev, dtype=torch.double), torch.ones(R, device=dev, dtype=torch.double))
        ref = (b["Eu"][b["inv"].long()].double() * keep * sc[:, None]) @ W_in.double().T + b_in.double()
        out = torch.zeros(4 * R, 3 * D, device=dev)
        inv = torch.zeros(4 * R, dtype=torch.int32, device=dev); inv[:R] = b["inv"]
        ti = torch.zeros(4 * R, dtype=torch.int32, device=dev); ti[:R] = b["tokinfo"]
        dr = ctypes.byref(LegoDropout(p, 1, 1, b["mask"].data_ptr()))
        call("lego_qkv_expand_dropcorr", Pp(b["QKVu"]), 3 * D, Pp(b["Eu"]), D, Pp(b["WinT"]), 3 * D, Pp(b_in), Pp(inv), Pp(ti), dr, cap, Pp(cnt), D, 3 * D,
             Pp(out), 3 * D, None)
        torch.cuda.synchronize()
        print(f"direct K1 p={p} cap={cap}: max err {float((out[:R].double() - ref).abs().max()):.3e}")
print("p_proj of the engine:", p_proj)

# which coordinates did the kernel subtract?  acc_k = q - (out - b) / s, then least squares against the rows of W^T
p = 0.1
out = torch.zeros(R, 3 * D, device=dev)
dr = ctypes.byref(LegoDropout(p, 1, 1, b["mask"].data_ptr()))
call("lego_qkv_expand_dropcorr", Pp(b["QKVu"]), 3 * D, Pp(b["Eu"]), D, Pp(b["WinT"]), 3 * D, Pp(b_in), Pp(b["inv"]), Pp(b["tokinfo"]), dr, R, Pp(cnt), D, 3 * D,
     Pp(out), 3 * D, None)
torch.cuda.synchronize()
rows = torch.nonzero(live).flatten()[:3].tolist() + [int(torch.nonzero(live).flatten()[-1])]
for rr in rows:
    k = int(b["inv"][rr])
    q = b["QKVu"][k].double()
    acc_k = q - (out[rr].double() - b_in.double()) * (1 - p)
    A = b["WinT"].double().T                     # [3D, D]
    x = torch.linalg.lstsq(A, acc_k[:, None]).solution.flatten()
    h = b["Eu"][k].double()
    used = torch.nonzero(x.abs() > 1e-6).flatten().tolist()
    dropped = torch.nonzero(keep[rr] == 0).flatten().tolist()
    print(f"row {rr} key {k}: dropped {dropped}\n   kernel used {used}\n   coeff/h {[round(float(x[c] / h[c]), 3) for c in used]}")
