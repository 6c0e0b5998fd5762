"""CPU model of the fp32 error of ONE F(4,3) x F(2,3) layer (csrc/conv_wino4.hip) on trained-like activations, term by term:
which of  the weight transform U = G4 g G2^T (packed in fp32) / the input transform B4^T d B2 / the fp32 accumulation over cin /
the output transform A4^T M A2  carries the distance from float64, and what other interpolation points would buy.
Run here (CPU only): python tools/probes/wino_error_model.py [layer_index]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import applied_image_processing_amd.synth as synth  # noqa: E402

f32, f64 = np.float32, np.float64


def winograd_mats(points):
    """F(4,3) matrices (AT 4x6, G 6x3, BT 6x6) for 5 finite points + infinity (Toom-Cook, as in wincnn), in float64."""
    from fractions import Fraction as Fr

    a = [Fr(p) for p in points]
    n = len(a) + 1      # 6
    # Vandermonde-based construction (Lavin's wincnn): AT[i][j] = a_j^i, G[j][k] = a_j^k / prod_{m != j}(a_j - a_m), BT from the Lagrange polynomials
    def poly_mul(p, q):
        r = [Fr(0)] * (len(p) + len(q) - 1)
        for i, x in enumerate(p):
            for j, y in enumerate(q):
                r[i + j] += x * y
        return r
    AT = [[a[j] ** i for j in range(n - 1)] + [Fr(1) if i == 3 else Fr(0)] for i in range(4)]
    G = []
    for j in range(n - 1):
        den = Fr(1)
        for m in range(n - 1):
            if m != j:
                den *= (a[j] - a[m])
        G.append([a[j] ** k / den for k in range(3)])
    G.append([Fr(0), Fr(0), Fr(1)])
    BT = []
    for j in range(n - 1):
        p = [Fr(1)]
        for m in range(n - 1):
            if m != j:
                p = poly_mul(p, [-a[m], Fr(1)])
        BT.append(p + [Fr(0)])
    full = [Fr(1)]
    for m in range(n - 1):
        full = poly_mul(full, [-a[m], Fr(1)])
    BT.append(full)
    to = lambda M: np.array([[float(x) for x in row] for row in M], dtype=f64)
    return to(AT), to(G), to(BT)


def check_mats(AT, G, BT):
    g = np.random.default_rng(0)
    d, w = g.standard_normal(6), g.standard_normal(3)
    y = AT @ ((G @ w) * (BT @ d))
    ref = np.array([d[i] * w[0] + d[i + 1] * w[1] + d[i + 2] * w[2] for i in range(4)])
    assert np.allclose(y, ref, atol=1e-9), (y, ref)


AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=f64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=f64)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=f64)


def fma_chain(U, V):
    """sum over cin of U[r,j,co,ci] * V[r,j,ci,t] accumulated sequentially in fp32 with fused multiply-adds -> [r,j,co,t]."""
    acc = np.zeros(U.shape[:3] + (V.shape[3],), dtype=f32)
    for ci in range(U.shape[3]):
        acc = (acc.astype(f64) + U[:, :, :, ci, None].astype(f64) * V[:, :, None, ci, :].astype(f64)).astype(f32)
    return acc


def wino_layer(x, w, b, mats, u_dtype=f32, v_dtype=f32, acc="fma32", out_dtype=f32):
    """x [cin,H,W] (H % 4 == 0, W % 2 == 0), w [cout,cin,3,3] -> relu-less conv output [cout,H,W] through F(4,3) rows x F(2,3) columns."""
    AT4, G4, BT4 = mats
    cin, H, W = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)), mode="reflect")
    th, tw = H // 4, W // 2
    # patches d[ci, t, 6, 4]
    d = np.stack([xp[:, 4 * i:4 * i + 6, 2 * j:2 * j + 4] for i in range(th) for j in range(tw)], axis=1)
    # input transform in v_dtype: columns first (d B2), then rows (B4^T .), as the kernel does
    dd = d.astype(v_dtype)
    t1 = np.einsum("ctab,jb->ctaj", dd, BT2.astype(v_dtype)).astype(v_dtype)
    V = np.einsum("ra,ctaj->rjct", BT4.astype(v_dtype), t1).astype(v_dtype)            # [6,4,cin,tiles]
    # weight transform
    if u_dtype == f32:
        U = np.zeros((6, 4) + w.shape[:2], dtype=f32)
        G4f, G2f = G4.astype(f32), G2.astype(f32)
        for a in range(3):
            for bb in range(3):
                U = (U + (G4f[:, None, None, None, a] * w[None, None, :, :, a, bb]).astype(f32) * G2f[None, :, None, None, bb]).astype(f32)
    else:
        U = np.einsum("ra,ocab,jb->rjoc", G4, w.astype(f64), G2).astype(f32)
    if acc == "fma32":
        M = fma_chain(U, V.astype(f32))
    else:
        M = np.einsum("rjoc,rjct->rjot", U.astype(f64), V.astype(f64))
    Md = M.astype(out_dtype)
    P = np.einsum("pr,rjot->pjot", AT4.astype(out_dtype), Md).astype(out_dtype)
    Y = np.einsum("pjot,qj->pqot", P, AT2.astype(out_dtype)).astype(out_dtype)         # [4,2,cout,tiles]
    Y = (Y + b.astype(out_dtype)[None, None, :, None]).astype(out_dtype)
    out = np.zeros((w.shape[0], H, W), dtype=out_dtype)
    t = 0
    for i in range(th):
        for j in range(tw):
            out[:, 4 * i:4 * i + 4, 2 * j:2 * j + 2] = Y[:, :, :, t].transpose(2, 0, 1)
            t += 1
    return out


def rel(a, b):
    return float(np.linalg.norm(a.astype(f64) - b) / np.linalg.norm(b))


def main():
    which = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    vgg, dec = synth.trained_like_state_dicts(0)
    from oracle import adain_oracle as O
    V64 = {k: torch.from_numpy(v).double() for k, v in vgg.items()}
    x = torch.from_numpy(synth.image(11, 1, 64, 64)).double()
    for e in O.ENC_CONVS:
        if e[0] == "pool":
            x = F.max_pool2d(x, 2, 2, 0, ceil_mode=True)
            continue
        idx, k, relu = e
        if idx == which:
            break
        x = F.conv2d(x, V64[f"{idx}.weight"], V64[f"{idx}.bias"]) if k == 1 else O._conv3x3_reflect(x, V64[f"{idx}.weight"], V64[f"{idx}.bias"])
        if relu:
            x = F.relu(x)
    xin = x[0, :, :16, :16].float().numpy()             # fp32 activations as the layer receives them (16 x 16 crop)
    w, b = vgg[f"{which}.weight"], vgg[f"{which}.bias"]
    truth = O._conv3x3_reflect(torch.from_numpy(xin).double()[None], torch.from_numpy(w).double(), torch.from_numpy(b).double())[0].numpy()
    direct = O._conv3x3_reflect(torch.from_numpy(xin)[None], torch.from_numpy(w), torch.from_numpy(b))[0].numpy()
    print(f"layer {which}: cin {w.shape[1]} cout {w.shape[0]}, input mean {xin.mean():.3f} std {xin.std():.3f}; direct fp32 (torch CPU) vs f64: {rel(direct, truth):.3e}")
    std = winograd_mats([0, 1, -1, 2, -2])
    check_mats(*std)
    rows = [("kernel model: all fp32", dict()),
            ("U from float64", dict(u_dtype=f64)),
            ("exact accumulation", dict(acc="f64")),
            ("exact input transform", dict(v_dtype=f64)),
            ("exact output transform", dict(out_dtype=f64)),
            ("U f64 + exact acc", dict(u_dtype=f64, acc="f64")),
            ("everything exact but fp32 U, V", dict(acc="f64", out_dtype=f64))]
    for name, kw in rows:
        print(f"  points 0,+-1,+-2   {name:34s} {rel(wino_layer(xin, w, b, std, **kw), truth):.3e}")
    for pts in ([0, 1, -1, 0.5, -0.5], [0, 1, -1, 0.5, -2], [0, 1, -1, 2, -0.5], [0, 0.5, -0.5, 2, -2], [0, 1, -1, 1.5, -1.5]):
        m = winograd_mats(pts)
        check_mats(*m)
        print(f"  points {str(pts):22s} all fp32: {rel(wino_layer(xin, w, b, m), truth):.3e}   U from f64: {rel(wino_layer(xin, w, b, m, u_dtype=f64), truth):.3e}")


if __name__ == "__main__":
    main()


def position_shares(which=12):
    """Which of the 24 transform positions' fp32 accumulation carries the error: accumulate ONE position exactly at a time."""
    vgg, _ = synth.trained_like_state_dicts(0)
    from oracle import adain_oracle as O
    V64 = {k: torch.from_numpy(v).double() for k, v in vgg.items()}
    x = torch.from_numpy(synth.image(11, 1, 64, 64)).double()
    for e in O.ENC_CONVS:
        if e[0] == "pool":
            x = F.max_pool2d(x, 2, 2, 0, ceil_mode=True)
            continue
        idx, k, relu = e
        if idx == which:
            break
        x = F.conv2d(x, V64[f"{idx}.weight"], V64[f"{idx}.bias"]) if k == 1 else O._conv3x3_reflect(x, V64[f"{idx}.weight"], V64[f"{idx}.bias"])
        if relu:
            x = F.relu(x)
    xin = x[0, :, :16, :16].float().numpy()
    w, b = vgg[f"{which}.weight"], vgg[f"{which}.bias"]
    truth = O._conv3x3_reflect(torch.from_numpy(xin).double()[None], torch.from_numpy(w).double(), torch.from_numpy(b).double())[0].numpy()
    AT4, G4, BT4 = winograd_mats([0, 1, -1, 2, -2])
    cin, H, W = xin.shape
    xp = np.pad(xin, ((0, 0), (1, 1), (1, 1)), mode="reflect")
    th, tw = H // 4, W // 2
    d = np.stack([xp[:, 4 * i:4 * i + 6, 2 * j:2 * j + 4] for i in range(th) for j in range(tw)], axis=1).astype(f32)
    t1 = np.einsum("ctab,jb->ctaj", d, BT2.astype(f32)).astype(f32)
    V = np.einsum("ra,ctaj->rjct", BT4.astype(f32), t1).astype(f32)
    U = np.einsum("ra,ocab,jb->rjoc", G4, w.astype(f64), G2).astype(f32)
    M32 = fma_chain(U, V).astype(f64)
    M64 = np.einsum("rjoc,rjct->rjot", U.astype(f64), V.astype(f64))

    def out_of(M):
        Y = np.einsum("pr,rjot,qj->pqot", AT4, M, AT2) + b.astype(f64)[None, None, :, None]
        out = np.zeros((w.shape[0], H, W))
        t = 0
        for i in range(th):
            for j in range(tw):
                out[:, 4 * i:4 * i + 4, 2 * j:2 * j + 2] = Y[:, :, :, t].transpose(2, 0, 1)
                t += 1
        return out

    base = rel(out_of(M32), truth)
    print(f"layer {which}: all positions fp32-accumulated {base:.3e}; exact everywhere {rel(out_of(M64), truth):.3e}")
    print("  error^2 share removed by accumulating ONE position exactly (rows r = points 0, 1, -1, 2, -2, inf; columns j):")
    for r in range(6):
        row = []
        for j in range(4):
            M = M32.copy()
            M[r, j] = M64[r, j]
            row.append(1.0 - (rel(out_of(M), truth) / base) ** 2)
        print("   r=%d  " % r + "  ".join(f"{v:6.3f}" for v in row) + f"     |M| rms per j: " + "  ".join(f"{np.sqrt((M64[r, j] ** 2).mean()):8.2f}" for j in range(4)))


if __name__ == "__main__" and len(sys.argv) > 2:
    position_shares(int(sys.argv[1]))
