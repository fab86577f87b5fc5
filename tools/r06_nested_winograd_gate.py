"""Round 6, VERDICT r5 item 1(a), the numeric gate BEFORE any kernel work: conv3 (64 -> 128 channels, 5x5 'same', 20x20) as
   (i)   the one-dimensional Winograd F(4,5) along x that k_conv5_wpair computes (8 position products per 4 outputs and kernel row),
   (ii)  the nested form F(2,5) along y on top of F(4,5) along x (48 position products per 8 outputs instead of 80: -40 % matrix work),
both carried out in float32 (24 mantissa bits: a little better than the two fp16 pieces = 22 bits the device multiplies with) against the
same sums in float64.  Cook-Toom matrices from the evaluation points: A^T = E_m^T, G = E_r, B^T = (E_n^-1)^T (transposed Toom-Cook).
   python tools/r06_nested_winograd_gate.py"""
import numpy as np


def evalm(points, k):
    n = len(points) + 1
    E = np.zeros((n, k))
    for j, a in enumerate(points):
        E[j] = [a ** i for i in range(k)]
    E[n - 1, k - 1] = 1.0          # the point at infinity takes the leading coefficient
    return E


def cook_toom(m, r, points):
    n = m + r - 1
    assert len(points) == n - 1
    At = evalm(points, m).T
    G = evalm(points, r)
    Bt = np.linalg.inv(evalm(points, n)).T
    return At, G, Bt


def check(At, G, Bt, m, r):
    rng = np.random.default_rng(0)
    d = rng.standard_normal(m + r - 1); g = rng.standard_normal(r)
    y = At @ ((G @ g) * (Bt @ d))
    ref = np.array([np.dot(d[i:i + r], g) for i in range(m)])
    assert np.abs(y - ref).max() < 1e-9, np.abs(y - ref).max()


def run(seed, pts_y):
    rng = np.random.default_rng(seed)
    CI, CO, S = 64, 128, 20
    # activations behind ReLU + pool (non-negative, a few large), weights of a trained-looking layer
    x = np.maximum(rng.standard_normal((CI, S + 4, S + 4)) * 1.5 + 0.3, 0.0)
    x[:, :2] = 0; x[:, -2:] = 0; x[:, :, :2] = 0; x[:, :, -2:] = 0
    w = rng.standard_normal((CO, CI, 5, 5)) * (1.0 / np.sqrt(CI * 25))
    ref = np.zeros((CO, S, S))
    for ky in range(5):
        for kx in range(5):
            ref += np.einsum("oc,cyx->oyx", w[:, :, ky, kx], x[:, ky:ky + S, kx:kx + S])
    Atx, Gx, Btx = cook_toom(4, 5, [0, 1, -1, 2, -2, 0.5, -0.5]); check(Atx, Gx, Btx, 4, 5)
    Aty, Gy, Bty = cook_toom(2, 5, pts_y); check(Aty, Gy, Bty, 2, 5)
    f32 = np.float32
    # (i) F(4,5) along x, direct along y -- all arithmetic in float32
    U = np.einsum("pk,ocyk->ocyp", Gx, w).astype(f32)                           # [co][ci][ky][8]
    out1 = np.zeros((CO, S, S), f32)
    for tx in range(S // 4):
        dwin = x[:, :, 4 * tx:4 * tx + 8]                                        # [ci][24][8]
        V = np.einsum("pk,cyk->cyp", Btx.astype(f32), dwin.astype(f32)).astype(f32)   # [ci][24][8]
        for y in range(S):
            M = np.zeros((CO, 8), f32)
            for ky in range(5):
                M += np.einsum("ocp,cp->op", U[:, :, ky], V[:, y + ky]).astype(f32)
            out1[:, y, 4 * tx:4 * tx + 4] = (M @ Atx.T.astype(f32))
    # (ii) nested: F(2,5) along y on top
    U2 = np.einsum("qk,ockp->ocqp", Gy, np.einsum("pk,ocyk->ocyp", Gx, w)).astype(f32)     # [co][ci][6][8]
    out2 = np.zeros((CO, S, S), f32)
    for tx in range(S // 4):
        dwin = x[:, :, 4 * tx:4 * tx + 8]
        V = np.einsum("pk,cyk->cyp", Btx.astype(f32), dwin.astype(f32)).astype(f32)           # x transform: what conv1 + conv2's tail writes as V3
        for ty in range(S // 2):
            V2 = np.einsum("qy,cyp->cqp", Bty.astype(f32), V[:, 2 * ty:2 * ty + 6]).astype(f32)   # y transform of 6 transformed rows
            M = np.einsum("ocqp,cqp->oqp", U2, V2).astype(f32)                     # 48 position products per (ci, co)
            Y = np.einsum("rq,oqp->orp", Aty.astype(f32), M).astype(f32)            # 2 rows x 8 positions
            out2[:, 2 * ty:2 * ty + 2, 4 * tx:4 * tx + 4] = np.einsum("orp,xp->orx", Y, Atx.astype(f32))
    sc = np.abs(ref).max()
    return np.abs(out1 - ref).max() / sc, np.abs(out2 - ref).max() / sc, np.abs(np.einsum("qy->q", np.abs(Bty))).max(), np.abs(Gy).sum(1).max()


if __name__ == "__main__":
    print("# relative error (max |err| / max |out|) of conv3 in float32 arithmetic against float64; the device's 1-D form sits at 3.5e-6 on the softmax")
    for name, pts in (("0, +-1, +-1/2", [0, 1, -1, 0.5, -0.5]), ("0, +-1, +-2", [0, 1, -1, 2, -2]), ("0, +-1/2, +-3/2", [0, 0.5, -0.5, 1.5, -1.5])):
        e1s, e2s = [], []
        for seed in (1, 2, 3):
            e1, e2, bnorm, gnorm = run(seed, pts)
            e1s.append(e1); e2s.append(e2)
        print("F(2,5) points %-16s: 1-D F(4,5) %.2e   nested F(2,5) x F(4,5) %.2e   (x %.1f)   max |B^T| row sum %.1f, max |G| row sum %.2f"
              % (name, max(e1s), max(e2s), max(e2s) / max(e1s), bnorm, gnorm))
