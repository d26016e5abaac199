"""Prototype of a TIME-PARTITIONED LDS E-step (round-3 verdict, item 3): four elimination chains per sequence -- the
two outer chains of the two-ended kernel (from t = 0 and t = T-1) and two INTERIOR chains that start next to a cut node
x_m and walk outwards, each carrying the n x n coupling block between its front and the cut -- then a reduced solve on
the three boundary nodes (x_a, x_m, x_b) and one moment-form smoother per chain.  Checked here against the dense
(Tn x Tn) solve; the point of the file is the COST MODEL at the bottom: what the partition would buy the headline
kernel (512 sequences x T = 200, n = 10: two wavefronts per sequence on 1024 SIMDs).

Test / design infrastructure only (NumPy, float64); nothing in svae_amd/ imports it.
    python tools/proto/partition4_proto.py
Reference algorithm: /root/reference/svae/lds/cython_lds_inference.pyx:28-90, 149-210 (one-directional filter + RTS).
"""
import numpy as np

MACS = {"elim_outer": 0, "elim_inner": 0, "smooth_outer": 0, "smooth_inner": 0, "reduced": 0}


def rand_problem(T, n, rng):
    A = rng.standard_normal((n, n)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
    Qi = np.linalg.inv(np.eye(n) * 0.5 + 0.1 * (lambda B: B @ B.T)(rng.standard_normal((n, n))))
    J11, J12, J22 = A.T @ Qi @ A, -A.T @ Qi, Qi            # info-form pair blocks: [[J11, J12], [J12', J22]]
    J0, h0 = np.eye(n) * 1.3, rng.standard_normal(n)
    Jn = np.log1p(np.exp(rng.standard_normal((T, n))))      # diagonal node precisions
    hn = rng.standard_normal((T, n))
    return (J0, h0), (J11, J12, J22), (Jn, hn)


def dense_solution(init, pair, node):
    (J0, h0), (J11, J12, J22), (Jn, hn) = init, pair, node
    T, n = hn.shape
    J = np.zeros((T * n, T * n)); h = hn.reshape(-1).copy()
    for t in range(T):
        s = slice(t * n, (t + 1) * n)
        J[s, s] += np.diag(Jn[t])
        if t == 0:
            J[s, s] += J0; h[s] += h0
        if t < T - 1:
            u = slice((t + 1) * n, (t + 2) * n)
            J[s, s] += J11; J[u, u] += J22; J[s, u] += J12; J[u, s] += J12.T
    S = np.linalg.inv(J); mu = S @ h
    Ex = mu.reshape(T, n)
    Exx = np.stack([S[t*n:(t+1)*n, t*n:(t+1)*n] + np.outer(Ex[t], Ex[t]) for t in range(T)])
    Exn = np.stack([S[t*n:(t+1)*n, (t+1)*n:(t+2)*n] + np.outer(Ex[t], Ex[t+1]) for t in range(T - 1)])
    return Ex, Exx, Exn


def eliminate(P, h, Jtu, Jtv):
    """Eliminate a node with precision P, potential h, coupling Jtu to its chain neighbour u and (interior chains) Jtv
    to the cut node v.  Returns the record for the smoother and the Schur updates.  MAC counts as the kernel would do
    them: Gauss-Jordan on [P | Jtu | Jtv | h] + the Schur products."""
    n = P.shape[0]
    Pi = np.linalg.inv(P)
    c = Pi @ h
    G1 = -Pi @ Jtu
    upd = {"uu": -Jtu.T @ Pi @ Jtu, "hu": -Jtu.T @ c}
    rec = {"Pi": Pi, "c": c, "G1": G1}
    if Jtv is None:
        MACS["elim_outer"] += n * n * (2 * n + 1) + n * n * (n + 1)          # GJ rows x cols + one Schur product (+h)
    else:
        rec["G2"] = -Pi @ Jtv
        upd.update(vv=-Jtv.T @ Pi @ Jtv, uv=-Jtu.T @ Pi @ Jtv, hv=-Jtv.T @ c)
        MACS["elim_inner"] += n * n * (3 * n + 1) + 3 * n * n * n + 2 * n * n  # wider GJ + three Schur products
    return rec, upd


def run(T, n, a, m, b, seed=0):
    """outer chain A: nodes 0..a-1 upward; interior C: m-1..a+1 downward; interior D: m+1..b-1 upward; outer B:
    T-1..b+1 downward; boundary nodes a < m < b."""
    rng = np.random.default_rng(seed)
    init, pair, node = rand_problem(T, n, rng)
    (J0, h0), (J11, J12, J22), (Jn, hn) = init, pair, node
    diag = lambda t: np.diag(Jn[t]) + (J0 if t == 0 else 0) + (J11 if t < T - 1 else 0) + (J22 if t > 0 else 0)
    pot = lambda t: hn[t] + (h0 if t == 0 else 0)
    recs = {}
    # --- outer chains (as the two-ended kernel does) ------------------------------------------------------------
    accP = {t: diag(t) for t in range(T)}; acch = {t: pot(t).copy() for t in range(T)}
    for t in range(0, a):                                   # chain A: neighbour u = t+1, coupling J_{t,t+1} = J12
        recs[t], upd = eliminate(accP[t], acch[t], J12, None); recs[t]["u"] = t + 1
        accP[t + 1] = accP[t + 1] + upd["uu"]; acch[t + 1] = acch[t + 1] + upd["hu"]
    for t in range(T - 1, b, -1):                           # chain B: neighbour u = t-1, coupling J_{t,t-1} = J12'
        recs[t], upd = eliminate(accP[t], acch[t], J12.T, None); recs[t]["u"] = t - 1
        accP[t - 1] = accP[t - 1] + upd["uu"]; acch[t - 1] = acch[t - 1] + upd["hu"]
    # --- interior chains: carry the coupling F between the front and the cut node m --------------------------------
    cross = {}                                              # reduced system: coupling blocks between boundary nodes
    for (rng_t, step, end) in ((range(m - 1, a, -1), -1, a), (range(m + 1, b, +1), +1, b)):
        F = None
        for t in rng_t:
            Jtu = J12.T if step < 0 else J12                # to the neighbour further from the cut
            Jtv = (J12 if step < 0 else J12.T) if F is None else F      # first node: direct pair coupling to x_m
            recs[t], upd = eliminate(accP[t], acch[t], Jtu, Jtv); recs[t]["u"] = t + step
            accP[t + step] = accP[t + step] + upd["uu"]; acch[t + step] = acch[t + step] + upd["hu"]
            accP[m] = accP[m] + upd["vv"]; acch[m] = acch[m] + upd["hv"]
            F = upd["uv"]                                   # fill-in: coupling J_{u, m} = -J_tu' P^-1 J_tv (the next node's Jtv)
        cross[end] = F if F is not None else (J12 if step < 0 else J12.T)
    # --- reduced solve on (x_a, x_m, x_b) ----------------------------------------------------------------------------
    Z = np.zeros((n, n))
    Jr = np.block([[accP[a], cross[a], Z], [cross[a].T, accP[m], cross[b].T], [Z, cross[b], accP[b]]])
    hr = np.concatenate([acch[a], acch[m], acch[b]])
    Sr = np.linalg.inv(Jr); mr = Sr @ hr
    MACS["reduced"] += 3 * (n * n * (2 * n + 1)) + 6 * n ** 3
    M2 = Sr + np.outer(mr, mr)                              # second moments of the boundary nodes
    idx = {a: slice(0, n), m: slice(n, 2 * n), b: slice(2 * n, 3 * n)}
    Ex = np.zeros((T, n)); Exx = np.zeros((T, n, n)); Exm = {}           # Exm[t] = E[x_t x_m']
    for t in (a, m, b):
        Ex[t] = mr[idx[t]]; Exx[t] = M2[idx[t], idx[t]]; Exm[t] = M2[idx[t], idx[m]]
    Exn = np.zeros((T - 1, n, n))
    Exn_set = lambda t, u, val: Exn.__setitem__(min(t, u), val if t < u else val.T)   # stores E[x_lo x_hi']
    # --- smoothers ------------------------------------------------------------------------------------------------------
    def outer(order):
        for t in order:
            r = recs[t]; u = r["u"]
            Ex[t] = r["c"] + r["G1"] @ Ex[u]
            W = r["G1"] @ Exx[u] + np.outer(r["c"], Ex[u])                 # E[x_t x_u']
            Exx[t] = r["Pi"] + W @ r["G1"].T + np.outer(Ex[t], r["c"])
            Exn_set(t, u, W)
            MACS["smooth_outer"] += 2 * (n + 1) ** 3
    def inner(order):
        for t in order:
            r = recs[t]; u = r["u"]
            Ex[t] = r["c"] + r["G1"] @ Ex[u] + r["G2"] @ Ex[m]
            Wu = r["G1"] @ Exx[u] + r["G2"] @ Exm[u].T + np.outer(r["c"], Ex[u])      # E[x_t x_u']
            Wm = r["G1"] @ Exm[u] + r["G2"] @ Exx[m] + np.outer(r["c"], Ex[m])        # E[x_t x_m']
            Exx[t] = r["Pi"] + Wu @ r["G1"].T + Wm @ r["G2"].T + np.outer(Ex[t], r["c"])
            Exm[t] = Wm
            Exn_set(t, u, Wu)
            MACS["smooth_inner"] += n * (2 * n + 1) * (2 * n + 1) + n * (2 * n + 1) * n
    outer(range(a - 1, -1, -1)); outer(range(b + 1, T))
    inner(range(a + 1, m)); inner(range(b - 1, m, -1))
    want = dense_solution(init, pair, node)
    # pairs next to the cut: (m-1, m) and (m, m+1) come from the interior records' E[x_t x_m']
    if m - 1 > a: Exn[m - 1] = Exm[m - 1]
    if m + 1 < b: Exn[m] = Exm[m + 1].T
    err = max(np.abs(Ex - want[0]).max(), np.abs(Exx - want[1]).max(), np.abs(Exn - want[2]).max())
    return err


if __name__ == "__main__":
    T, n = 200, 10
    for (a, m, b) in ((66, 100, 133), (60, 100, 139)):
        for k in MACS: MACS[k] = 0
        err = run(T, n, a, m, b)
        print("T=%d n=%d boundaries (%d, %d, %d): max abs error vs the dense solve %.2e" % (T, n, a, m, b, err))
        print("   multiply-adds:", MACS)
    # ---- cost model in INSTRUCTIONS of the existing kernels (ISA counts): per step of both chains of a
    # kind in one instruction stream
    E2, S1 = 426, 199          # two-ended elimination step (chains A and B together); S4 smoother step (one chain)
    # interior steps: Gauss-Jordan with 2n+1 right-hand-side columns needs a second register per matrix row (+100 DPP
    # FMAs), three Schur products instead of one (+100), a wider hand-off record (+~100); the interior smoother's two
    # products are n x (2n+1) x (2n+1) and n x (2n+1) x n instead of two (n+1)^3: x 2.4
    E2i, S1i = 426 + 300, int(199 * 2.4)
    now = 100 * E2 + 100 * S1
    print("\ninstructions on the longest wavefront of a sequence, T = 200 (two wavefronts per sequence):")
    print("   today (two chains, S4 smoother):                 %6d" % now)
    for to, ti in ((67, 33), (60, 40), (75, 25)):
        wf0 = to * E2 + to * S1 + ti * S1i                   # outer eliminations, then smoothers of chains A and C
        wf1 = ti * E2i                                       # interior eliminations (in parallel with wf0's)
        crit = max(to * E2, ti * E2i) + 3000 + to * S1 + ti * S1i
        print("   four chains, outer %2d + interior %2d steps:       %6d   (eliminations %d | %d, reduced solve ~3000, "
              "smoothers %d per wavefront)  -> %+.1f %%" % (to, ti, crit, to * E2, ti * E2i, to * S1 + ti * S1i,
                                                           100.0 * (crit - now) / now))
