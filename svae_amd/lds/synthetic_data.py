"""Synthetic LDS models and recognition potentials for tests and benchmarks.

Mirrors the reference's generator /root/reference/svae/lds/synthetic_data.py:8-28 (`rand_lds`:
A = randn scaled to spectral radius < 1 via /(max|eig| + 0.1), Sigma_states = B B', Sigma_init =
rand_psd, mu_init = randn) and the conversions of svae/lds/gaussian.py:110-114,130-143
(`mean_to_natural`, `pair_mean_to_natural`); node potentials mimic the recognition head
`gaussian_info` (svae/nnet.py:43-47): J = -1/2 softplus(z), h ~ N(0,1)  (SURVEY.md section 8d).
Host-side NumPy only (this is input construction, not the hot path).
"""
import numpy as np


def rand_lds_natparam(n, rng):
    """-> (init_params, pair_params) in natural form for a random stable LDS."""
    A = rng.standard_normal((n, n))
    A /= np.max(np.abs(np.linalg.eigvals(A))) + 0.1
    Bm = rng.standard_normal((n, n))
    sigma_states = Bm @ Bm.T
    S0 = rng.standard_normal((n, n))
    sigma_init = S0 @ S0.T
    mu_init = rng.standard_normal(n)
    # mean_to_natural, gaussian.py:110-114
    J0 = -0.5 * np.linalg.inv(sigma_init)
    h0 = np.linalg.solve(sigma_init, mu_init)
    logZ0 = -0.5 * mu_init @ h0 - 0.5 * np.linalg.slogdet(sigma_init)[1]
    # pair_mean_to_natural, gaussian.py:130-143
    temp = np.linalg.solve(sigma_states, A)
    J11 = -0.5 * A.T @ temp
    J12 = temp.T
    J22 = -0.5 * np.linalg.inv(sigma_states)
    logZ = -0.5 * np.linalg.slogdet(sigma_states)[1]
    return (J0, h0, logZ0), (J11, J12, J22, logZ)


def rotation_lds_natparam(n, rng):
    """A well-conditioned LDS in natural form: rotation-like dynamics A = 0.97 Q (Q orthogonal, spectral radius
    0.97) with nearly isotropic state noise (cond ~ 10), unit initial covariance.  `rand_lds_natparam`'s state
    noise B B' has cond ~ n^2, on which the reference's own fp64 path drifts by ~1e-5 from extended precision at
    n = 64, T = 1000 (tests/test_lds_tile_hip.py); this is the model the latent-dim-64 workload of bench.py runs and
    the tests pin against the reference's compiled path at 1e-8."""
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = 0.97 * Q
    Bm = rng.standard_normal((n, n))
    S = 0.3 * np.eye(n) + 0.02 * (Bm @ Bm.T) / n
    Si = np.linalg.inv(S)
    J0, h0 = -0.5 * np.eye(n), rng.standard_normal(n)
    return (J0, h0, 0.), (-0.5 * A.T @ Si @ A, A.T @ Si, -0.5 * Si, -0.5 * np.linalg.slogdet(S)[1])


def rand_node_potentials(shape, rng, with_logZ=False):
    """shape = (T, n) or (B, T, n) -> (J diag, h[, logZ]) like nnet.gaussian_info (nnet.py:43-47)."""
    J = -0.5 * np.log1p(np.exp(rng.standard_normal(shape)))
    h = rng.standard_normal(shape)
    if with_logZ:
        return J, h, 0.1 * rng.standard_normal(shape[:-1])
    return J, h


def rand_slds_global_natparam(K, n, rng):
    """SLDS global natural parameters for benchmarks (BASELINE configs[3]): K rotation-like dynamics with different
    angles, as ((dirichlet (K), dirichlet rows (K,K)), [(NIW dense, MNIW 4-tuple)] * K) of float64 CPU tensors --
    the nesting of svae/models/slds_svae.py:27-31 (standard -> natural parameters through svae_amd.distributions)."""
    import torch
    from ..distributions import expfam
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64)
    lds = []
    for k in range(K):
        nu, S = n + 1. + rng.random(), 2. * (n + 1) * np.eye(n)
        th = 0.3 * (k + 1)
        M = 0.95 * np.eye(n)
        if n >= 2:
            M[:2, :2] = 0.95 * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        # (contiguous leaves: mniw_standard_to_natural returns a transposed view as its first block, and every consumer
        #  that packs the parameters -- the device maps, the host copy for the prior term -- would copy it per state per call)
        lds.append((expfam.niw_standard_to_natural(t(S), t(0.3 * rng.standard_normal(n)), t(0.5), t(nu)).contiguous(),
                    tuple(x.contiguous() for x in expfam.mniw_standard_to_natural(t(nu), t(S), t(M), t(0.2 * np.eye(n))))))
    return (t(rng.random(K) * 2.), t(rng.random((K, K)) * 2. + 3. * np.eye(K))), lds
