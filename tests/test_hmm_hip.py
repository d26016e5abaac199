"""GPU parity tests of the HMM E-step kernel against the reference's compiled hmm_logZ /
hmm_logZ_grad (oracle/_ref, built from svae/hmm/cython_hmm_inference.pyx) and the NumPy oracle."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import hmm_numpy, ref  # noqa: E402  (checker only)


def _np(x):
    return x.detach().cpu().numpy()


def _problem(B, T, K, rng, scale=1.0):
    init = np.log(rng.dirichlet(np.ones(K)))
    pair = np.log(rng.dirichlet(np.ones(K), size=K)) + 0.3 * rng.standard_normal((K, K))   # unnormalised
    node = scale * rng.standard_normal((B, T, K))
    return init, pair, node


@pytest.mark.parametrize("B,T,K,scale", [(5, 7, 3, 1.0), (9, 50, 8, 3.0), (2, 500, 8, 50.0), (3, 1, 4, 1.0),
                                         (4, 33, 16, 1.0), (1, 12, 1, 2.0), (6, 2, 5, 1.0), (7, 3, 9, 2.0),
                                         (13, 17, 8, 1.0), (5, 16, 2, 4.0), (3, 25, 12, 1.0)])
def test_hmm_estep_against_oracle_and_reference(B, T, K, scale):
    from svae_amd.hmm.hmm_inference import hmm_estep
    rng = np.random.default_rng(B * 100 + T + K)
    init, pair, node = _problem(B, T, K, rng, scale)
    logZ, (Ei, Et, Es) = hmm_estep((init, pair, node))
    for b in range(B):
        lz, (oi, ot, os_) = hmm_numpy.hmm_estep((init, pair, node[b]))
        assert float(logZ[b]) == pytest.approx(lz, rel=1e-10, abs=1e-10)
        np.testing.assert_allclose(_np(Ei[b]), oi, rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(_np(Et[b]), ot, rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(_np(Es[b]), os_, rtol=1e-8, atol=1e-12)
        if ref.available():
            rz, aux = ref.hmm_logZ((init, pair, node[b]))
            gi, gp, gn = ref.hmm_logZ_grad(1.0, aux)
            assert float(logZ[b]) == pytest.approx(rz, rel=1e-10, abs=1e-10)
            np.testing.assert_allclose(_np(Et[b]), gp, rtol=1e-8, atol=1e-11)
            np.testing.assert_allclose(_np(Es[b]), gn, rtol=1e-8, atol=1e-12)
            np.testing.assert_allclose(_np(Ei[b]), gi, rtol=1e-8, atol=1e-12)
    # properties: marginals sum to one, transition counts to T-1
    assert float((Es.sum(-1) - 1).abs().max()) < 1e-12
    assert float((Et.sum((-1, -2)) - (T - 1)).abs().max()) < 1e-10 * max(1, T)


def test_hmm_batched_pair_params_and_unbatched_call():
    from svae_amd.hmm.hmm_inference import hmm_estep, hmm_logZ
    rng = np.random.default_rng(3)
    B, T, K = 3, 9, 4
    init, _, node = _problem(B, T, K, rng)
    pairs = np.stack([_problem(1, 1, K, rng)[1] for _ in range(B)])
    logZ, (Ei, Et, Es) = hmm_estep((init, pairs, node))
    for b in range(B):
        lz, (oi, ot, os_) = hmm_numpy.hmm_estep((init, pairs[b], node[b]))
        assert float(logZ[b]) == pytest.approx(lz, rel=1e-10)
        np.testing.assert_allclose(_np(Et[b]), ot, rtol=1e-8, atol=1e-11)
    lz1 = hmm_logZ((init, pairs[0], node[0]))
    assert lz1.dim() == 0 and float(lz1) == pytest.approx(float(logZ[0]), rel=1e-13)


@pytest.mark.parametrize("gap", [40.0, 800.0])
def test_hmm_near_deterministic_transitions(gap):
    """Transition log-potentials `gap` below the diagonal (exp(-800) underflows in a scaled recursion;
    the reference's log-space pass, cython_hmm_inference.pyx:93-121, keeps it).  The chain can only stay
    where the initial potential puts it unless the likelihood gain of a switch exceeds the gap."""
    from svae_amd.hmm.hmm_inference import hmm_estep
    rng = np.random.default_rng(int(gap))
    B, T, K = 4, 60, 6
    init = np.log(rng.dirichlet(np.ones(K)))
    pair = -gap * (1.0 - np.eye(K)) + 0.1 * rng.standard_normal((K, K))
    node = 3.0 * rng.standard_normal((B, T, K))
    logZ, (Ei, Et, Es) = hmm_estep((init, pair, node))
    for b in range(B):
        lz, (oi, ot, os_) = hmm_numpy.hmm_estep((init, pair, node[b]))
        assert float(logZ[b]) == pytest.approx(lz, rel=1e-10, abs=1e-9)
        np.testing.assert_allclose(_np(Es[b]), os_, rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(_np(Et[b]), ot, rtol=1e-8, atol=1e-11)
        if ref.available():
            rz, aux = ref.hmm_logZ((init, pair, node[b]))
            assert float(logZ[b]) == pytest.approx(rz, rel=1e-10, abs=1e-9)
    assert bool(torch.isfinite(logZ).all())


def test_hmm_forced_transition_through_a_tiny_entry():
    """The only path goes through a transition of log-potential -800 at a fixed time (state 0 is the only
    possible state up to t = 5, impossible afterwards, and its only exit is that entry): a scaled step
    underflows there; log Z ~ -800 must come out finite and equal to the reference's log-space value."""
    from svae_amd.hmm.hmm_inference import hmm_estep
    K, T = 3, 12
    init = np.array([0.0, -1e4, -1e4])
    pair = np.array([[0.0, -800.0, -1e4], [-1e4, 0.0, -1.0], [-1e4, -1.0, 0.0]])
    node = np.zeros((2, T, K))
    node[0, :6, 1:] = -1e4
    node[0, 6:, 0] = -1e4
    node[1] = 0.3 * np.random.default_rng(0).standard_normal((T, K))      # an ordinary sequence in the same wave
    logZ, (Ei, Et, Es) = hmm_estep((init, pair, node))
    for b in range(2):
        lz, (oi, ot, os_) = hmm_numpy.hmm_estep((init, pair, node[b]))
        assert np.isfinite(lz) and float(logZ[b]) == pytest.approx(lz, rel=1e-9)
        np.testing.assert_allclose(_np(Es[b]), os_, rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(_np(Et[b]), ot, rtol=1e-7, atol=1e-9)
    assert float(logZ[0]) < -790


def test_hmm_two_ended_kernel_with_one_flagged_sequence_among_many():
    """hmm_estep2_kernel (alpha and beta recursions on two wavefronts, each with its own scaling) serves every sequence
    whose normalisers stay above 1e-200; a sequence that underflows is flagged and redone by the one-directional kernel
    (log-space steps), wavefront by wavefront: here sequence 6 of 11 (third of its wavefront) is the forced-transition
    chain of the test above, the others are ordinary -- all of them must match the log-space oracle."""
    from svae_amd.hmm.hmm_inference import hmm_estep
    K, T, B = 3, 12, 11
    init = np.array([0.0, -1e4, -1e4])
    pair = np.array([[0.0, -800.0, -1e4], [-1e4, 0.0, -1.0], [-1e4, -1.0, 0.0]])
    rng = np.random.default_rng(4)
    node = 0.5 * rng.standard_normal((B, T, K))
    node[6] = 0.0
    node[6, :6, 1:] = -1e4
    node[6, 6:, 0] = -1e4
    logZ, (Ei, Et, Es) = hmm_estep((init, pair, node))
    for b in range(B):
        lz, (oi, ot, os_) = hmm_numpy.hmm_estep((init, pair, node[b]))
        assert np.isfinite(lz) and float(logZ[b]) == pytest.approx(lz, rel=1e-9, abs=1e-9)
        np.testing.assert_allclose(_np(Ei[b]), oi, rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(_np(Es[b]), os_, rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(_np(Et[b]), ot, rtol=1e-7, atol=1e-9)
    assert float(logZ[6]) < -790


# ---- 17 <= K <= 64: one wavefront per sequence (csrc/hmm_estep_wide.hip, round 6) -------------------------------------
@pytest.mark.parametrize("B,T,K,scale", [(5, 7, 17, 1.0), (3, 50, 20, 3.0), (2, 120, 32, 2.0), (4, 9, 33, 1.0),
                                         (3, 1, 40, 1.0), (2, 2, 64, 1.0), (6, 31, 48, 4.0), (2, 500, 64, 20.0)])
def test_hmm_wide_kernel_against_oracle_and_reference(B, T, K, scale):
    """The reference's compiled kernels take any number of states (cython_hmm_inference.pyx:93-166); the DPP-row kernels
    stop at 16.  17 <= K <= 64 run one wavefront per sequence, lane = state: against the reference's hmm_logZ /
    hmm_logZ_grad and the NumPy restatement, every sequence."""
    from svae_amd.hmm.hmm_inference import hmm_estep
    rng = np.random.default_rng(B * 100 + T + K)
    init, pair, node = _problem(B, T, K, rng, scale)
    logZ, (Ei, Et, Es) = hmm_estep((init, pair, node))
    for b in range(B):
        lz, (oi, ot, os_) = hmm_numpy.hmm_estep((init, pair, node[b]))
        assert float(logZ[b]) == pytest.approx(lz, rel=1e-10, abs=1e-10)
        np.testing.assert_allclose(_np(Ei[b]), oi, rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(_np(Et[b]), ot, rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(_np(Es[b]), os_, rtol=1e-8, atol=1e-12)
        if ref.available():
            rz, aux = ref.hmm_logZ((init, pair, node[b]))
            gi, gp, gn = ref.hmm_logZ_grad(1.0, aux)
            assert float(logZ[b]) == pytest.approx(rz, rel=1e-10, abs=1e-10)
            np.testing.assert_allclose(_np(Et[b]), gp, rtol=1e-8, atol=1e-11)
            np.testing.assert_allclose(_np(Es[b]), gn, rtol=1e-8, atol=1e-12)
            np.testing.assert_allclose(_np(Ei[b]), gi, rtol=1e-8, atol=1e-12)
    assert float((Es.sum(-1) - 1).abs().max()) < 1e-12
    assert float((Et.sum((-1, -2)) - (T - 1)).abs().max()) < 1e-10 * max(1, T)


def test_hmm_wide_kernel_log_space_redo_of_a_flagged_sequence_and_batched_pairs():
    """A forced transition through a log-potential of -800 underflows the scaled recursion: that sequence (1 of 5) is
    flagged and redone in log space -- the reference's own arithmetic --, the others keep the scaled pass; per-sequence
    transition matrices (pair_batched) at K = 24."""
    from svae_amd.hmm.hmm_inference import hmm_estep
    K, T, B = 24, 14, 5
    rng = np.random.default_rng(8)
    init = np.full(K, -1e4); init[0] = 0.0
    pair = -2.0 + 0.3 * rng.standard_normal((K, K))
    pair[0, :] = -1e4; pair[0, 0] = 0.0; pair[0, 1] = -800.0
    pair[1:, 0] = -1e4
    node = 0.5 * rng.standard_normal((B, T, K))
    node[3] = 0.0
    node[3, :7, 1:] = -1e4
    node[3, 7:, 0] = -1e4
    logZ, (Ei, Et, Es) = hmm_estep((init, pair, node))
    for b in range(B):
        lz, (oi, ot, os_) = hmm_numpy.hmm_estep((init, pair, node[b]))
        assert np.isfinite(lz) and float(logZ[b]) == pytest.approx(lz, rel=1e-9, abs=1e-9)
        np.testing.assert_allclose(_np(Ei[b]), oi, rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(_np(Es[b]), os_, rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(_np(Et[b]), ot, rtol=1e-7, atol=1e-9)
    assert float(logZ[3]) < -790
    pairs = np.stack([_problem(1, 1, K, rng)[1] for _ in range(B)])
    init2 = np.log(rng.dirichlet(np.ones(K)))
    logZ, (Ei, Et, Es) = hmm_estep((init2, pairs, node))
    for b in range(B):
        lz, (oi, ot, os_) = hmm_numpy.hmm_estep((init2, pairs[b], node[b]))
        assert float(logZ[b]) == pytest.approx(lz, rel=1e-10)
        np.testing.assert_allclose(_np(Et[b]), ot, rtol=1e-8, atol=1e-11)
