"""Random shapes through the parametrised GPU parity tests of the n <= 15 paths (E-step / sampler / filter / VJPs against the
reference's compiled code, lean against full records, HMM two-ended and wide kernels, the SLDS consumer against the table
kernel): the shapes the fixed parametrisations do not list.  Usage: python tools/fuzz_paths.py [seconds] [seed] [a|b|c]   (b: GMM, latent dimension 16 .. 64, dense node potentials; c: SLDS ascent and
its glue kernels, model-level run_inference, GMM local step against autograd; d: batch sizes across the kernel-selection
thresholds -- E-step against the reference on a few sequences, run-to-run bit equality, one-call inference against the E-step)"""
import os, sys, time, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest  # noqa: E402
import test_lds_hip as tl, test_vjp_hip as tv, test_lean_hip as tn, test_hmm_hip as th, test_slds_hip as ts  # noqa: E402
import test_gmm_hip as tg, test_lds_tile_hip as tt, test_lds_dense_hip as td  # noqa: E402
import test_models_hip as tm, test_svae_hip as tsv  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
group = sys.argv[3] if len(sys.argv) > 3 else "a"
ri = lambda a, b: int(rng.integers(a, b + 1))
un = lambda f: getattr(f, "__wrapped__", f)


def draw_b():
    kind = ri(0, 8)
    if kind == 0:
        return "gmm", un(tg.test_against_oracle), (ri(1, 3000), ri(1, 8), ri(1, 64), bool(ri(0, 1)))
    if kind == 1:
        return "gmm_global", un(tg.test_global_step_kernel_against_the_torch_maps), (ri(1, 64), ri(1, 8))
    if kind == 2:
        return "tile_oracle", un(tt.test_tile_estep_matches_oracle), (ri(16, 64), ri(1, 12))
    if kind == 3:
        return "tile_inh", un(tt.test_tile_estep_inhomogeneous_batched_pairs), (ri(16, 64), ri(2, 10), ri(1, 4))
    if kind == 4:
        return "tile_vjp_torch", un(tt.test_tile_vjp_kernels_match_the_torch_adjoint), (ri(16, 64), ri(1, 8), ri(1, 3), ri(0, 3), str(rng.choice(["homog", "inhomog", "batched"])))
    if kind == 5:
        return "tile_halves", un(tt.test_tile_estep_halves_equal_the_whole), (ri(16, 64), ri(1, 14), ri(1, 4))
    if kind == 6:
        return "tile_2wg", un(tt.test_tile_estep_two_workgroups_per_cu_instances), (ri(16, 64), ri(1, 8), bool(ri(0, 1)))
    if kind == 7:
        return "dense", un(td.test_dense_estep_filter_and_sampler_against_the_oracle), (ri(1, 15), ri(1, 40), ri(1, 6), bool(ri(0, 1)))
    return "dense_grad", un(td.test_gradients_through_dense_node_potentials), (ri(1, 10), ri(2, 14), ri(1, 3), ri(1, 2), bool(ri(0, 1)))


def draw_c():
    kind = ri(0, 7)
    if kind == 0:
        return "slds_ascent", un(ts.test_optimize_local_meanfield_matches_oracle), (ri(1, 8), ri(1, 10), ri(2, 40), ri(1, 6), bool(ri(0, 1)), bool(ri(0, 1)))
    if kind == 1:
        return "slds_contract", un(ts.test_contraction_kernels_against_the_dense_forms), (ri(1, 16), ri(1, 15), ri(2, 40), ri(1, 12))
    if kind == 2:
        return "slds_final_contract", un(ts.test_final_pass_contractions_against_the_two_library_forms), (ri(1, 8), ri(1, 10), ri(2, 60), ri(1, 40))
    if kind == 3:
        return "slds_init_path", un(ts.test_initial_sample_path_diagonal_kernel_equals_the_dense_filter_and_sampler), (ri(1, 80), ri(1, 60), ri(1, 10))
    if kind == 4:
        return "gmm_local_autograd", un(tsv.test_gmm_local_step_kernels_against_torch_autograd), (ri(1, 20), ri(1, 8), ri(1, 200), ri(1, 3))
    if kind == 5:
        return "lds_model", un(tm.test_lds_run_inference_against_oracle), (ri(1, 15), ri(1, 40), ri(1, 6), ri(1, 3))
    if kind == 6:
        return "global_step", un(tm.test_global_step_kernel_matches_the_exponential_family_maps), (ri(1, 64),)
    return "slds_maps", un(tm.test_slds_global_maps_in_one_launch_equal_one_launch_per_state), (ri(1, 16), ri(1, 15))


def _batch_case(n, T, B, S, seed):
    """well-conditioned homogeneous model, B sequences: E-step vs the reference's compiled E-step on 5 of them, twice (bit
    equality), and the one-call inference path's statistics against the E-step path's"""
    import torch
    from oracle import ref
    from svae_amd.lds.lds_inference import natural_lds_estep_general, natural_lds_inference_general
    from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
    r = np.random.default_rng(seed)
    while True:
        init, pair = rand_lds_natparam(n, r)
        if max(np.linalg.cond(np.asarray(pair[0])), np.linalg.cond(np.asarray(pair[2]))) < 1e3:
            break
    node = rand_node_potentials((B, T, n), r, with_logZ=True)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    nat = (tuple(t(x) for x in init), tuple(t(x) for x in pair))
    nd = tuple(t(x) for x in node)
    ln, (Ei, Ep, En) = natural_lds_estep_general(nat, nd)
    ln2, (Ei2, Ep2, En2) = natural_lds_estep_general(nat, nd)
    assert torch.equal(ln, ln2) and torch.equal(En[1], En2[1]) and torch.equal(Ep[1], Ep2[1]), "run-to-run"
    for b in sorted(set(int(x) for x in r.integers(0, B, 5)) | {0, B - 1}):
        wl, (wi, wp, wn) = ref.estep((init, pair), tuple(x[b] for x in node))
        assert tl._rel(ln[b], wl) < 1e-7 and tl._rel(En[1][b], wn[1]) < 1e-7 and tl._rel(En[0][b], wn[0]) < 1e-7, ("estep", b)
        assert tl._rel(Ei[0][b], wi[0]) < 1e-7 and all(tl._rel(Ep[k][b], wp[k]) < 1e-7 for k in range(3)), ("estep pair", b)
    if S:
        eps = torch.as_tensor(r.standard_normal((B, T, S, n)), device=dev)
        samples, (Ei3, Ep3, En3), ln3 = natural_lds_inference_general(nat, nd, eps=eps)
        assert tl._rel(ln3, ln.cpu().numpy()) < 1e-9 and tl._rel(En3[1], En[1].cpu().numpy()) < 1e-9, "inference vs estep"


def draw_d():
    B = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 255, 256, 257, 511, 512, 513, 700, 1023, 1024, 1025, 1026, 1500, 2047, 2048, 2049, 3000, 4095, 4096, 4097, 4300]))
    return "batch", _batch_case, (ri(1, 15), ri(1, 40), B, ri(0, 2), ri(0, 10 ** 6))


def draw():
    if group == "b":
        return draw_b()
    if group == "d":
        return draw_d()
    if group == "c":
        return draw_c()
    kind = ri(0, 9)
    if kind == 0:
        return "vjp_ref", un(tv.test_vjp_against_reference_compiled_vjps), (ri(1, 15), ri(1, 60), ri(1, 9), ri(1, 4), bool(ri(0, 1)))
    if kind == 1:
        return "vjp_inh", un(tv.test_vjp_inhomogeneous_with_statistics_cotangents), (ri(1, 15), ri(2, 40), ri(1, 6), ri(1, 3), bool(ri(0, 1)), bool(ri(0, 1)))
    if kind == 2:
        return "vjp_hom_sums", un(tv.test_vjp_homogeneous_summed_pair_statistics_cotangents), (ri(1, 15), ri(2, 40), ri(1, 6), ri(1, 3), bool(ri(0, 1)))
    if kind == 3:
        return "lean_vs_full", un(tn.test_lean_and_full_records_agree), (ri(1, 10), ri(2, 120), ri(1, 70), ri(1, 2), bool(ri(0, 1)))
    if kind == 4:
        return "lean_inh", un(tn.test_lean_forward_only_with_per_step_pair_parameters), (ri(1, 10), ri(2, 60), ri(1, 12), ri(0, 2), bool(ri(0, 1)))
    if kind == 5:
        return "sampler", un(tl.test_sampler_against_oracle), (ri(1, 15), ri(1, 40), ri(1, 20))
    if kind == 6:
        return "filter_msgs", un(tl.test_filter_messages_against_reference_build), (ri(1, 15), ri(1, 60), bool(ri(0, 1)))
    if kind == 7:
        return "smoother_on_msgs", un(tl.test_smoother_and_sampler_on_caller_supplied_messages), (ri(1, 15), ri(1, 60), bool(ri(0, 1)))
    if kind == 8:
        K = ri(1, 64)
        f = th.test_hmm_estep_against_oracle_and_reference if K <= 16 else th.test_hmm_wide_kernel_against_oracle_and_reference
        return "hmm", un(f), (ri(1, 40), ri(1, 300), K, float(rng.choice([1.0, 3.0, 10.0])))
    return "slds_consumer", None, (ri(1, 8), ri(1, 10), ri(4, 90))


# every module's _rel records what it measured, so that a failure can say by how much
seen = []
for m in (tl, tv, tn, th, ts, tg, tt, td, tm, tsv):
    if hasattr(m, "_rel"):
        def wrap(f):
            def g(*a, **k):
                v = f(*a, **k)
                seen.append(float(v))
                return v
            return g
        m._rel = wrap(m._rel)

t0, done, bad, worst = time.time(), {}, 0, {}
while time.time() - t0 < budget:
    name, fn, args = draw()
    del seen[:]
    try:
        if fn is None:
            rel, case = ts._consumer_sweep(args[0], ns=(args[1],), Ts=(args[2],))
            assert rel < 1e-10, (case, rel)
        else:
            fn(*args)
    except pytest.skip.Exception:
        continue
    except Exception:
        bad += 1
        big = max(seen) if seen else float("nan")
        worst[name] = max(worst.get(name, 0.0), big)
        print("FAIL %s %s  largest distance measured in the call: %.2e  (%s)" % (name, args, big, traceback.format_exc(limit=3).strip().splitlines()[-2].strip()[:110]), flush=True)
    done[name] = done.get(name, 0) + 1
print("cases", done, "failures", bad, "worst distance among failures", worst)
sys.exit(1 if bad else 0)
