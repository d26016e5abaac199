"""Where run_inference of the SLDS (models/slds_svae.py: ascent + final pass) spends its time at configs[3]: CUDA events
around the stages of the final pass.  Usage: python tools/slds_final_pass_timeline.py [B T n K] [--library-contractions]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd.models import slds_svae as S
from svae_amd.lds.lds_inference import LDSEStepPlan
from svae_amd.hmm.hmm_inference import hmm_estep
from svae_amd.lds.synthetic_data import rand_slds_global_natparam

_a = [x for x in sys.argv[1:] if not x.startswith('--')]
B, T, n, K = (int(x) for x in _a[:4]) if len(_a) >= 4 else (2048, 500, 10, 8)
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
glob = rand_slds_global_natparam(K, n, rng)
prior = rand_slds_global_natparam(K, n, rng)
node = (torch.as_tensor(-0.5 * (0.5 + rng.random((B, T, n))), device=dev), torch.as_tensor(2. * rng.standard_normal((B, T, n)), device=dev))
g = torch.Generator(device=dev).manual_seed(1)
init_eps = torch.randn(B, T, 1, n, dtype=torch.float64, device=dev, generator=g)
eps = torch.randn(B, T, 1, n, dtype=torch.float64, device=dev, generator=g)


def once(record):
    marks = []
    def mark(name):
        if record:
            e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
    mark("start")
    maps = S.global_to_local_maps(glob, dev)
    (hmm_stats, _), (hmm_nat, (lds_init, lds_pair)), _, _ = S.optimize_local_meanfield(glob, node, init_eps, pair_stats=False,
                                                                                      local_maps=maps)
    mark("ascent (incl. the per-step pair parameters)")
    plan = LDSEStepPlan(B, T, n, dev, inhomog=True, pair_batched=True)
    mark("plan buffers")
    lognorm, (Ei, Ep, En) = S._lds_estep_batched_init(plan, lds_init, lds_pair, node, keep_factor=True)
    mark("LDS E-step on per-step parameters, keeping the factor")
    samples = plan.sample(eps)
    mark("sampler")
    _, _, dense_init, dense_pair = maps
    fused = None if "--library-contractions" in sys.argv else \
        S.final_pass_contractions(dense_init, dense_pair, (Ei[0], Ei[1]), plan.E_pair, hmm_stats[2])
    node_hmm, pair_sums = fused if fused is not None else \
        (S.get_arhmm_local_nodeparams(dense_init, dense_pair, (Ei[0], Ei[1]), plan.E_pair), None)
    mark("HMM node potentials (+ weighted sums) from the pair statistics")
    hmm_vlb, _ = hmm_estep((hmm_nat[0], hmm_nat[1], node_hmm))
    mark("HMM E-step")
    stats = S.get_global_stats(hmm_stats, (Ei[0], Ei[1]), plan.E_pair, pair_sums)
    mark("global statistics")
    lds_vlb = lognorm - ((node[0] * En[0]).sum((1, 2)) + (node[1] * En[1]).sum((1, 2)))
    local_vlb = (hmm_vlb + lds_vlb).sum()
    gv = S.slds_prior_vlb(glob, prior, dev)
    mark("bounds")
    return marks


for _ in range(2):
    once(False)
torch.cuda.synchronize()
t0 = time.perf_counter(); m = once(True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("run_inference body B=%d T=%d n=%d K=%d: %.2f ms wall" % (B, T, n, K, dt * 1e3))
for (a, ea), (b, eb) in zip(m[:-1], m[1:]):
    print("  %-60s %7.3f ms" % (b, ea.elapsed_time(eb)))
ts = []
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = S.run_inference(prior, glob, node, 1, init_eps=init_eps, eps=eps); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("run_inference: %.2f ms (best of %s)" % (min(ts), [round(x, 2) for x in ts]))
# the fused contraction alone
_, _, dense_init, dense_pair = S.global_to_local_maps(glob, dev)
E_pair = torch.randn(B, T - 1, 3, n, n, dtype=torch.float64, device=dev)
init_stats = (torch.randn(B, n, n, dtype=torch.float64, device=dev), torch.randn(B, n, dtype=torch.float64, device=dev))
Es = torch.softmax(torch.randn(B, T, K, dtype=torch.float64, device=dev), -1)
for name, fn in (("svae_slds_pair_contract_f64", lambda: S.final_pass_contractions(dense_init, dense_pair, init_stats, E_pair, Es)),
                 ("library: node potentials", lambda: S.get_arhmm_local_nodeparams(dense_init, dense_pair, init_stats, E_pair)),
                 ("library: weighted sums", lambda: S.get_global_stats((Es[:, 0], Es[:, 0, :, None] * Es[:, 0, None], Es), init_stats, E_pair))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print("  %-40s %.3f ms per call" % (name, e0.elapsed_time(e1) / 5))
