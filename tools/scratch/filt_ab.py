import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from svae_amd import _lib
from svae_amd.lds.lds_inference import LDSEStepPlan, natural_lds_sample, natural_lds_inference_general
dev = torch.device("cuda:0")
for B, T in ((2048, 500), (4096, 200), (1024, 500)):
    n = 10
    g = torch.Generator().manual_seed(0)
    nJ = (-0.5 - torch.rand(B, T, n, dtype=torch.float64, generator=g)).to(dev)
    nh = torch.randn(B, T, n, dtype=torch.float64, generator=g).to(dev)
    eps = torch.randn(B, T, 1, n, dtype=torch.float64, generator=g).to(dev)
    eye = torch.eye(n, dtype=torch.float64, device=dev); A = 0.9 * eye
    z = torch.zeros((), dtype=torch.float64, device=dev)
    natparam = ((-0.5 * eye, torch.zeros(n, dtype=torch.float64, device=dev), z), (-0.5 * A.T @ A, A.T.contiguous(), -0.5 * eye, z))
    def timeit(f, k=5):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k): out = f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k, out
    res = {}
    for name in ("auto", "twoend_seq"):
        plan = LDSEStepPlan(B, T, n, dev, options=_lib.KERNEL_OPTIONS[name])
        ms, x = timeit(lambda: natural_lds_sample(natparam, (nJ, nh), eps=eps, plan=plan))
        res[name] = x
        print("B=%d T=%d filter+sample [%s]: %.3f ms" % (B, T, name, ms))
    plan = LDSEStepPlan(B, T, n, dev)
    ms, out = timeit(lambda: natural_lds_inference_general(natparam, (nJ, nh), eps=eps, plan=plan))
    print("B=%d T=%d E-step+sample: %.3f ms; diff between filter paths %.2e" % (B, T, ms, float((res["auto"] - res["twoend_seq"]).abs().max())))
