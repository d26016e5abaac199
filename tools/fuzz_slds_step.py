"""Random shapes through the parity test of the fused SLDS mean-field step (tests/test_slds_hip.py::
test_fused_lds_meanfield_step_matches_materialised: every kernel variant against the materialised path, a shuffled
index list with an unused slot and a frozen sequence).  Usage: python tools/fuzz_slds_step.py [cases] [seed]"""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest  # noqa: E402
import test_slds_hip as ts  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fn = ts.test_fused_lds_meanfield_step_matches_materialised
fn = getattr(fn, "__wrapped__", fn)
bad = 0
for i in range(cases):
    K = int(rng.integers(1, 9)); n = int(rng.integers(1, 11)); T = int(rng.integers(4, 130))
    B = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 31, 33, 64, 100, 255, 256, 257, 300, 513, 600]))
    for kernel in ("rpc_mfma", "default", "rpc_ref"):
        try:
            fn(K, n, T, B, kernel)
        except pytest.skip.Exception:
            pass
        except Exception:
            bad += 1
            print("FAIL K=%d n=%d T=%d B=%d kernel=%s" % (K, n, T, B, kernel))
            traceback.print_exc(limit=2)
print("%d cases x 3 kernels, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
