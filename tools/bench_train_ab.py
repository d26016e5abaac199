"""A/B of the training path (svae_lds_inference_f64 + VJP) with lean / full per-step records at several batch sizes.
usage: python tools/bench_train_ab.py [B ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from svae_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
for B in [int(x) for x in sys.argv[1:]] or [4096, 2048, 1024, 512]:
    for name, opt in (("default", None), ("lean_on", _lib.OPT_LEAN_ON), ("lean_off", _lib.OPT_LEAN_OFF)):
        r = bench.measure_training_path(dev, 200, 10, B, reps=7, options=opt)
        print(json.dumps({"B": B, "options": name, "format": r["record_format"],
                          "infer_ms": round(r["estep_and_sampler_ms"], 4), "vjp_ms": round(r["vjp_ms"], 4),
                          "ms_per_pass": round(r["ms_per_pass"], 4)}), flush=True)
