"""The end-to-end LDS-SVAE training step through make_gradfun (bench.py extra[9]) on its own, eager, for profiling
(tools/prof_generic.sh r5_gradfun python tools/bench_gradfun.py): which kernels the 1.6 - 1.9 ms are."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
print(json.dumps(bench.measure_gradfun_step(dev, reps=15)))
