"""The GMM workload of bench.py's extras (BASELINE configs[0]: K = 5, 2-D, 1000 points) for the profiler: 20 fixed points
(global-step kernel + persistent fixed-point kernel + statistics kernel) and 20 training steps (+ sampler, adjoint)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda:0")
print(bench.measure_gmm(dev))
print(bench.measure_gmm_training(dev))
