"""Time the reference's CPU E-step on this box's host cores.  TEST/BENCH INFRASTRUCTURE ONLY.

Run by bench.py's `cpu_baseline` leg as a separate process (never forked from a process that has
initialised the HIP runtime):  python -m oracle.cpu_baseline --B 512 --T 200 --n 10

Times `cython_natural_lds_estep_general` (svae/lds/lds_inference.py:232-237) = the reference's own
compiled filter + smoother built by oracle/build_ref.py ("reference"), or, when oracle/_ref is
absent, the NumPy restatement ("port"), per sequence, on a BOUNDED sample of the bench workload
(same generator and seeds as bench.py rank 0): first on 1 core, then on all USABLE host cores
(min of len(os.sched_getaffinity(0)), the cgroup CPU quota -- cpu.max / cfs_quota, which the affinity mask
does not show -- and the CPU time a pool of busy processes is MEASURED to get; not os.cpu_count()) with a
process pool (BLAS pinned to 1 thread per process) in which every process works until a COMMON
deadline, so that no straggler sets the wall time and the leg takes `budget` seconds whatever the
scaling.  Reports, next to the aggregate rate, the mean
per-process rate and the scaling efficiency = aggregate / (processes x one-core rate): SMT siblings,
shared caches and memory bandwidth make it far less than 1 on a 256-thread box, and the GPU/CPU
ratio must be read with it.  Prints one JSON object.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_G = {}


def _estep_fn():
    from oracle import lds_numpy, ref
    if ref.available():
        return "reference", ref.estep
    return "port", lds_numpy.natural_lds_estep_general


def _worker(count):
    """E-step `count` sequences (cycling through the sample); returns (elapsed seconds, count)."""
    natparam, node_J, node_h = _G["data"]
    _, est = _estep_fn()
    T = node_h.shape[1]
    z = np.zeros(T)
    B = node_h.shape[0]
    t0 = time.perf_counter()
    for i in range(count):
        b = i % B
        est(natparam, (node_J[b], node_h[b], z))
    return time.perf_counter() - t0, count


def _worker_until(deadline):
    """E-step sequences (cycling through the sample) until the system-wide monotonic clock passes `deadline`;
    returns (busy seconds, count)."""
    natparam, node_J, node_h = _G["data"]
    _, est = _estep_fn()
    z = np.zeros(node_h.shape[1])
    B = node_h.shape[0]
    t0, i = time.perf_counter(), 0
    while time.perf_counter() < deadline:
        est(natparam, (node_J[i % B], node_h[i % B], z))
        i += 1
    return time.perf_counter() - t0, i


def affinity_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _cpuset_count(text):
    n = 0
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        n += int(hi or lo) - int(lo) + 1
    return n or None


def quota_cores():
    """CPU bandwidth this container is granted by its cgroup, in cores (None = no limit visible): cgroup v2
    `cpu.max` ("<quota> <period>" | "max ..."), cgroup v1 `cpu.cfs_quota_us` / `cpu.cfs_period_us`, and the
    effective cpuset -- the limits sched_getaffinity does not show."""
    found = []
    v2 = _read("/sys/fs/cgroup/cpu.max")
    if v2:
        q, _, per = v2.partition(" ")
        if q != "max":
            found.append(float(q) / float(per or 100000))
    q1, p1 = _read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), _read("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
    if q1 and p1 and int(q1) > 0:
        found.append(float(q1) / float(p1))
    for path in ("/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpuset/cpuset.effective_cpus"):
        c = _cpuset_count(_read(path))
        if c:
            found.append(float(c))
    return min(found) if found else None


def _spin_until(deadline):
    """busy loop until the common deadline; returns the CPU seconds this process was actually given"""
    c0 = time.process_time()
    x = 0
    while time.perf_counter() < deadline:
        for _ in range(2000):
            x += 1
    return time.process_time() - c0


def granted_cores(procs, seconds=1.5):
    """MEASURED: run `procs` busy processes for `seconds` of wall time and add up the CPU time they were given --
    the cores the box really grants this container, whatever mechanism limits it (quota of a parent cgroup the
    container cannot see, SMT siblings counted as cores, other tenants)."""
    import multiprocessing as mp
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(_spin_until, [time.perf_counter() + 0.2] * procs, chunksize=1)         # start every process
        t0 = time.perf_counter()
        used = pool.map(_spin_until, [t0 + seconds] * procs, chunksize=1)
        wall = time.perf_counter() - t0
    return sum(used) / wall


def usable_cores():
    """(pool size, details): min(affinity, cgroup quota, measured grant) -- one BLAS-bound process per core the box
    actually delivers, not per hardware thread it lists."""
    aff = affinity_cores()
    quota = quota_cores()
    granted = granted_cores(aff) if aff > 1 else 1.0
    pool = aff
    if quota is not None:
        pool = min(pool, int(quota + 0.999))
    pool = max(1, min(pool, int(round(granted))))
    return pool, {"affinity_cores": aff, "quota_cores": quota, "granted_cores_measured": granted}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=512)
    ap.add_argument("--T", type=int, default=200)
    ap.add_argument("--n", type=int, default=10)
    ap.add_argument("--budget", type=float, default=10.0, help="seconds of wall time per leg")
    a = ap.parse_args()
    from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials, rotation_lds_natparam
    natparam = (rotation_lds_natparam if a.n > 15 else rand_lds_natparam)(a.n, np.random.default_rng(0))   # = bench.bench_natparam
    node_J, node_h = rand_node_potentials((a.B, a.T, a.n), np.random.default_rng(1000))
    _G["data"] = (natparam, node_J, node_h)
    kind, _ = _estep_fn()

    lo = 32 if a.n * a.n * a.T < 100000 else 4        # smallest sample (big sequences cost ~0.1 s each)
    _worker(max(1, lo // 4))                          # warm-up / page-in
    probe = _worker(lo)[0] / float(lo)                # seconds per sequence
    n1 = int(max(lo, min(a.B, a.budget / probe)))
    dt1 = _worker(n1)[0]
    one_core = n1 / dt1
    cores, core_info = usable_cores()
    out = {"value": one_core, "unit": "sequences/s", "cores": 1, "kind": kind,
           "one_core_value": one_core, "host_cores": cores, "effective_cores": cores, "os_cpu_count": os.cpu_count(),
           "core_info": core_info,
           "sample": "%d of the %d bench sequences (T=%d, n=%d), 1 core" % (n1, a.B, a.T, a.n)}
    if cores > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(cores) as pool:      # no GPU runtime in this process
            pool.map(_worker, [max(1, lo // 8)] * cores)      # warm every process
            t0 = time.perf_counter()
            res = pool.map(_worker_until, [t0 + a.budget] * cores, chunksize=1)   # one task per process
            wall = time.perf_counter() - t0
        done = sum(c for _, c in res)
        busy = sum(dt for dt, _ in res)
        allc = done / wall
        out["all_cores_value"] = allc
        out["all_cores"] = {"processes": cores, "sequences": done, "wall_s": wall,
                            "per_process_value": done / busy,                  # mean rate of a process while it runs
                            "scaling_efficiency": allc / (cores * one_core)}   # 1.0 = linear in the process count
        if allc > one_core:
            out.update(value=allc, cores=cores,
                       sample="%d sequences in %.1f s over %d processes (one per core the box grants: min of "
                              "sched_getaffinity, the cgroup CPU quota and the measured CPU time of a busy pool; common deadline), drawn from the %d bench sequences (T=%d, n=%d); "
                              "scaling efficiency %.2f vs one core" % (done, wall, cores, a.B, a.T, a.n,
                                                                       allc / (cores * one_core)))
    # the "NumPy/autograd path" north_star names (lds_inference.py:223-229): its NumPy restatement
    # (oracle/lds_numpy.py), one core, a few seconds -- reported beside the compiled path above
    from oracle import lds_numpy
    z = np.zeros(a.T)
    lds_numpy.natural_lds_estep_general(natparam, (node_J[0], node_h[0], z))
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < min(5.0, a.budget) and k < a.B:
        lds_numpy.natural_lds_estep_general(natparam, (node_J[k], node_h[k], z))
        k += 1
    out["numpy_path"] = {"value": k / (time.perf_counter() - t0), "unit": "sequences/s", "cores": 1,
                         "kind": "numpy", "sample": "%d of the bench sequences, oracle/lds_numpy.py" % k}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
