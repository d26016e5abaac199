#!/usr/bin/env python
"""Lists, per kernel of an hipcc -S listing, the `s_waitcnt vmcnt(0)` that sit INSIDE a loop within a few instructions
of a global load: the signature of a request that was meant to stay in flight (a one-step-ahead prefetch) but is waited
for at once -- hipcc does that at the end of a conditional block that contains the load, and `__syncthreads()` does it
for every outstanding memory operation.  Usage: python tools/audit_load_waits.py file.s [kernel-name-substring]

It is a reading aid, not a pass/fail check: a wait right after a load is fine where the value is needed at once."""
import re
import sys


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name is not None:
            body.append(line.rstrip("\n"))
            if "s_endpgm" in line:
                yield name, body
                name, body = None, []


def audit(body, near=12):
    """-> (loads in loops, [(line index, loads pending nearby)])"""
    in_loop = [False] * len(body)
    # a block is "in a loop" if hipcc's comment says so
    cur = False
    for i, l in enumerate(body):
        if re.match(r"^\.LBB\d+_\d+:", l):
            cur = "Loop" in l or "in Loop" in l
        in_loop[i] = cur
    hits, nloads = [], 0
    last_load = -10 ** 9
    for i, l in enumerate(body):
        s = l.strip()
        if s.startswith(";") or not s:
            continue
        if s.startswith(("global_load", "buffer_load", "flat_load")):
            last_load = i
            nloads += in_loop[i]
        elif s.startswith("s_waitcnt") and "vmcnt(0)" in s and in_loop[i] and i - last_load <= near:
            hits.append((i, i - last_load))
    return nloads, hits


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, body in kernels(path):
        if want not in name:
            continue
        nloads, hits = audit(body)
        if nloads:
            print("%-100s loads in loops %4d   vmcnt(0) within 12 lines of a load: %d" % (name[:100], nloads, len(hits)))


if __name__ == "__main__":
    main()
