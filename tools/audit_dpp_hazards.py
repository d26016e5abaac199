#!/usr/bin/env python
"""Static audit of gfx950 assembly for the two DPP hazards hipcc cannot pad around inline asm
(svae_amd/csrc/dpp.hpp):

  H1  a VALU instruction writes VGPR v, and a DPP instruction reads v as its DPP operand (src0, or
      the tied `old`/vdst of a DPP mov) fewer than 2 wait states later;
  H2  a VALU instruction writes EXEC (v_cmpx*) fewer than 5 wait states before a DPP instruction.

Wait states: every instruction issues in one wait state; `s_nop N` supplies N+1.  The scan is
linear per function; at a label (other predecessors possible) the history is reset to "unknown
writer of everything" unless the block itself supplies the wait states.

usage: audit_dpp_hazards.py file.s [...]     exit code 1 if any hazard is found.
"""
import re
import sys

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def audit(path):
    problems = []
    func = None
    hist = []          # list of (wait_states, written_vgprs or None(=unknown), writes_exec)
    n_dpp = 0
    for ln, line in enumerate(open(path), 1):
        t = line.split(";")[0].strip()
        if not t:
            continue
        if t.endswith(":"):
            if not t.startswith(".L"):
                func = t[:-1]
            hist = [(0, None, True)]        # unknown predecessor state
            continue
        if t.startswith("."):
            continue
        parts = t.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        if "_dpp" in op:
            n_dpp += 1
            src = regs(args[1]) if len(args) > 1 else set()
            if op.startswith("v_mov"):
                src |= regs(args[0])                 # tied old operand
            ws = 0
            for w, wr, wex in reversed(hist):
                if ws < 2 and (wr is None or (wr & src)):
                    problems.append("%s:%d [%s] H1: DPP read of v%s %d wait state(s) after its write: %s"
                                    % (path, ln, func, sorted(src), ws, t))
                    break
                if ws < 5 and wex and wr is not None:
                    problems.append("%s:%d [%s] H2: DPP %d wait state(s) after an EXEC write: %s"
                                    % (path, ln, func, ws, t))
                    break
                ws += w
                if ws >= 5:
                    break
        if op == "s_nop":
            hist.append((int(args[0], 0) + 1, set(), False))
        elif op.startswith("v_"):
            wex = op.startswith("v_cmpx")
            wr = set() if op.startswith("v_cmp") or op.startswith("v_readlane") or \
                op.startswith("v_readfirstlane") else (regs(args[0]) if args else set())
            hist.append((1, wr, wex))
        else:
            hist.append((1, set(), False))
        hist = hist[-8:]
    return n_dpp, problems


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        n, probs = audit(p)
        print("%s: %d DPP instructions, %d hazards" % (p, n, len(probs)))
        for q in probs[:20]:
            print("  " + q)
        bad += len(probs)
    sys.exit(1 if bad else 0)
