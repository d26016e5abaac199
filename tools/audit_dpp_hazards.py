#!/usr/bin/env python
"""Static audit of gfx950 assembly for the two DPP hazards hipcc cannot pad around inline asm
(svae_amd/csrc/dpp.hpp):

  H1  a VALU instruction writes VGPR v, and a DPP instruction reads v as its DPP operand (src0, or
      the tied `old`/vdst of a DPP mov) fewer than 2 wait states later;
  H2  a VALU instruction writes EXEC (v_cmpx*) fewer than 5 wait states before a DPP instruction;
  H3  a VALU instruction writes VGPR v, and v_permlane{16,32}_swap reads v fewer than 2 wait states later
      (hipcc pads this for values it produced itself, not for values produced by inline asm);
  H4  a transcendental instruction (v_rcp/v_rsq/v_sqrt/v_exp/v_log/v_sin/v_cos) writes VGPR v and the
      very next instruction is a non-transcendental VALU instruction reading v (gfx940+: 1 wait state).

Wait states: every instruction issues in one wait state; `s_nop N` supplies N+1.  The scan is
linear per function; at a label (other predecessors possible) the history is reset to "unknown
writer of everything" unless the block itself supplies the wait states.

A DPP instruction the compiler emitted itself (outside ;;#ASMSTART .. ;;#ASMEND: the update_dpp
intrinsic of dpp.hpp's bcast<>) is padded by hipcc against producers it knows; for those only a
producer INSIDE an inline-asm block (invisible to the hazard recogniser) is reported.

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


TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def is_trans(op):
    return op.startswith(TRANS)


def audit(path):
    problems = []
    func = None
    hist = []          # list of (wait_states, written_vgprs or None(=unknown), writes_exec)
    n_dpp = 0
    in_asm = False
    for ln, line in enumerate(open(path), 1):
        if ";;#ASMSTART" in line:
            in_asm = True
        elif ";;#ASMEND" in line:
            in_asm = False
        t = line.split(";")[0].strip()
        if not t:
            continue
        if t.endswith(":"):
            if not t.startswith(".L"):
                func = t[:-1]
            hist = [(0, None, True, None, False)]        # unknown predecessor state
            continue
        if t.startswith("."):
            continue
        parts = t.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        is_swap = op.startswith("v_permlane") and "swap" in op
        if op.startswith("v_") and not is_trans(op) and hist and hist[-1][3] is not None:
            used = set()
            for a_ in args[1:]:
                used |= regs(a_)
            if op.startswith("v_fmac") or "_dpp" in op or is_swap:
                used |= regs(args[0])
            if True:
                if hist[-1][3] & used:
                    problems.append("%s:%d [%s] H4: VALU use of v%s right after the transcendental that wrote it: %s"
                                    % (path, ln, func, sorted(hist[-1][3] & used), t))
        if "_dpp" in op or is_swap:
            n_dpp += 1 if "_dpp" in op else 0
            src = regs(args[1]) if len(args) > 1 else set()
            if (op.startswith("v_mov") and "bound_ctrl:1" not in t) or is_swap:
                src |= regs(args[0])                 # tied old operand (unless every lane is written) / both operands of a swap
            ws = 0
            for w, wr, wex, _tr, w_asm in reversed(hist):
                if not in_asm and not w_asm:       # compiler consumer, compiler (or unknown) producer: padded by hipcc
                    ws += w
                    if ws >= 5:
                        break
                    continue
                if ws < 2 and (wr is None or (wr & src)):
                    problems.append("%s:%d [%s] %s read of v%s %d wait state(s) after its write: %s"
                                    % (path, ln, func, "H3: permlane-swap" if is_swap else "H1: DPP", sorted(src), ws, t))
                    break
                if not is_swap and ws < 5 and wex and wr is not None:
                    problems.append("%s:%d [%s] H2: DPP %d wait state(s) after an EXEC write: %s"
                                    % (path, ln, func, ws, t))
                    break
                ws += w
                if ws >= 5:
                    break
        if op == "s_nop":
            hist.append((int(args[0], 0) + 1, set(), False, None, in_asm))
        elif op.startswith("v_"):
            wex = op.startswith("v_cmpx")
            wr = set() if op.startswith("v_cmp") or op.startswith("v_readlane") or \
                op.startswith("v_readfirstlane") else (regs(args[0]) if args else set())
            if is_swap and len(args) > 1:
                wr |= regs(args[1])
            hist.append((1, wr, wex, wr if is_trans(op) else None, in_asm))
        else:
            hist.append((1, set(), False, None, in_asm))
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
