#!/usr/bin/env python
"""Generates svae_amd/csrc/gj1r_gen.hpp: the one-register-per-row Gauss-Jordan of the two-ended LDS
E-step kernel (lds_estep_twoend.hpp) with every pivot as ONE hand-scheduled inline-asm block.

Why generated asm: hipcc inserts an s_nop between two inline-asm statements that share a register and
cannot schedule around the DPP / trans hazards inside them, which cost ~85 of ~290 issue slots per
10 x 10 elimination.  Here the row updates themselves provide the wait states:
  * VALU write of a VGPR -> DPP read of it: 2 wait states (two other row updates sit in between);
  * v_rcp_f64 (trans) result -> non-trans VALU use: 1 wait state (one row update in between);
  * the reciprocal chain of the NEXT pivot (v_rcp_f64 + two Newton steps) is interleaved with the row
    updates of the current one, dependent instructions one issue slot apart.
The asm template must be a string literal, hence a generator (python tools/gen_gj_asm.py).
"""
import os

DPP = "row_mask:0xf bank_mask:0xf"
# 1/pn to full fp64 accuracy from the v_rcp_f64 seed t0 (relative error e0 = 1 - pn t0 ~ 2^-23: "2^29 ulp"):
# ONE cubic step  rn = t0 (1 + e0 + e0^2)  -- error e0^3 ~ 2^-69 -- instead of two Newton steps (4 FMAs, 2^-92)
RCP_CHAIN = ["v_rcp_f64 %[t0], %[pn]",
             "v_fma_f64 %[e0], -%[pn], %[t0], 1.0",
             "v_fma_f64 %[e0], %[e0], %[e0], %[e0]",
             "v_fma_f64 %[rn], %[t0], %[e0], %[t0]"]


def pivot_block(N, k):
    """-> (asm lines, has_next).  Operands: %[m0..] rows, %[e] = E[k], %[p], %[ri] in; %[q], %[vf] in/out;
    %[pn], %[rn] out (next pivot, reciprocal); %[ru], %[t0], %[e0] scratch."""
    L = []
    L.append("v_fma_f64 %[ru], -%[p], %[e], %[m{k}]".format(k=k))          # lane k -> exactly 0
    L.append("v_mul_f64 %[ru], %[ru], %[ri]")                               # scaled pivot row
    L.append("v_fma_f64 %[q], %[m{k}], %[ru], %[q]".format(k=k))
    L.append("v_fma_f64 %[vf], -%[ri], %[e], %[vf]")
    upd = lambda i: "v_fmac_f64_dpp %[m{i}], %[m{i}], -%[ru] row_newbcast:{k} {d}".format(i=i, k=k, d=DPP)
    if k + 1 < N:
        others = list(range(k + 2, N)) + list(range(0, k))
        L.append(upd(k + 1))
        pre = others[:2]
        others = others[2:]
        for i in pre:
            L.append(upd(i))
        if len(pre) < 2:
            L.append("s_nop %d" % (1 - len(pre)))
        L.append("v_mov_b64_dpp %[pn], %[m{i}] row_newbcast:{i} {d}".format(i=k + 1, d=DPP))
        chain = RCP_CHAIN
        for ci, ins in enumerate(chain):
            L.append(ins)
            if ci < len(chain) - 1:
                if others:
                    L.append(upd(others.pop(0)))
                elif ci == 0:
                    L.append("s_nop 0")            # trans result -> VALU use
        for i in others:
            L.append(upd(i))
        L.append("v_add_f64 %[m{k}], %[ru], -%[e]".format(k=k))            # pivot row as kept: lane k = -1
        return L, True
    for i in range(0, N - 1):
        L.append(upd(i))
    L.append("v_add_f64 %[m{k}], %[ru], -%[e]".format(k=k))
    return L, False


def emit(N, rows=False):
    """rows: the variant that also returns every scaled pivot row (RU[k]; lanes j > k: L[j][k], the unit LDL' factor
    the backward sampler reads -- lds_filter_1r.hpp)."""
    out = []
    out.append("template <>")
    if rows:
        out.append("__device__ __forceinline__ void gauss_jordan_1r_asm_rows<%d>(double (&M)[%d], const double (&E)[%d], "
                   "double& qacc, double& vfull, double (&RU)[%d]) {" % (N, N, N, N))
    else:
        out.append("__device__ __forceinline__ void gauss_jordan_1r_asm<%d>(double (&M)[%d], const double (&E)[%d], "
                   "double& qacc, double& vfull) {" % (N, N, N))
    out.append("  double p = bcast_fenced<0>(M[0]);")
    out.append("  double rinv = rcp_nr(p);")
    out.append("  double %st0, e0, pn, rn;" % ("" if rows else "ru, "))
    out.append("  (void)t0; (void)e0; (void)pn; (void)rn;")
    for k in range(N):
        lines, has_next = pivot_block(N, k)
        body = "\\n\\t\"\n      \"".join(lines)
        rowops = ", ".join('[m%d] "+v"(M[%d])' % (i, i) for i in range(N))
        outs = rowops + ', [q] "+v"(qacc), [vf] "+v"(vfull), [ru] "=&v"(%s)' % ("RU[%d]" % k if rows else "ru")
        if has_next:
            outs += ', [t0] "=&v"(t0), [e0] "=&v"(e0), [pn] "=&v"(pn), [rn] "=&v"(rn)'
        ins = '[e] "v"(E[%d]), [p] "v"(p), [ri] "v"(rinv)' % k
        out.append("  asm volatile(\n      \"%s\"\n      : %s\n      : %s);" % (body, outs, ins))
        if has_next:
            out.append("  p = pn; rinv = rn;")
    out.append("}")
    return "\n".join(out)


def pivot_block2(N, k):
    """Two registers per row (row-per-chain layout, lds_estep_twoend_rpc.hpp): a_i = [P row | .. | h] (lanes < N, lane
    15), b_i = right-hand-side columns (lanes < N).  Row update = two DPP FMAs, both broadcasting the multiplier
    f_i = lane k of a_i: the b update first (the a update rewrites a_i; its lane k keeps f_i because ru lane k = 0).
    Operands as pivot_block plus %[b0..]; the scaled pivot row of the b set is formed in place (b_k *= 1/p_k)."""
    L = []
    L.append("v_fma_f64 %[ru], -%[p], %[e], %[a{k}]".format(k=k))          # lane k -> exactly 0
    L.append("v_mul_f64 %[b{k}], %[b{k}], %[ri]".format(k=k))               # scaled pivot row, b set (in place)
    L.append("v_mul_f64 %[ru], %[ru], %[ri]")                               # scaled pivot row, a set
    L.append("v_fma_f64 %[vf], -%[ri], %[e], %[vf]")
    L.append("v_fma_f64 %[q], %[a{k}], %[ru], %[q]".format(k=k))
    updb = lambda i: "v_fmac_f64_dpp %[b{i}], %[a{i}], -%[b{k}] row_newbcast:{k} {d}".format(i=i, k=k, d=DPP)
    upda = lambda i: "v_fmac_f64_dpp %[a{i}], %[a{i}], -%[ru] row_newbcast:{k} {d}".format(i=i, k=k, d=DPP)
    others = list(range(k + 2, N)) + list(range(0, k))
    work = []
    for i in others:
        work += [updb(i), upda(i)]
    if k + 1 < N:
        L += [updb(k + 1), upda(k + 1)]
        pre = work[:2]
        work = work[2:]
        L += pre
        if len(pre) < 2:
            L.append("s_nop %d" % (1 - len(pre)))
        L.append("v_mov_b64_dpp %[pn], %[a{i}] row_newbcast:{i} {d}".format(i=k + 1, d=DPP))
        for ci, ins in enumerate(RCP_CHAIN):
            L.append(ins)
            if ci < len(RCP_CHAIN) - 1:
                if work:
                    L.append(work.pop(0))
                elif ci == 0:
                    L.append("s_nop 0")            # trans result -> VALU use
        L += work
        L.append("v_add_f64 %[a{k}], %[ru], -%[e]".format(k=k))            # pivot row as kept: lane k = -1
        return L, True
    L += work
    L.append("v_add_f64 %[a{k}], %[ru], -%[e]".format(k=k))
    return L, False


def emit2(N):
    out = []
    out.append("template <>")
    out.append("__device__ __forceinline__ void gauss_jordan_2r_asm<%d>(double (&A)[%d], double (&B)[%d], const double (&E)[%d], "
               "double& qacc, double& vfull) {" % (N, N, N, N))
    out.append("  double p = bcast_fenced<0>(A[0]);")
    out.append("  double rinv = rcp_nr(p);")
    out.append("  double ru, t0, e0, pn, rn;")
    out.append("  (void)t0; (void)e0; (void)pn; (void)rn;")
    for k in range(N):
        lines, has_next = pivot_block2(N, k)
        body = "\\n\\t\"\n      \"".join(lines)
        rowops = ", ".join('[a%d] "+v"(A[%d])' % (i, i) for i in range(N)) + ", " + \
            ", ".join('[b%d] "+v"(B[%d])' % (i, i) for i in range(N))
        outs = rowops + ', [q] "+v"(qacc), [vf] "+v"(vfull), [ru] "=&v"(ru)'
        if has_next:
            outs += ', [t0] "=&v"(t0), [e0] "=&v"(e0), [pn] "=&v"(pn), [rn] "=&v"(rn)'
        ins = '[e] "v"(E[%d]), [p] "v"(p), [ri] "v"(rinv)' % k
        out.append("  asm volatile(\n      \"%s\"\n      : %s\n      : %s);" % (body, outs, ins))
        if has_next:
            out.append("  p = pn; rinv = rn;")
    out.append("}")
    return "\n".join(out)


HEADER = '''// gj1r_gen.hpp -- GENERATED by tools/gen_gj_asm.py; do not edit.
// In-place Gauss-Jordan on one register per row (see gauss_jordan_1r in lds_estep_twoend.hpp for the
// algorithm and the meaning of the operands), one hand-scheduled inline-asm block per pivot.
#pragma once
#include "dpp.hpp"

namespace svae {

template <int N>
__device__ __forceinline__ void gauss_jordan_1r_asm(double (&M)[N], const double (&E)[N], double& qacc,
                                                    double& vfull);
// the same, also returning the scaled pivot rows (RU[k], lanes j > k: L[j][k])
template <int N>
__device__ __forceinline__ void gauss_jordan_1r_asm_rows(double (&M)[N], const double (&E)[N], double& qacc,
                                                         double& vfull, double (&RU)[N]);
// two registers per row (row-per-chain layout, lds_estep_twoend_rpc.hpp): A = [P | h], B = right-hand-side columns
template <int N>
__device__ __forceinline__ void gauss_jordan_2r_asm(double (&A)[N], double (&B)[N], const double (&E)[N], double& qacc,
                                                    double& vfull);

'''

def generate():
    """The whole header as a string (tests/test_build_audit.py diffs it against the committed file)."""
    parts = [HEADER]
    for N in range(1, 11):
        parts.append(emit(N) + "\n\n")
        parts.append(emit(N, rows=True) + "\n\n")
        parts.append(emit2(N) + "\n\n")
    parts.append("}  // namespace svae\n")
    return "".join(parts)


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "svae_amd", "csrc", "gj1r_gen.hpp")
    with open(path, "w") as f:
        f.write(generate())
    print("wrote", path)
