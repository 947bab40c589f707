#!/usr/bin/env python
"""Generates pda_amd/csrc/pda_v4_block_asm.h: the LDS reads and MFMAs of one 64-item block of sweep4_kernel (d = 64 / 128) as ONE
inline-asm statement with a fixed software pipeline -- F fragments in flight, every s_waitcnt counted by hand.

Why: hipcc placed every B read one or two instructions in front of its MFMA whatever prefetch depth the source asked
for, so each pair of MFMAs paid an LDS round trip and the two MFMA waves of a SIMD kept the pipe 66 % busy.  sched_group_barrier
pipelines were tried first and came out differently for every small change of the surrounding source (once a regular 3-deep
pipeline, once the two accumulator chains one after the other with the reads one ahead).

    python tools/gen_v4_block_asm.py > pda_amd/csrc/pda_v4_block_asm.h
"""

F_DEFAULT = 6
F_WIDE = 6


def block(D, F, UA=1):
    NM = D // 16
    RB = 2 * D + 48
    HB = 32 * RB
    S = 2 * NM
    # loads in consumption order: fragments (k-step major, the two half-tiles alternating), the two test operands, the pop/id pair
    loads = [("frag", (s % 2) * HB + 32 * (s // 2)) for s in range(S)] + [("frag", cb * HB + 2 * D) for cb in range(2)] + [("pi", HB // 512)]
    n_mfma = S + 2
    n_loads = len(loads)
    lines = []

    def issue(i):
        kind, off = loads[i]
        if kind == "frag":
            lines.append("ds_read_b128 %%[t%d], %%[addr] offset:%d" % (i % F, off))
        else:
            lines.append("ds_read2st64_b64 %%[pi], %%[addrpi] offset1:%d" % off)

    for i in range(min(F, n_loads)):
        issue(i)
    for s in range(n_mfma):
        issued = min(s + F, n_loads)
        lines.append("s_waitcnt lgkmcnt(%d)" % (issued - (s + 1)))
        m, cb = (s // 2, s % 2) if s < S else (NM, s - S)
        for u in range(UA):
            sfx = "" if UA == 1 else "_%d" % u
            a = ("%%[a%d%s]" % (m, sfx)) if s < S else ("%%[aex%s]" % sfx)
            acc = "%%[acc%d%s]" % (cb, sfx)
            lines.append("v_mfma_f32_32x32x16_bf16 %s, %s, %%[t%d], %s" % (acc, a, s % F, "0" if m == 0 else acc))
        if s + F < n_loads:
            issue(s + F)
    lines.append("s_waitcnt lgkmcnt(0)")
    # XDL write -> VALU read of the accumulators: 18 wait states for a 16-pass MFMA; the compiler cannot see through the statement
    lines.append("s_nop 15")
    lines.append("s_nop 3")
    return NM, lines


def emit(D, F):
    NM, lines = block(D, F)
    out = []
    out.append("template <>")
    out.append("struct BlockAsm<%d> {" % D)
    out.append("    static constexpr int kFragmentsInFlight = %d;" % F)
    out.append("    // addr: LDS byte address of the lane's B fragment (half-tile 0, k-step 0) of the block; addr_pi: that of its (pop, id) pair")
    out.append("    static __device__ __forceinline__ void run(f32x16& acc0, f32x16& acc1, u32x4& pi, const u32x4 (&ah)[%d], const u32x4& aex, unsigned addr," % NM)
    out.append("                                               unsigned addr_pi) {")
    out.append("#if defined(__HIP_DEVICE_COMPILE__)")
    out.append("        u32x4 " + ", ".join("t%d" % i for i in range(F)) + ";")
    out.append("        asm volatile(")
    for l in lines:
        out.append('            "%s\\n\\t"' % l)
    outs = ['[acc0] "=&v"(acc0)', '[acc1] "=&v"(acc1)', '[pi] "=&v"(pi)'] + ['[t%d] "=&v"(t%d)' % (i, i) for i in range(F)]
    ins = ['[a%d] "v"(ah[%d])' % (m, m) for m in range(NM)] + ['[aex] "v"(aex)', '[addr] "v"(addr)', '[addrpi] "v"(addr_pi)']
    out.append("            : " + ", ".join(outs))
    out.append("            : " + ", ".join(ins))
    out.append('            : "memory");')
    out.append("#endif")
    out.append("    }")
    out.append("};")
    return "\n".join(out)


def block_wide(D, F):
    """one 32-item half-tile, TWO A operands (64 user rows) per B read: 2 NM + 2 MFMAs on two chains, NM + 2 LDS reads"""
    NM = D // 16
    loads = [("frag", 32 * m) for m in range(NM)] + [("frag", 2 * D)] + [("pi", 0)]
    n_steps = NM + 1
    n_loads = len(loads)
    lines = []

    def issue(i):
        kind, off = loads[i]
        if kind == "frag":
            lines.append("ds_read_b128 %%[t%d], %%[addr] offset:%d" % (i % F, off))
        else:
            lines.append("ds_read_b64 %[pi], %[addrpi]")

    for i in range(min(F, n_loads)):
        issue(i)
    for s in range(n_steps):
        issued = min(s + F, n_loads)
        lines.append("s_waitcnt lgkmcnt(%d)" % (issued - (s + 1)))
        for u in range(2):
            a = ("%%[a%d_%d]" % (s, u)) if s < NM else ("%%[aex_%d]" % u)
            lines.append("v_mfma_f32_32x32x16_bf16 %%[acc%d], %s, %%[t%d], %s" % (u, a, s % F, "0" if s == 0 else "%%[acc%d]" % u))
        if s + F < n_loads:
            issue(s + F)
    lines.append("s_waitcnt lgkmcnt(0)")
    lines.append("s_nop 15")
    lines.append("s_nop 3")
    return NM, lines


def emit2(D, F):
    """the wide geometry (512 users per workgroup)"""
    NM, lines = block_wide(D, F)
    out = []
    out.append("template <>")
    out.append("struct BlockAsm2<%d> {" % D)
    out.append("    static constexpr int kFragmentsInFlight = %d;" % F)
    out.append("    // one 32-item half-tile against 64 user rows: acc0 / acc1 = the two row sets; every B fragment feeds two MFMAs; pi = (pop, id)")
    out.append("    static __device__ __forceinline__ void run(f32x16& acc0, f32x16& acc1, u32x2& pi, const u32x4 (&ah)[2][%d], const u32x4 (&aex)[2], unsigned addr," % NM)
    out.append("                                               unsigned addr_pi) {")
    out.append("#if defined(__HIP_DEVICE_COMPILE__)")
    out.append("        u32x4 " + ", ".join("t%d" % i for i in range(F)) + ";")
    out.append("        asm volatile(")
    for l in lines:
        out.append('            "%s\\n\\t"' % l)
    outs = ['[acc0] "=&v"(acc0)', '[acc1] "=&v"(acc1)', '[pi] "=&v"(pi)'] + ['[t%d] "=&v"(t%d)' % (i, i) for i in range(F)]
    ins = ['[a%d_%d] "v"(ah[%d][%d])' % (m, u, u, m) for u in range(2) for m in range(NM)] + ['[aex_%d] "v"(aex[%d])' % (u, u) for u in range(2)] + \
          ['[addr] "v"(addr)', '[addrpi] "v"(addr_pi)']
    out.append("            : " + ", ".join(outs))
    out.append("            : " + ", ".join(ins))
    out.append('            : "memory");')
    out.append("#endif")
    out.append("    }")
    out.append("};")
    return "\n".join(out)


def block_wide_pf(D, pre, nxt):
    """block_wide with cross-block prefetch (d = 128: nine fragment reads per block, six registers).  Registers p0..p2 carry own
    fragments 0..2 and 6..8, q0..q2 own fragments 3..5 -- and, behind them (NXT), fragments 0..2 of the NEXT block, read under
    this block's last MFMAs: the caller passes them as p0..p2 of the next block (PRE), i.e. the two triples swap roles from block to
    block.  Every wait is counted on the simulated issue order; the statement ends with lgkmcnt(0): nothing is pending when the
    compiler gets the registers back."""
    NM = D // 16
    assert NM == 8
    own = [("frag", 32 * m) for m in range(NM)] + [("frag", 2 * D)] + [("pi", 0)]
    n_frag = NM + 1
    reg = lambda i: ("p%d" % (i % 3)) if (i % 6) < 3 else ("q%d" % (i % 3))
    lines, order = [], []

    def issue_own(i):
        kind, off = own[i]
        lines.append(("ds_read_b128 %%[%s], %%[addr] offset:%d" % (reg(i), off)) if kind == "frag" else "ds_read_b64 %[pi], %[addrpi]")
        order.append(("own", i))

    def issue_next(j):
        lines.append("ds_read_b128 %%[q%d], %%[addrn] offset:%d" % (j, own[j][1]))
        order.append(("next", j))

    for i in range(3 if pre else 0, 6):
        issue_own(i)
    nj = 0
    for s in range(n_frag):
        if not (pre and s < 3):
            pos = order.index(("own", s))
            lines.append("s_waitcnt lgkmcnt(%d)" % (len(order) - (pos + 1)))
        for u in range(2):
            a = ("%%[a%d_%d]" % (s, u)) if s < NM else ("%%[aex_%d]" % u)
            lines.append("v_mfma_f32_32x32x16_bf16 %%[acc%d], %s, %%[%s], %s" % (u, a, reg(s), "0" if s == 0 else "%%[acc%d]" % u))
        if s + 6 < len(own):
            issue_own(s + 6)
        if nxt and nj < 3 and s >= 3 + nj:          # q(nj) held own fragment 3 + nj: free once its MFMAs have been issued
            issue_next(nj)
            nj += 1
    assert not nxt or nj == 3
    lines.append("s_waitcnt lgkmcnt(0)")
    lines.append("s_nop 15")
    lines.append("s_nop 3")
    return NM, lines


def emit2pf(D, pre, nxt):
    NM, lines = block_wide_pf(D, pre, nxt)
    out = []
    out.append("template <>")
    out.append("struct BlockAsm2P<%d, %s, %s> {" % (D, "true" if pre else "false", "true" if nxt else "false"))
    out.append("    // p0..p2: PRE: fragments 0..2 of this block on entry.  q0..q2: NXT: fragments 0..2 of the next block on exit.")
    out.append("    static __device__ __forceinline__ void run(f32x16& acc0, f32x16& acc1, u32x2& pi, u32x4& p0, u32x4& p1, u32x4& p2, u32x4& q0, u32x4& q1, u32x4& q2,")
    out.append("                                               const u32x4 (&ah)[2][%d], const u32x4 (&aex)[2], unsigned addr, unsigned addr_pi, unsigned addrn) {" % NM)
    out.append("#if defined(__HIP_DEVICE_COMPILE__)")
    out.append("        asm volatile(")
    for l in lines:
        out.append('            "%s\\n\\t"' % l)
    outs = ['[acc0] "=&v"(acc0)', '[acc1] "=&v"(acc1)', '[pi] "=&v"(pi)'] + ['[p%d] "+v"(p%d)' % (i, i) for i in range(3)] + ['[q%d] "=&v"(q%d)' % (i, i) for i in range(3)]
    ins = ['[a%d_%d] "v"(ah[%d][%d])' % (m, u, u, m) for u in range(2) for m in range(NM)] + ['[aex_%d] "v"(aex[%d])' % (u, u) for u in range(2)] + \
          ['[addr] "v"(addr)', '[addrpi] "v"(addr_pi)', '[addrn] "v"(addrn)']
    out.append("            : " + ", ".join(outs))
    out.append("            : " + ", ".join(ins))
    out.append('            : "memory");')
    out.append("#endif")
    out.append("    }")
    out.append("};")
    return "\n".join(out)


ACC_BASE = 136        # hard accumulator registers of BlockAsm2D: v[136:151] (row set 0), v[152:167] (row set 1) of the 168 a wave of the
                      # wide geometry has (12 waves per CU)


def block_wide_double(D):
    """TWO 32-item half-tiles against 64 user rows in one statement (the wide geometry, d = 128): 36 MFMAs, the accumulators in hard
    registers that never leave the statement -- what leaves is, per half-tile and row set, the OR of the accumulators' bit patterns
    (sign bit = "some pair of this lane passed the folded test"), and the (pop, id) pairs.  A block with a set sign bit (next to none
    in a dense sweep in visiting order) is scored again on its own by BlockAsm2.  p0..p2 hold fragments 0..2 of the first half-tile on
    entry and those of the NEXT tile's first half on exit (18 fragment reads per statement: the ring of six registers comes round)."""
    NM = D // 16
    assert NM == 8
    n_frag = 2 * (NM + 1)
    frag_off = lambda i: 32 * (i % (NM + 1)) if (i % (NM + 1)) < NM else 2 * D
    reg = lambda i: ("p%d" % (i % 3)) if (i % 6) < 3 else ("q%d" % (i % 3))
    acc = lambda u: "v[%d:%d]" % (ACC_BASE + 16 * u, ACC_BASE + 16 * u + 15)
    lines, order = [], []

    def issue_own(i):
        lines.append("ds_read_b128 %%[%s], %%[addr%d] offset:%d" % (reg(i), i // (NM + 1), frag_off(i)))
        order.append(("own", i))

    def issue_pi(hh):
        lines.append("ds_read_b64 %%[pi%d], %%[addrpi%d]" % (hh, hh))
        order.append(("pi", hh))

    def issue_next(j):
        lines.append("ds_read_b128 %%[p%d], %%[addrn] offset:%d" % (j, frag_off(j)))
        order.append(("next", j))

    def or_phase(hh):
        lines.append("s_nop 15")          # XDL write -> VALU read of the accumulators
        lines.append("s_nop 3")
        for u in range(2):
            b = ACC_BASE + 16 * u
            lines.append("v_or_b32 %%[mo%d%d], v%d, v%d" % (hh, u, b, b + 1))
            for r in range(2, 16, 2):
                lines.append("v_or3_b32 %%[mo%d%d], v%d, v%d, %%[mo%d%d]" % (hh, u, b + r, b + r + 1, hh, u))

    for i in range(3, 6):
        issue_own(i)
    nj = 0
    for s in range(n_frag):
        hh, m = s // (NM + 1), s % (NM + 1)
        if s >= 3:
            pos = order.index(("own", s))
            lines.append("s_waitcnt lgkmcnt(%d)" % (len(order) - (pos + 1)))
        for u in range(2):
            a = ("%%[a%d_%d]" % (m, u)) if m < NM else ("%%[aex_%d]" % u)
            lines.append("v_mfma_f32_32x32x16_bf16 %s, %s, %%[%s], %s" % (acc(u), a, reg(s), "0" if m == 0 else acc(u)))
        if s + 6 < n_frag:
            issue_own(s + 6)
        if s == 2:
            issue_pi(0)
        if s == 10:
            issue_pi(1)
        if nj < 3 and s >= 12 + nj:         # p(nj) held own fragment 12 + nj
            issue_next(nj)
            nj += 1
        if m == NM:
            or_phase(hh)
    assert nj == 3
    lines.append("s_waitcnt lgkmcnt(0)")
    return NM, lines


def emit2d(D):
    NM, lines = block_wide_double(D)
    out = []
    out.append("template <int D>")
    out.append("struct BlockAsm2D;")
    out.append("template <>")
    out.append("struct BlockAsm2D<%d> {" % D)
    out.append("    static constexpr int kAccBase = %d;      // v[kAccBase .. kAccBase + 31] are clobbered" % ACC_BASE)
    out.append("    static __device__ __forceinline__ void run(unsigned& mo00, unsigned& mo01, unsigned& mo10, unsigned& mo11, u32x2& pi0, u32x2& pi1,")
    out.append("                                               u32x4& p0, u32x4& p1, u32x4& p2, const u32x4 (&ah)[2][%d], const u32x4 (&aex)[2], unsigned addr0," % NM)
    out.append("                                               unsigned addr1, unsigned addr_pi0, unsigned addr_pi1, unsigned addrn) {")
    out.append("#if defined(__HIP_DEVICE_COMPILE__)")
    out.append("        u32x4 q0, q1, q2;")
    out.append("        asm volatile(")
    for l in lines:
        out.append('            "%s\\n\\t"' % l)
    outs = ['[mo%d%d] "=&v"(mo%d%d)' % (hh, u, hh, u) for hh in range(2) for u in range(2)] + ['[pi0] "=&v"(pi0)', '[pi1] "=&v"(pi1)'] + \
           ['[p%d] "+v"(p%d)' % (i, i) for i in range(3)] + ['[q%d] "=&v"(q%d)' % (i, i) for i in range(3)]
    ins = ['[a%d_%d] "v"(ah[%d][%d])' % (m, u, u, m) for u in range(2) for m in range(NM)] + ['[aex_%d] "v"(aex[%d])' % (u, u) for u in range(2)] + \
          ['[addr0] "v"(addr0)', '[addr1] "v"(addr1)', '[addrpi0] "v"(addr_pi0)', '[addrpi1] "v"(addr_pi1)', '[addrn] "v"(addrn)']
    clob = ['"memory"'] + ['"v%d"' % r for r in range(ACC_BASE, ACC_BASE + 32)]
    out.append("            : " + ", ".join(outs))
    out.append("            : " + ", ".join(ins))
    out.append("            : " + ", ".join(clob) + ");")
    out.append("#endif")
    out.append("    }")
    out.append("};")
    return "\n".join(out)


def main():
    print("// GENERATED by tools/gen_v4_block_asm.py -- do not edit.  One 64-item block of sweep4_kernel: %d fragments in flight." % F_DEFAULT)
    print("// acc0 / acc1: the two half-tiles' accumulators (the folded test included); pi = (pop, id) of half-tile 0 in .xy, of half-tile 1 in .zw.")
    print("#pragma once")
    print("template <int D>")
    print("struct BlockAsm;")
    for D in (64, 128, 256):
        print(emit(D, F_DEFAULT))
    print("typedef unsigned u32x2 __attribute__((ext_vector_type(2)));")
    print("template <int D>")
    print("struct BlockAsm2;")
    for D in (64, 128):
        print(emit2(D, F_WIDE))
    assert F_WIDE == 6
    print("// the wide block with cross-block prefetch (see block_wide_pf in the generator)")
    print("template <int D, bool PRE, bool NXT>")
    print("struct BlockAsm2P;")
    for D in (128,):                 # (d = 64 has five fragment reads per block: it keeps BlockAsm2)
        # only <true, true> is used: with self-loading variants beside it in the loop hipcc spilled 108 registers.  The kernel reads
        # fragments 0..2 of its first block with BlockAsm2Pro and waits for block b + 1 to have landed before it runs block b.
        print(emit2pf(D, True, True))
        print("template <int D>")
        print("struct BlockAsm2Pro;")
        print("template <>")
        print("struct BlockAsm2Pro<%d> {" % D)
        print("    static __device__ __forceinline__ void run(u32x4& p0, u32x4& p1, u32x4& p2, unsigned addr) {")
        print("#if defined(__HIP_DEVICE_COMPILE__)")
        print('        asm volatile("ds_read_b128 %0, %3\\n\\tds_read_b128 %1, %3 offset:32\\n\\tds_read_b128 %2, %3 offset:64\\n\\ts_waitcnt lgkmcnt(0)"')
        print('                     : "=&v"(p0), "=&v"(p1), "=&v"(p2) : "v"(addr) : "memory");')
        print("#endif")
        print("    }")
        print("};")
        print(emit2d(D))


if __name__ == "__main__":
    main()
