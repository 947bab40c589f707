#!/usr/bin/env python
"""Generates pda_amd/csrc/pda_v6_loop_asm.h: the main loop of the "huge" geometry of the sweep (sweep5_kernel, pda_v5_sweep.h) on the
16 x 16 x 32 MFMA shape -- d = 64 / 128, popularity head, dense sweep in visiting order.  Same protocol as tools/gen_v5_loop_asm.py (one
inline-asm statement per (re-)entry, four waves in step, one s_barrier per 32-item half-tile, shared flag words, LDS-DMA by the MFMA waves
themselves; read that file's header first); what changes is the matrix instruction and with it the register mapping.

Why: the block loop is power-limited (profiles/round4a_c3_pmc.txt: pipe 79 % busy at 1.70 GHz), and a pure stream of
v_mfma_f32_16x16x32_bf16 delivers 2 005 TFLOP/s on this chip where the stream of v_mfma_f32_32x32x16_bf16 in sweep5's operand
configuration delivers 1 730 (tools/ubench/mfma_pure6.hip): half the accumulator registers read and written per MAC.

The mapping: a wave owns 256 users as SIXTEEN B operands of 16 users (lane l: user 16 u + (l & 15), elements 32 k + 8 (l >> 4) .. + 7;
a[0 .. 64 NK)), NK = d / 32 k-steps.  A 32-item half-tile is TWO A operands of 16 items per k-step (lane l: row 16 ib + (l & 15), chunk
4 k + (l >> 4) of the same XOR-swizzled LDS image the 32 x 32 loop reads: the image does not change).  32 accumulator chains of 4
VGPRs (u, ib) in v[128:255]: a lane holds 4 items of ONE user per chain, so the threshold test stays a per-lane compare -- per user block
three v_max3 over the 8 accumulator registers of its two chains, then max3(m, b3, pmax - ct), + ct, compare: 7 VALU per 16 users.

Order of the 32 NK MFMAs of a half-tile: two GROUPS of eight user blocks; for g: for k: for ib: for j -- every fragment read from the LDS
feeds EIGHT MFMAs in a row, a chain is touched every 16 slots and idle n_half - 16 (NK - 1) slots before its restart (80 of 128 at d = 128):
its maxima are spread over that window.

    python tools/gen_v6_loop_asm.py > pda_amd/csrc/pda_v6_loop_asm.h

V5_VARIANT=<list> builds timing-only variants as in gen_v5_loop_asm.py.  V5_LOADS=1 prints the fillers per MFMA slot.
"""
import os
import sys

VARIANT = set(filter(None, os.environ.get("V5_VARIANT", "").split(",")))     # timing-only A/B knobs; the product build has none

GU, IB = 8, 2                        # user blocks per group; 16-item blocks per half-tile
NSLOT = 8
PFD = int(os.environ.get("V5_PFD", "4"))     # half-tile h issues the pieces of h + PFD
RD = int(os.environ.get("V6_RD", "10"))      # slots between a chain's last MFMA and the first VALU read of its accumulator


def gen(D, UB):
    """UB = 16: a wave owns 256 users (one 1 024-user workgroup per CU, 512 registers per wave).  UB = 8: 128 users (512-user
    workgroups, TWO per CU, 256 registers per wave: while one wave's VALU tests run, the other wave of the SIMD has the matrix pipe)."""
    NK = D // 32
    if UB == 16:
        ACC0, FRAG0, THR0, M0T, MISC0 = 128, 96, 80, 64, 34
        NSETK = NK                       # fragment register sets per 16-item block: the whole half-tile
    else:
        ACC0, FRAG0, THR0, M0T, MISC0 = 64, 48, 40, 32, 8
        NSETK = min(NK, 2)               # two k-steps in registers: fragment (k, ib) is read while k - 2 .. k - 1 run
    LO_CLOBBER = MISC0
    # two quads (one per half-tile parity) holding ct in all four registers: the C operand of a chain's first MFMA | ... | meta pairs
    # (pmax, nmax)[2] | address temporaries[2]
    CTQ0, VFLAG, VPUB, VRD, VSB, VFB, VOFF0, VGOFF0, VZERO, VRDB, META0, ATMP0 = (MISC0 + x for x in (0, 8, 9, 10, 11, 12, 13, 14, 16, 17, 18, 22))
    if D == 256:
        # d = 256 (bf16 tables of config 5): UB = 8 user blocks of 16 x NK = 8 k-steps fill the 256 AGPRs; ONE 512-user workgroup per CU (512
        # registers per wave), four LDS-DMA pieces per wave and half-tile: their lane offsets live above the accumulators
        assert UB == 8
        VGOFF0 = 128
    assert MISC0 + 24 <= M0T and M0T + UB <= THR0 and THR0 + UB <= FRAG0 and FRAG0 + 8 * NSETK <= ACC0

    def acc(u, ib):
        c = ACC0 + 4 * (2 * u + ib)
        return "v[%d:%d]" % (c, c + 3)

    def accr(u, ib, r):
        return "v%d" % (ACC0 + 4 * (2 * u + ib) + r)

    HB = 32 * 2 * D                      # one half-tile: 32 rows of 2 D bytes, 16-byte chunks XOR-swizzled (no padding)
    # LDS slot: the rows, then the half-tile's meta entry (pmax, nmax, 0, 0).  Slots start at multiples of 256 -- of 512 at d = 256: the
    # fragment addresses are formed by XOR with k << 6 (frag_read), which reaches bit 8 from k = 4 on
    SS = HB + (512 if D == 256 else 256)
    PW = HB // 1024 // 4                 # LDS-DMA pieces per wave and half-tile (2 at d = 128, 1 at d = 64)
    OPS = 0 if "nodma" in VARIANT else PW + 1   # vector-memory operations per half-tile, ALL of them LDS-DMA (in order among themselves)
    G = UB // GU
    n_half = NK * IB * UB                # MFMA slots per half-tile (16 cycles each)
    usr = lambda u, k: "a[%d:%d]" % (4 * (u * NK + k), 4 * (u * NK + k) + 3)
    frag = lambda k, ib: "v[%d:%d]" % (FRAG0 + 4 * (2 * (k % NSETK) + ib), FRAG0 + 4 * (2 * (k % NSETK) + ib) + 3)
    thr = lambda u: "v%d" % (THR0 + u)
    mt = lambda u: "v%d" % (M0T + u)
    ctq = lambda p: "v[%d:%d]" % (CTQ0 + 4 * p, CTQ0 + 4 * p + 3)
    ctr = lambda p, r: "v%d" % (CTQ0 + 4 * p + r)
    metap = lambda p: "v%d" % (META0 + 2 * p)
    metan = lambda p: "v%d" % (META0 + 2 * p + 1)
    metapair = lambda p: "v[%d:%d]" % (META0 + 2 * p, META0 + 2 * p + 1)
    vflag, vpub, vrd, vsb, vfb, voff0, vzero, vrdb = ("v%d" % x for x in (VFLAG, VPUB, VRD, VSB, VFB, VOFF0, VZERO, VRDB))
    vgoff = lambda j: "v%d" % (VGOFF0 + j)
    atmp = lambda i: "v%d" % (ATMP0 + (i & 1))

    # hard SGPRs: s80 / s81 / s82 scratch of the DMA, s83 LDS slot of the half-tile being issued, s[84:85] its source, s86 / s87 the
    # pointers' next steps, s[88:89] its meta entry's source | s90 / s91 the steps behind an odd half-tile (rows, meta) | s[92:93] the
    # wave's own flags | s95 the LDS slot of half-tile h + 1 | s97 scratch | s98 the flag word read back
    def slot_addr(dst, idx_sgpr):
        return ["s_and_b32 %s, %s, %d" % (dst, idx_sgpr, NSLOT - 1), "s_mul_i32 %s, %s, %d" % (dst, dst, SS), "s_add_u32 %s, %s, %%[ring]" % (dst, dst)]

    def frag_read(k, ib, i, base=None):
        """fragment (k, ib) of the half-tile whose slot address (+ the lane's swizzled offset of row l & 15, chunk l >> 4) is in `base`
        (vrd: the half-tile in progress; vrdb: the next one, from slot 1 on):
        chunk 4 k + (l >> 4) sits at offset_0 ^ (k << 6) -- the XOR of the swizzle touches bits 4 .. 7 only, and slots start at multiples
        of 256; the second 16 rows are 16 x 2 D bytes behind the first (same swizzle: it depends on row & 15 / (row >> 1) & 7 only)"""
        off = (" offset:%d" % (16 * 2 * D)) if ib else ""
        base = base or vrdb
        if k == 0:
            return ["ds_read_b128 %s, %s%s" % (frag(0, ib), base, off)]
        return ["v_xor_b32 %s, %d, %s" % (atmp(i), 64 * k, base), "ds_read_b128 %s, %s%s" % (frag(k, ib), atmp(i), off)]

    def pointers_from_scratch():
        """s[84:85], s[88:89] := the sources of local half-tile %[issued] (clamped to the split's last: the loop runs two half-tiles past
        the end, and every half-tile issues the same number of operations)"""
        return ["s_sub_u32 s81, %[hend], 1", "s_min_u32 s81, %[issued], s81", "s_lshr_b32 s82, s81, 1", "s_mul_i32 s82, s82, %[nsplit]", "s_add_u32 s82, s82, %[t0]",
                "s_lshl_b32 s82, s82, 1", "s_and_b32 s81, s81, 1", "s_add_u32 s82, s82, s81",          # the global half-tile index
                "s_mul_hi_u32 s85, s82, %d" % HB, "s_mul_i32 s84, s82, %d" % HB, "s_add_u32 s84, s84, %[imglo]", "s_addc_u32 s85, s85, %[imghi]",
                "s_lshl_b32 s82, s82, 4", "s_add_u32 s88, %[metalo], s82", "s_addc_u32 s89, %[metahi], 0"]

    def dma_ops(x_sgpr):
        """the wave's PW pieces of the half-tile at s[84:85] into LDS slot x & 7, and its meta entry behind the slot's rows (every wave
        loads it: the same bytes to the same place -- all waves issue the same number of operations)"""
        Gs = [slot_addr("s83", x_sgpr)]
        if "nodma" in VARIANT:
            return Gs
        for j in range(PW):
            Gs.append([("s_add_u32 m0, s83, %[w1024]" if j == 0 else "s_add_u32 m0, m0, 4096"), "s_nop 0", "global_load_lds_dwordx4 %s, s[84:85]" % vgoff(j)])
        Gs.append(["s_add_u32 m0, s83, %d" % HB, "s_mov_b64 exec, 1", "global_load_lds_dwordx4 %s, s[88:89]" % vzero, "s_mov_b64 exec, -1"])
        return Gs

    flat = lambda Gs: [l for g in Gs for l in g]

    # ---- the loop body: parity 0 half-tile, parity 1 half-tile --------------------------------------------------------------
    EV = [dict(), dict()]      # events per parity and slot: (kind, tag, lines)
    LOAD = [[0] * n_half, [0] * n_half]

    def ev(p, s, kind, tag, lines, cat=None):
        if not lines or (cat is not None and ("no" + cat) in VARIANT):
            return
        p, s = (p + s // n_half) % 2, s % n_half
        EV[p].setdefault(s, []).append((kind, tag, lines))
        LOAD[p][s] += len(lines)

    def spread(p, lo, hi, groups):
        """the groups, in order, into the least loaded slots of [lo, hi] (a later group never before an earlier one)"""
        cur = lo
        for g in groups:
            best = min(range(cur, hi + 1), key=lambda s: (LOAD[p][s], s))
            ev(p, best, "valu", None, g)
            cur = best
        return cur

    slot_of = lambda g, k, ib, j: ((g * NK + k) * IB + ib) * GU + j
    # the maxima of user block (g, j): chain ib is final behind slot_of(g, NK - 1, ib, j) (+ 2: the XDL write has landed) and restarts at
    # n_half + slot_of(g, 0, ib, j); W0 slots between "both final" and the restart of chain 0
    W0 = n_half - slot_of(0, NK - 1, 1, 0) - 2
    PUB = n_half // 2 - 6 if n_half >= 64 else n_half // 2 + 2     # the wave's flags of h - 1 are published here (in half-tile h): every compare of h - 1 lies in front of it
    TESTS = []          # the wave's flags of h - 1 are complete here (in half-tile h)
    AHEAD = []                           # fragments the body reads in the half-tile BEFORE theirs
    for p in range(2):
        q = 1 - p
        # -- fixed places first
        # slot 1: the flag word of h - 2's parity (complete behind the barrier that ended h - 1); reads now go to the slot of h + 1
        ev(p, 1, "lds", ("flag", p), ["ds_read_b32 %s, %s offset:%d" % (vflag, vfb, 4 * p)], "flag")
        ev(p, 0, "valu", None, ["v_mov_b32 %s, %s" % (vrd, vrdb)], "salu")                   # (the next half-tile has become this one)
        ev(p, 1, "valu", None, ["v_add_u32 %s, s95, %s" % (vrdb, voff0), "v_mov_b32 %s, s95" % vsb], "salu")
        # slot 2: the meta pair (pmax, nmax) of the NEXT half-tile: its ct is the C operand of that half-tile's first MFMAs
        ev(p, 2, "lds", ("meta", q), ["ds_read_b64 %s, %s offset:%d" % (metapair(q), vsb, HB)], "flag")
        # fragment (k, ib) is read right behind the last MFMA that reads the register set it goes into: the set's previous occupant is
        # fragment (k - NSETK, ib) of this half-tile, or (k < NSETK) fragment (NK - NSETK + k, ib) of the previous one -- read in the
        # half-tile BEFORE its own then, through vrdb (the very last set frees up at the end of the half-tile: read in slot 0 of its own)
        n_rd = 0
        for k in range(NK):
            for ib in range(IB):
                if k >= NSETK:
                    s = slot_of(G - 1, k - NSETK, ib, GU - 1) + 1
                    assert 1 <= s < slot_of(0, k, ib, 0) - 4
                    ev(p, s, "lds", ("frag", p, k, ib), frag_read(k, ib, n_rd, vrd), "frag")
                else:
                    s = slot_of(G - 1, NK - NSETK + k, ib, GU - 1) + 1
                    if s == n_half:
                        ev(p, 0, "lds", ("frag", p, k, ib), frag_read(k, ib, n_rd, vrd), "frag")
                    else:
                        assert s >= 2
                        ev(p, s, "lds", ("frag", q, k, ib), frag_read(k, ib, n_rd, vrdb), "frag")
                        if p == 0:
                            AHEAD.append((k, ib))
                n_rd += 1
        TESTS.append([])
        for g in range(G):
            for j in range(GU):
                u = g * GU + j
                c0, c1 = slot_of(g, NK - 1, 0, j), slot_of(g, NK - 1, 1, j)
                r0, r1 = n_half + slot_of(g, 0, 0, j), n_half + slot_of(g, 0, 1, j)
                a, b = (lambda r: accr(u, 0, r)), (lambda r: accr(u, 1, r))
                # flag <=> max(m + ct, pmax) > thr: the product could reach the threshold, or (clamp) a popularity of the half-tile beats
                # it -- a head below 1 x pop may qualify whatever the product says.  ct comes in through the C operand of the chain's first
                # MFMA (the accumulators hold s~' + ct); the clamp is wave-uniform (pmax > the wave's lowest threshold: one compare per
                # half-tile, below); what is left per user block is the maximum of its 8 accumulator registers and one compare.
                # (earliest slot, last slot, lines): placed below, every operation into the least loaded slot of its window
                cmp_lo, cmp_hi = max(PUB, c1 + 6), min(PUB + n_half - 1, c0 + n_half + 1)      # (the next half-tile's first maximum overwrites m)
                # (RD slots behind the chain's last MFMA: a VALU read of an accumulator holds its wave until the matrix pipe has worked its
                # way up to that MFMA -- the further behind, the further the wave may run ahead of the pipe)
                rd = RD if n_half - slot_of(0, NK - 1, 0, 0) - RD >= 12 else 2
                mx = [(c0 + rd, r0 - 1, ["v_max3_f32 %s, %s, %s, %s" % (mt(u), a(0), a(1), a(2))]),
                      (c0 + rd, r0 - 1, ["v_max_f32 %s, %s, %s" % (mt(u), mt(u), a(3))]),
                      (c1 + rd, r1 - 1, ["v_max3_f32 %s, %s, %s, %s" % (mt(u), mt(u), b(0), b(1))]),
                      (c1 + rd, r1 - 1, ["v_max3_f32 %s, %s, %s, %s" % (mt(u), mt(u), b(2), b(3))])]
                mx = [(lo, min(hi, cmp_hi - (4 - i)), ln) for i, (lo, hi, ln) in enumerate(mx)]
                TESTS[p].append(mx + [(cmp_lo, cmp_hi, ["v_cmp_gt_f32 vcc, %s, %s" % (mt(u), thr(u)), "s_or_b64 s[92:93], s[92:93], vcc"])])
        # the clamp of THIS half-tile (its pmax is in the parity's meta pair until slot 2 of the next half-tile reads h + 2's)
        ev(p, PUB + 2, "valu", None, ["v_cmp_lt_f32 vcc, %%[tmin], %s" % metap(p), "s_or_b64 s[92:93], s[92:93], vcc"], "test")
    # ---- the tests (tight windows: before what may move): every operation into the least loaded slot of its window (in order within a user block)
    if "notest" not in VARIANT:
        for p in range(2):
            for seq in sorted(TESTS[p], key=lambda q_: q_[0][0]):
                prev = -1
                for i, (lo, hi, lines) in enumerate(seq):
                    lo = max(lo, prev + 1)
                    hi = min(seq[t][1] - (t - i) for t in range(i, len(seq)))        # (leave a slot for each operation behind this one)
                    assert lo <= hi, (D, p, i, lo, hi)
                    load = lambda sl: LOAD[(p + sl // n_half) % 2][sl % n_half]
                    best = min(range(lo, hi + 1), key=lambda sl: (load(sl), sl))
                    if i == len(seq) - 1:
                        assert PUB <= best < PUB + n_half            # (every compare of half-tile h between the publish of h - 1 and of h)
                    ev(p, best, "valu", None, lines)
                    prev = best

    for p in range(2):
        q = 1 - p
        # -- then what may move
        # the flag word: it holds h + 1 <=> half-tile h - 2 raised a flag in some wave -> everybody leaves here (h - 2 and h - 1 are scored
        # again outside); parity 1 only: the sweep is over once the flags of its last half-tile (hend - 1, looked at in hend + 1) are in
        chk = ["v_readfirstlane_b32 s98, %s" % vflag, "s_add_u32 s97, %[h], 1", "s_cmp_eq_u32 s98, s97"] + ([] if "noexit" in VARIANT else ["s_cbranch_scc1 91f"])
        if "noflag" in VARIANT:
            chk = []
        if p == 1:
            chk += ["s_cmp_gt_u32 %[h], %[hend]", "s_cbranch_scc1 92f"]
        s0 = 10
        ev(p, s0, "check", ("flag", p), chk)
        # ct of the NEXT half-tile from its meta pair (pmax, nmax) -- the slack between the bf16 product and a bound of the exact head --
        # into all four registers of that parity's quad (this half-tile's last reader of it: the first MFMAs of its second group)
        sq = slot_of(G - 1, 0, 1, GU - 1) + 2
        ev(p, sq, "check", ("meta", q), ["v_fma_f32 %s, %%[eu], %s, %s" % (ctr(q, 0), metan(q), metap(q))], "flag")
        ev(p, sq + 1, "valu", None, ["v_mov_b32 %s, %s" % (ctr(q, r), ctr(q, 0)) for r in (1, 2, 3)], "flag")
        # the LDS slot of h + 2 (s95 was read in slot 1)
        s1 = spread(p, s0 + 1, s0 + 4, [] if "nosalu" in VARIANT else [["s_add_u32 s97, %[h], 2"] + slot_addr("s95", "s97")])
        # the pieces (and the meta entry) of h + PFD -- their slot held h + PFD - 8 -- and the pointers' step to the next half-tile: none
        # past the end; behind an even half-tile the tile's other half, behind an odd one the split's next tile
        x_odd = (p + PFD) & 1
        step = ["s_add_u32 s80, %%[h], %d" % (PFD + 1), "s_cmp_lt_u32 s80, %[hend]", "s_cselect_b32 s86, %s, 0" % ("s90" if x_odd else "%d" % HB),
                "s_cselect_b32 s87, %s, 0" % ("s91" if x_odd else "16")]
        adv = [["s_add_u32 s84, s84, s86", "s_addc_u32 s85, s85, 0"], ["s_add_u32 s88, s88, s87", "s_addc_u32 s89, s89, 0", "s_mov_b32 %[issued], s80"]]
        lastdma = n_half - 3
        assert lastdma > s1 + 4 and s0 not in (PUB, PUB + 1)
        spread(p, s1, lastdma, [] if "nosalu" in VARIANT else [["s_add_u32 s81, %%[h], %d" % PFD]] + dma_ops("s81") + [step[:2], step[2:]] + adv)
        # my own flags of h - 1 are complete: publish them (h + 2 into the word of h - 1's parity when any is set), start afresh
        # (s94: 0 until the first publish behind an entry -- the compares in front of it looked at what the code outside left in the
        # accumulators; the half-tile they would speak for has been scored again outside)
        ev(p, PUB, "valu", None, ["s_add_u32 s97, %[h], 2", "s_cmp_lg_u64 s[92:93], 0", "s_cselect_b32 s98, s97, 0", "s_and_b32 s98, s98, s94", "s_mov_b32 s94, -1",
                                  "s_mov_b64 s[92:93], 0"], "flag")
        ev(p, PUB + 1, "valu", None, ["v_mov_b32 %s, s98" % vpub, "s_mov_b64 exec, 1"], "flag")
        ev(p, PUB + 1, "lds", ("pub", p), ["ds_max_u32 %s, %s offset:%d" % (vfb, vpub, 4 * q)], "flag")
        ev(p, PUB + 1, "valu", None, ["s_mov_b64 exec, -1"], "flag")

    def build_body(state_in):
        lg = list(state_in)
        out = []

        def wait_for(tag):
            if tag in lg:
                pos = len(lg) - 1 - lg[::-1].index(tag)
                out.append("s_waitcnt lgkmcnt(%d)" % min(15, len(lg) - 1 - pos))
                del lg[:pos + 1]

        for p in range(2):
            out.append("2%d:" % p)                           # entry label of the parity-p half-tile (local half-tile %[h])
            for g in range(G):
                for k in range(NK):
                    for ib in range(IB):
                        for j in range(GU):
                            s, u = slot_of(g, k, ib, j), g * GU + j
                            if g == 0 and j == 0:
                                wait_for(("frag", p, k, ib))
                            out.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (acc(u, ib), frag(k, ib), usr(u, k), ctq(p) if k == 0 else acc(u, ib)))
                            for kind, tag, lines in EV[p].get(s, []):
                                if kind == "lds":
                                    out.extend(lines)
                                    lg.append(tag)
                                elif kind == "check":               # an LDS read must have returned
                                    wait_for(tag)
                                    out.extend(lines)
                                else:
                                    out.extend(lines)
            # end of the half-tile: my pieces of h + 2 have landed (everything but the operations of the PFD - 2 half-tiles behind it), and so
            # will everybody's behind the barrier.  The next half-tile becomes current.
            out += ["s_waitcnt vmcnt(%d)" % ((PFD - 2) * OPS)] + ([] if "nobarrier" in VARIANT else ["s_barrier"]) + ["s_add_u32 %[h], %[h], 1"]
        out.append("s_branch 20b")
        return out, lg

    # steady state of the counted LDS waits
    _, st1 = build_body([])
    b2, st2 = build_body(st1)
    b3, st3 = build_body(st2)
    assert st2 == st3 and b2 == b3, D
    assert (PFD - 2) * OPS <= 63 and 2 <= PFD <= NSLOT - 2

    # ---- prologue (every entry) ------------------------------------------------------------------------------------------------
    P = []
    P += ["s_mov_b32 %[m0save], m0", "s_waitcnt vmcnt(0) lgkmcnt(0)", "v_mov_b32 %s, 0" % vzero, "v_mov_b32 %s, %%[flags]" % vfb]
    # (the thresholds come in in v[THR0 : THR0 + UB): physical-register inputs)
    for j in range(PW):
        P.append("v_add_u32 %s, %d, %%[lane16]" % (vgoff(j), 4096 * j))
        P.append("v_add_u32 %s, %%[w1024], %s" % (vgoff(j), vgoff(j)))
    # the lane's swizzled offset of fragment (0, 0): row r = lane & 15, chunk c = lane >> 4 -> r * 2D + ((c ^ swz(r)) << 4)
    t0r, t1r = atmp(0), atmp(1)
    P += ["v_lshrrev_b32 %s, 4, %%[lane16]" % t0r, "v_and_b32 %s, 15, %s" % (t1r, t0r), "v_lshrrev_b32 %s, 4, %s" % (t0r, t0r)]
    if D >= 128:
        P.append("v_and_b32 %s, 15, %s" % (voff0, t1r))                            # swz(r) = r & 15
    else:
        P += ["v_lshrrev_b32 %s, 1, %s" % (voff0, t1r), "v_and_b32 %s, 7, %s" % (voff0, voff0)]         # (r >> 1) & 7
    P += ["v_xor_b32 %s, %s, %s" % (voff0, voff0, t0r), "v_lshlrev_b32 %s, 4, %s" % (voff0, voff0),
          "v_lshl_add_u32 %s, %s, %d, %s" % (voff0, t1r, (2 * D).bit_length() - 1, voff0)]
    # the pointers' steps behind an odd half-tile: to the first half of the split's next tile
    P += ["s_mul_i32 s90, %%[nsplit], %d" % (2 * HB), "s_sub_u32 s90, s90, %d" % HB, "s_lshl_b32 s91, %[nsplit], 5", "s_sub_u32 s91, s91, 16"]
    # the wave's user fragments -> AGPRs, at EVERY entry: the compiler uses AGPRs as spill space between the statements
    P.append("s_mov_b64 s[88:89], %[ufrag]")
    for i in range(UB * NK):
        if i % 4 == 0 and i > 0:
            P += ["s_add_u32 s88, s88, 4096", "s_addc_u32 s89, s89, 0"]
        P.append("global_load_dwordx4 a[%d:%d], %%[lane16], s[88:89] offset:%d" % (4 * i, 4 * i + 3, 1024 * (i % 4)))
    P.append("s_waitcnt vmcnt(0)")
    # the flag words are dealt with outside: clear them (every wave; the barrier below orders it)
    P += ["ds_write_b32 %s, %s" % (vfb, vzero), "ds_write_b32 %s, %s offset:4" % (vfb, vzero)]
    # catch up with the loads: the pieces of half-tiles issued .. h + PFD - 1 (first entry: all of them; their slots are free)
    P += ["5:", "s_add_u32 s97, %%[h], %d" % PFD, "s_cmp_ge_u32 %[issued], s97", "s_cbranch_scc1 6f"]
    P += pointers_from_scratch() + flat(dma_ops("%[issued]")) + ["s_add_u32 %[issued], %[issued], 1", "s_branch 5b", "6:"]
    P += pointers_from_scratch()                             # the running pointers of the body: half-tile h + PFD
    # everything issued has landed; behind the barrier everybody's has
    P += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
    P += slot_addr("s97", "%[h]") + ["v_add_u32 %s, s97, %s" % (vrdb, voff0), "v_mov_b32 %s, s97" % vsb, "s_add_u32 s97, %[h], 1"] + slot_addr("s95", "s97")
    P += ["s_mov_b64 s[92:93], 0", "s_mov_b32 s94, 0"]
    # the entry half-tile's ct quad (both parities' registers: the branch below picks the parity), from its meta pair
    P += ["ds_read_b64 %s, %s offset:%d" % (metapair(0), vsb, HB), "s_waitcnt lgkmcnt(0)", "v_fma_f32 %s, %%[eu], %s, %s" % (ctr(0, 0), metan(0), metap(0))]
    P += ["v_mov_b32 %s, %s" % (ctr(0, r), ctr(0, 0)) for r in (1, 2, 3)] + ["v_mov_b32 %s, %s" % (ctr(1, r), ctr(0, 0)) for r in range(4)]
    P += ["v_mov_b32 %s, %s" % (metap(1), metap(0))]
    # every fragment the body reads AHEAD of the half-tile it belongs to
    for i, (k, ib) in enumerate(AHEAD):
        P += frag_read(k, ib, i, vrdb)
    P += ["s_bitcmp1_b32 %[h], 0", "s_cbranch_scc1 9f"]
    for par in range(2):
        # (the first slots of the entry half-tile carry the tail of the PREVIOUS half-tile's tests: -inf makes them fail)
        P += ["s_waitcnt lgkmcnt(0)", "s_branch 2%df" % par]
        if par == 0:
            P.append("9:")
    # ---- exits ------------------------------------------------------------------------------------------------------------------
    drain = ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15", "s_mov_b32 m0, %[m0save]"]
    E = []
    E += ["91:", "s_mov_b32 %[reason], 1"] + drain + ["s_branch 99f"]          # half-tile h - 2 raised a flag in some wave (h - 1 has not been looked at)
    E += ["92:", "s_mov_b32 %[reason], 0"] + drain + ["99:"]                   # the sweep is over
    if os.environ.get("V5_LOADS"):
        for p in range(2):
            print("D=%d UB=%d parity %d fillers per slot: %s" % (D, UB, p, " ".join("%d" % x for x in LOAD[p])), file=sys.stderr)
    return P + b2 + E, THR0, LO_CLOBBER


def emit(D, UB):
    L, THR0, LO = gen(D, UB)
    out = []
    out.append("template <>")
    out.append("struct Loop6<%d, %d> {" % (D, UB))
    out.append("    static constexpr int kSlotBytes = %d, kPfd = %d;" % (64 * D + (512 if D == 256 else 256), PFD))
    out.append("    // h: the local half-tile to run next (in: where to (re)start; out: the half-tile in progress when the statement left).")
    out.append("    // issued: half-tiles whose pieces this wave has issued.  reason: 0 = the sweep is over, 1 = half-tile h - 2 raised a flag in some")
    out.append("    // wave of the workgroup (all four leave together; h - 1 has not been looked at).")
    out.append("    static __device__ __forceinline__ void run(unsigned& h, unsigned& issued, unsigned& reason, unsigned hend, unsigned ring, unsigned flags, unsigned w1024,")
    out.append("                                               unsigned t0, unsigned nsplit, unsigned imglo, unsigned imghi, unsigned metalo, unsigned metahi, float eu, float tmin, const void* ufrag,")
    out.append("                                               const float (&thr)[%d], unsigned lane16) {" % UB)
    out.append("#if defined(__HIP_DEVICE_COMPILE__)")
    out.append("        unsigned m0save;")
    out.append("        asm volatile(")
    for l in L:
        out.append('            "%s\\n\\t"' % l)
    out.append('            : [h] "+&s"(h), [issued] "+&s"(issued), [reason] "=&s"(reason), [m0save] "=&s"(m0save)')
    ins = ['[hend] "s"(hend)', '[ring] "s"(ring)', '[flags] "s"(flags)', '[w1024] "s"(w1024)', '[t0] "s"(t0)', '[nsplit] "s"(nsplit)',
           '[imglo] "s"(imglo)', '[imghi] "s"(imghi)', '[metalo] "s"(metalo)', '[metahi] "s"(metahi)', '[eu] "s"(eu)', '[tmin] "s"(tmin)', '[ufrag] "s"(ufrag)', '[lane16] "v"(lane16)']
    ins += ['"{v%d}"(thr[%d])' % (THR0 + u, u) for u in range(UB)]
    out.append("            : " + ", ".join(ins))
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"s%d"' % r for r in range(80, 100)] + \
           ['"v%d"' % r for r in range(LO, 16 * UB) if not THR0 <= r < THR0 + UB] + ['"a%d"' % r for r in range(4 * UB * (D // 32))] + \
           (['"v%d"' % r for r in range(128, 132)] if D == 256 else [])
    out.append("            : " + ", ".join(clob) + ");")
    out.append("#endif")
    out.append("    }")
    out.append("};")
    return "\n".join(out)


def main():
    print("// GENERATED by tools/gen_v6_loop_asm.py -- do not edit.")
    print("#pragma once")
    print("template <int D, int UB> struct Loop6;")
    for UB in (16, 8):
        for D in (64, 128):
            print(emit(D, UB))
    print(emit(256, 8))


if __name__ == "__main__":
    main()
