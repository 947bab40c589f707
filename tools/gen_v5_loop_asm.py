#!/usr/bin/env python
"""Generates pda_amd/csrc/pda_v5_loop_asm.h: the main loop of the "huge" geometry of the sweep (sweep5_kernel, pda_v5_sweep.h) as ONE
inline-asm statement per (re-)entry -- d = 64 / 128, popularity head, dense sweep in visiting order.

The mapping (measured first in tools/ubench/mfma_struct5.hip, V4): four waves per workgroup, one per SIMD, 512 registers
each.  A wave owns 256 users: their bf16 rows sit in AGPRs as eight B operands of 32 users (a[0 .. 4 UA NK)), the eight accumulators
in v[128:255].  Every item fragment read from the LDS (one ds_read_b128) feeds EIGHT MFMAs, and the product is transposed -- A = item
fragment, B = user fragment -- so that a lane of an accumulator holds 16 items of ONE user: the threshold test is a per-lane compare
(7 v_max3 + v_max, then v_add + v_max + v_cmp per 32 x 32 block, in the MFMA shadow), and the folded test k-step of generation 4 (1/9
of all MFMAs at d = 128, 1/5 at d = 64) is gone.

Order of the MFMAs of a 32-item half-tile: two GROUPS of four accumulator chains, each group running all NK k-steps before the other
starts (for g: for k: for j).  An accumulator is then idle for n_half - 4 (NK - 1) slots between its last MFMA and its restart (36 of
64 slots at d = 128; the k-major order of the first version left 8): its eight maxima are spread over that window instead of eight per
slot in a burst (PMC: the bursts cost 10 % of the matrix pipe's time).  All NK fragments of a half-tile stay in registers (NK sets);
fragment k of the NEXT half-tile is read right behind its last use, 4 (NK - 1) slots before its first.

The four waves run in step, ONE s_barrier per half-tile:
  * half-tile h: every wave issues its LDS-DMA pieces of half-tile h + PFD (and the half-tile's 16-byte meta entry, behind the slot's
    rows), runs the 8 NK MFMAs of h, finishes the tests of h - 1 in the shadow, publishes "h - 1 raised a flag in one of my lanes" (a
    ds_max of h + 2 into the flag word of h - 1's parity), waits for its own pieces of h + 2 (a counted vmcnt: all of the loop's
    vector-memory operations are LDS-DMA, in order among themselves) and meets the others at the barrier: behind it h + 2 has landed for
    everybody;
  * the flag word of h - 2's parity is read behind the barrier: when it holds h + 1, ALL four waves leave the statement at the same
    place; the C++ around it scores h - 2 and h - 1 again with compiler-visible MFMAs (h - 1's flags were still in flight), rescores
    the candidates exactly and re-enters at h.  (Values instead of bits: a word never has to be cleared inside the loop, so two words
    do -- a clear would race with the faster waves' next publish.)

    python tools/gen_v5_loop_asm.py > pda_amd/csrc/pda_v5_loop_asm.h

V5_VARIANT=<list> builds timing-only variants (tools/ab_huge.sh; results WRONG by construction): notest, nodma, nobarrier, noexit, noflag.
V5_LOADS=1 prints the fillers per MFMA slot.
"""
import os
import sys

VARIANT = set(filter(None, os.environ.get("V5_VARIANT", "").split(",")))     # timing-only A/B knobs; the product build has none

UA, GU = 8, 4                        # accumulator chains per wave; chains per group
ACC0, FRAG0 = 128, 96                # acc: v[128:255]; fragment sets: v[96 : 96 + 4 NK)
THR0, M0T = 88, 80                   # thr[8], m[8]
LO_CLOBBER = 60
VFLAG, VPUB, VRD, VSB, VFB, VOFF0, VGOFF0, VZERO = 60, 61, 62, 63, 64, 65, 66, 68     # (VGOFF0: two registers)
CT0, META0, ATMP0 = 70, 72, 76       # ct[2], meta pairs (pmax, nmax)[2], address temporaries[2]
NSLOT = 8
PFD = int(os.environ.get("V5_PFD", "4"))     # half-tile h issues the pieces of h + PFD


def acc(u):
    return "v[%d:%d]" % (ACC0 + 16 * u, ACC0 + 16 * u + 15)


def accr(u, r):
    return "v%d" % (ACC0 + 16 * u + r)


def gen(D):
    NK = D // 16
    HB = 32 * 2 * D                      # one half-tile: 32 rows of 2 D bytes, 16-byte chunks XOR-swizzled (no padding)
    SS = HB + 256                        # LDS slot: the rows, then the half-tile's meta entry (pmax, nmax, 0, 0); a multiple of 256
    PW = HB // 1024 // 4                 # LDS-DMA pieces per wave and half-tile (2 at d = 128, 1 at d = 64)
    OPS = 0 if "nodma" in VARIANT else PW + 1   # vector-memory operations per half-tile, ALL of them LDS-DMA (in order among themselves)
    G = UA // GU
    n_half = NK * UA                     # MFMA slots per half-tile
    W = n_half - (NK - 1) * GU - 2       # slots between an accumulator's last MFMA (+ 2: the XDL write has landed) and its restart
    usr = lambda u, k: "a[%d:%d]" % (4 * (u * NK + k), 4 * (u * NK + k) + 3)
    frag = lambda k: "v[%d:%d]" % (FRAG0 + 4 * k, FRAG0 + 4 * k + 3)
    thr = lambda u: "v%d" % (THR0 + u)
    mt = lambda u: "v%d" % (M0T + u)
    ct = lambda p: "v%d" % (CT0 + p)
    metap = lambda p: "v%d" % (META0 + 2 * p)
    metan = lambda p: "v%d" % (META0 + 2 * p + 1)
    metapair = lambda p: "v[%d:%d]" % (META0 + 2 * p, META0 + 2 * p + 1)
    vflag, vpub, vrd, vsb, vfb, voff0, vzero = ("v%d" % x for x in (VFLAG, VPUB, VRD, VSB, VFB, VOFF0, VZERO))
    vgoff = lambda j: "v%d" % (VGOFF0 + j)
    atmp = lambda i: "v%d" % (ATMP0 + (i & 1))

    # hard SGPRs: s80 / s81 / s82 scratch of the DMA, s83 LDS slot of the half-tile being issued, s[84:85] its source, s86 / s87 the
    # pointers' next steps, s[88:89] its meta entry's source | s90 / s91 the steps behind an odd half-tile (rows, meta) | s[92:93] the
    # wave's own flags | s95 the LDS slot of half-tile h + 1 | s97 scratch | s98 the flag word read back
    def slot_addr(dst, idx_sgpr):
        return ["s_and_b32 %s, %s, %d" % (dst, idx_sgpr, NSLOT - 1), "s_mul_i32 %s, %s, %d" % (dst, dst, SS), "s_add_u32 %s, %s, %%[ring]" % (dst, dst)]

    def frag_read(k, i):
        """fragment k of the half-tile whose slot address (+ the lane's swizzled offset of chunk hh) is in vrd: chunk 2 k + hh sits at
        offset_0 ^ (k << 5) -- the XOR of the swizzle touches bits 4 .. 7 only, and slots start at multiples of 256"""
        if k == 0:
            return ["ds_read_b128 %s, %s" % (frag(0), vrd)]
        return ["v_xor_b32 %s, %d, %s" % (atmp(i), 32 * k, vrd), "ds_read_b128 %s, %s" % (frag(k), atmp(i))]

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

    slot_of = lambda g, k, j: (g * NK + k) * GU + j
    TAIL0 = 4 + W                        # tails of a group: slots base + TAIL0 .. + 7
    PUB = slot_of(G - 1, NK - 1, 0) + TAIL0 + 2 * GU - n_half          # the wave's flags of h - 1 are complete here (in half-tile h)
    for p in range(2):
        q = 1 - p
        # -- fixed places first
        # slot 0: the last fragment of THIS half-tile (its register set was in use until the end of h - 1) and its meta pair
        ev(p, 0, "lds", ("frag", p, NK - 1), frag_read(NK - 1, 0), "frag")
        ev(p, 0, "lds", ("meta", p), ["ds_read_b64 %s, %s offset:%d" % (metapair(p), vsb, HB)], "flag")
        # slot 1: the flag word of h - 2's parity (complete behind the barrier that ended h - 1); reads now go to the slot of h + 1
        ev(p, 1, "lds", ("flag", p), ["ds_read_b32 %s, %s offset:%d" % (vflag, vfb, 4 * p)], "flag")
        ev(p, 1, "valu", None, ["v_add_u32 %s, s95, %s" % (vrd, voff0), "v_mov_b32 %s, s95" % vsb], "salu")
        # fragments 0 .. NK - 2 of h + 1: right behind the last MFMA of this half-tile that reads the register set
        for k in range(NK - 1):
            ev(p, slot_of(G - 1, k, GU - 1) + 1, "lds", ("frag", q, k), frag_read(k, k), "frag")
        # tests: chain (g, j) is final behind slot_of(g, NK - 1, j) and restarts n_half - (NK - 1) GU slots later: its eight maxima (which
        # read the accumulator) spread over that window; the add, maximum, compare and OR (which do not) behind it, two slots per chain
        for g in range(G):
            for j in range(GU):
                u = g * GU + j
                c = slot_of(g, NK - 1, j)
                ops = [["v_max3_f32 %s, %s, %s, %s" % (mt(u), accr(u, 0), accr(u, 1), accr(u, 2))]]
                for r in range(3, 15, 2):
                    ops.append(["v_max3_f32 %s, %s, %s, %s" % (mt(u), mt(u), accr(u, r), accr(u, r + 1))])
                ops.append(["v_max_f32 %s, %s, %s" % (mt(u), mt(u), accr(u, 15))])
                for i, o in enumerate(ops):
                    sl = c + 2 + (i * W) // len(ops)
                    assert sl < n_half + slot_of(g, 0, j), (D, g, j, i)
                    ev(p, sl, "valu", None, o, "test")
                # flag <=> max(m + ct, pmax) > thr: the product could reach the threshold, or (clamp) a popularity of the half-tile beats
                # it -- a head below 1 x pop may qualify whatever the product says
                base = slot_of(g, NK - 1, 0)
                assert base + TAIL0 + 2 * j > c + 2 + (7 * W) // 8
                ev(p, base + TAIL0 + 2 * j, "valu", None, ["v_add_f32 %s, %s, %s" % (mt(u), mt(u), ct(p)), "v_max_f32 %s, %s, %s" % (mt(u), mt(u), metap(p))], "test")
                ev(p, base + TAIL0 + 2 * j + 1, "valu", None, ["v_cmp_gt_f32 vcc, %s, %s" % (mt(u), thr(u)), "s_or_b64 s[92:93], s[92:93], vcc"], "test")
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
        assert s0 > 2 * GU + 1           # (the tails of the previous half-tile's first group are behind us: an exit loses nothing of them)
        ev(p, s0, "check", ("flag", p), chk)
        # ct of THIS half-tile from its meta pair (pmax, nmax): the slack between the bf16 product and a bound of the exact head
        ev(p, s0 + 1, "check", ("meta", p), ["v_fma_f32 %s, %%[eu], %s, %s" % (ct(p), metan(p), metap(p))], "flag")
        # the LDS slot of h + 2 (s95 was read in slot 1)
        s1 = spread(p, s0 + 1, s0 + 4, [] if "nosalu" in VARIANT else [["s_add_u32 s97, %[h], 2"] + slot_addr("s95", "s97")])
        # the pieces (and the meta entry) of h + PFD -- their slot held h + PFD - 8 -- and the pointers' step to the next half-tile: none
        # past the end; behind an even half-tile the tile's other half, behind an odd one the split's next tile
        x_odd = (p + PFD) & 1
        step = ["s_add_u32 s80, %%[h], %d" % (PFD + 1), "s_cmp_lt_u32 s80, %[hend]", "s_cselect_b32 s86, %s, 0" % ("s90" if x_odd else "%d" % HB),
                "s_cselect_b32 s87, %s, 0" % ("s91" if x_odd else "16")]
        adv = [["s_add_u32 s84, s84, s86", "s_addc_u32 s85, s85, 0"], ["s_add_u32 s88, s88, s87", "s_addc_u32 s89, s89, 0", "s_mov_b32 %[issued], s80"]]
        lastdma = min(PUB - 2, slot_of(G - 1, 0, GU - 1)) if D >= 128 else n_half - 2
        spread(p, s1, lastdma, [] if "nosalu" in VARIANT else [["s_add_u32 s81, %%[h], %d" % PFD]] + dma_ops("s81") + [step[:2], step[2:]] + adv)
        # my own flags of h - 1 are complete: publish them (h + 2 into the word of h - 1's parity when any is set), start afresh
        ev(p, PUB, "valu", None, ["s_add_u32 s97, %[h], 2", "s_cmp_lg_u64 s[92:93], 0", "s_cselect_b32 s98, s97, 0", "s_mov_b64 s[92:93], 0"], "flag")
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
                    for j in range(GU):
                        s, u = slot_of(g, k, j), g * GU + j
                        if g == 0:
                            wait_for(("frag", p, k))
                        out.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc(u), frag(k), usr(u, k), "0" if k == 0 else acc(u)))
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
    for u in range(UA):
        P.append("v_mov_b32 %s, %%[thr%d]" % (thr(u), u))
    for j in range(PW):
        P.append("v_add_u32 %s, %d, %%[lane16]" % (vgoff(j), 4096 * j))
        P.append("v_add_u32 %s, %%[w1024], %s" % (vgoff(j), vgoff(j)))
    # the lane's swizzled offset of fragment 0: row r = lane & 31, half hh = lane >> 5 -> r * 2D + ((hh ^ swz(r)) << 4)
    t0r, t1r = atmp(0), atmp(1)
    P += ["v_lshrrev_b32 %s, 4, %%[lane16]" % t0r, "v_and_b32 %s, 31, %s" % (t1r, t0r), "v_lshrrev_b32 %s, 5, %s" % (t0r, t0r)]
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
    for i in range(UA * NK):
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
    P += slot_addr("s97", "%[h]") + ["v_add_u32 %s, s97, %s" % (vrd, voff0), "v_mov_b32 %s, s97" % vsb, "s_add_u32 s97, %[h], 1"] + slot_addr("s95", "s97")
    P += ["s_mov_b64 s[92:93], 0"]
    # every fragment the body reads AHEAD of the half-tile it belongs to
    for k in range(NK - 1):
        P += frag_read(k, k)
    P += ["s_bitcmp1_b32 %[h], 0", "s_cbranch_scc1 9f"]
    for par in range(2):
        # (the first slots of the entry half-tile carry the tail of the PREVIOUS half-tile's tests: -inf makes them fail)
        P += ["v_mov_b32 %s, 0xff800000" % ct(1 - par), "v_mov_b32 %s, 0xff800000" % metap(1 - par), "s_waitcnt lgkmcnt(0)", "s_branch 2%df" % par]
        if par == 0:
            P.append("9:")
    # ---- exits ------------------------------------------------------------------------------------------------------------------
    drain = ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15", "s_mov_b32 m0, %[m0save]"]
    E = []
    E += ["91:", "s_mov_b32 %[reason], 1"] + drain + ["s_branch 99f"]          # half-tile h - 2 raised a flag in some wave (h - 1 has not been looked at)
    E += ["92:", "s_mov_b32 %[reason], 0"] + drain + ["99:"]                   # the sweep is over
    if os.environ.get("V5_LOADS"):
        for p in range(2):
            print("D=%d parity %d fillers per slot: %s" % (D, p, " ".join("%d" % x for x in LOAD[p])), file=sys.stderr)
    return P + b2 + E


def emit(D):
    L = gen(D)
    out = []
    out.append("template <>")
    out.append("struct Loop5<%d> {" % D)
    out.append("    static constexpr int kSlotBytes = %d, kPfd = %d;" % (64 * D + 256, PFD))
    out.append("    // h: the local half-tile to run next (in: where to (re)start; out: the half-tile in progress when the statement left).")
    out.append("    // issued: half-tiles whose pieces this wave has issued.  reason: 0 = the sweep is over, 1 = half-tile h - 2 raised a flag in some")
    out.append("    // wave of the workgroup (all four leave together; h - 1 has not been looked at).")
    out.append("    static __device__ __forceinline__ void run(unsigned& h, unsigned& issued, unsigned& reason, unsigned hend, unsigned ring, unsigned flags, unsigned w1024,")
    out.append("                                               unsigned t0, unsigned nsplit, unsigned imglo, unsigned imghi, unsigned metalo, unsigned metahi, float eu, const void* ufrag,")
    out.append("                                               const float (&thr)[8], unsigned lane16) {")
    out.append("#if defined(__HIP_DEVICE_COMPILE__)")
    out.append("        unsigned m0save;")
    out.append("        asm volatile(")
    for l in L:
        out.append('            "%s\\n\\t"' % l)
    out.append('            : [h] "+&s"(h), [issued] "+&s"(issued), [reason] "=&s"(reason), [m0save] "=&s"(m0save)')
    ins = ['[hend] "s"(hend)', '[ring] "s"(ring)', '[flags] "s"(flags)', '[w1024] "s"(w1024)', '[t0] "s"(t0)', '[nsplit] "s"(nsplit)',
           '[imglo] "s"(imglo)', '[imghi] "s"(imghi)', '[metalo] "s"(metalo)', '[metahi] "s"(metahi)', '[eu] "s"(eu)', '[ufrag] "s"(ufrag)', '[lane16] "v"(lane16)']
    ins += ['[thr%d] "v"(thr[%d])' % (u, u) for u in range(8)]
    out.append("            : " + ", ".join(ins))
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"s%d"' % r for r in range(80, 100)] + ['"v%d"' % r for r in range(LO_CLOBBER, 256)] + \
           ['"a%d"' % r for r in range(4 * UA * (D // 16))]
    out.append("            : " + ", ".join(clob) + ");")
    out.append("#endif")
    out.append("    }")
    out.append("};")
    return "\n".join(out)


def main():
    print("// GENERATED by tools/gen_v5_loop_asm.py -- do not edit.")
    print("#pragma once")
    print("template <int D> struct Loop5;")
    for D in (64, 128):
        print(emit(D))


if __name__ == "__main__":
    main()
