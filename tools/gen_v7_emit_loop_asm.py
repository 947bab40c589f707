#!/usr/bin/env python
"""Generates pda_amd/csrc/pda_v7_emit_loop_asm.h: the EMITTING loop of the huge geometry (sweep7_kernel, pda_v7_funnel.h; round 5).

Same machine mapping as tools/gen_v6_loop_asm.py (read its header first): four waves per workgroup, one per SIMD; a wave's bf16 user rows in
AGPRs as B operands of v_mfma_f32_16x16x32_f16 (FP16 here, not bf16: see pda_v7_funnel.h); 32-item half-tiles DMA-ed into eight LDS slots by the MFMA waves themselves, one s_barrier per
half-tile; the product transposed, so that an accumulator lane holds 8 items of ONE user per half-tile (two chains of four registers).

What differs: nothing leaves the loop.  Where the testing loop raises a flag (and all four waves leave, score the half-tile again and rescore its
candidates exactly -- fine for a fraction of a candidate per user, hopeless for the raw head's hundreds), this loop WRITES the lane's eight
accumulator registers out whenever their maximum beats the lane's threshold:

    v_cmp_gt_f32 vcc, m, thr             s_cbranch_vccz over the block; exec := the lanes whose maximum beats the threshold of their user
    buffer_store_dwordx4 acc(u, 0)       } one 48-byte entry per flagged lane: the 8 bounds s~ + ct of its items of this half-tile, then the
    buffer_store_dwordx4 acc(u, 1)       } half-tile's index -- into the lane's OWN list of user block u (a list per (user, quarter of the
    buffer_store_dword   h               } half-tile's items): no cross-lane work, no atomics, no ring)
    v_add_u32 cur, stride, cur           the list's cursor (a VGPR per user block)
    s_mov_b64 exec, -1

(No lane flagged -- two blocks of three -- : a branch over the stores; issuing them with an empty mask cost 1 ms per sweep more.)  The lists are raw buffers: a store past the end of the wave's region
is dropped by the address unit (num_records), the cursor keeps counting and the consumer sees the overflow.  Entry e of lane l, user block u of
the wave sits at  e * 64 * 768 + l * 768 + u * 48  of the wave's region.  The thresholds are fixed for the whole launch: a sweep is a sequence
of launches over growing parts of the catalogue with a selection kernel between them (pda_v7_funnel.h).

The landing of the tiles.  The testing loop waits for its LDS-DMA with a COUNTED s_waitcnt vmcnt(n): loads return in order.  Stores share
the counter and complete out of order with respect to loads (measured: with the counted wait the MFMAs read stale tiles), so here the only
safe wait is vmcnt(0) -- which also waits for the acknowledgement of every store issued so far.  Hence: ONE wait + s_barrier per TWO half-tiles,
in the middle of the odd one (slot A = n_half / 2), where the stores of the previous half-tile's last blocks are ~50 MFMA slots old and the
next burst has not begun; odd half-tiles issue their pieces (of h + 3) behind it, even ones early: every piece has more than a half-tile to
land, and the slot it lands in was read four half-tiles ago.  Nothing of half-tile h + 1 is read before slot A of an odd h (the meta pair
moved behind it).

    python tools/gen_v7_emit_loop_asm.py > pda_amd/csrc/pda_v7_emit_loop_asm.h
"""
import os
import sys

GU, IB = 8, 2                        # user blocks per group; 16-item blocks per half-tile
NSLOT = 8
PFD = int(os.environ.get("V7_PFD", "3"))    # half-tile h issues the pieces of h + PFD (see "the landing of the tiles" below)
RD = int(os.environ.get("V6_RD", "10"))      # slots between a chain's last MFMA and the first VALU read of its accumulator
ENTRY = 48                           # bytes per entry
BAR_EVERY = int(os.environ.get("V7_BAR_EVERY", "0"))   # debugging: vmcnt(0) + barrier at the end of EVERY half-tile as well
BRANCH = int(os.environ.get("V7_BRANCH", "1"))         # 1: the stores of a user block behind a branch (s_cbranch_vccz); 0: issued with an empty exec mask (measured: +1 ms per sweep)
ONLY_U = int(os.environ.get("V7_ONLY_U", "-1"))        # debugging: only this user block emits
SPLIT_EMIT = int(os.environ.get("V7_SPLIT", "0"))   # 1: the emission as two events in consecutive slots (the MFMA between them ignores exec)


def gen(D, UB, maxmode=False):
    """maxmode: the loop of the funnel's FIRST launch -- nothing is written per half-tile; every lane keeps the two largest maxima of its user blocks
    (run1 in the cursor registers, run2 in the threshold registers) and the wave the largest ct it formed; they are stored at the exit."""
    NK = D // 32
    if UB == 16:
        ACC0, FRAG0, THR0, M0T, CUR0, MISC0 = 128, 96, 80, 64, 40, 16
        NSETK = NK
    else:
        ACC0, FRAG0, THR0, M0T, CUR0, MISC0 = 64, 48, 40, 32, 132, 8
        NSETK = min(NK, 2)
    LO_CLOBBER = MISC0
    CTQ0, VH, VRD, VSB, VOFF0, VGOFF0, VZERO, VRDB, META0, ATMP0 = (MISC0 + x for x in (0, 8, 10, 11, 13, 14, 16, 17, 18, 22))
    if D == 256:
        assert UB == 8
        VGOFF0 = 128
    # the meta entry of a half-tile, four registers per parity: (pmax, nmax, rmax, 0)
    META0 = 56 if UB == 16 else 140
    assert MISC0 + 24 <= (CUR0 if UB == 16 else M0T) and M0T + UB <= THR0 and THR0 + UB <= FRAG0 and FRAG0 + 8 * NSETK <= ACC0
    LSTRIDE, ESTRIDE = UB * ENTRY, 64 * UB * ENTRY        # bytes between the lists of two lanes, between two entries of a list

    def acc(u, ib):
        c = ACC0 + 4 * (2 * u + ib)
        return "v[%d:%d]" % (c, c + 3)

    def accr(u, ib, r):
        return "v%d" % (ACC0 + 4 * (2 * u + ib) + r)

    HB = 32 * 2 * D
    SS = HB + (512 if D == 256 else 256)
    PW = HB // 1024 // 4
    G = UB // GU
    n_half = NK * IB * UB
    S_VH = n_half // 2                   # the header register takes the half-tile's index here
    A_SLOT = n_half // 2                 # odd half-tiles: s_waitcnt vmcnt(0), s_barrier (everything issued so far has landed, everywhere)
    usr = lambda u, k: "a[%d:%d]" % (4 * (u * NK + k), 4 * (u * NK + k) + 3)
    frag = lambda k, ib: "v[%d:%d]" % (FRAG0 + 4 * (2 * (k % NSETK) + ib), FRAG0 + 4 * (2 * (k % NSETK) + ib) + 3)
    thr = lambda u: "v%d" % (THR0 + u)
    mt = lambda u: "v%d" % (M0T + u)
    cur = lambda u: "v%d" % (CUR0 + u)
    ctq = lambda p: "v[%d:%d]" % (CTQ0 + 4 * p, CTQ0 + 4 * p + 3)
    ctr = lambda p, r: "v%d" % (CTQ0 + 4 * p + r)
    metap = lambda p: "v%d" % (META0 + 4 * p)
    metan = lambda p: "v%d" % (META0 + 4 * p + 1)
    metar = lambda p: "v%d" % (META0 + 4 * p + 2)
    metapair = lambda p: "v[%d:%d]" % (META0 + 4 * p, META0 + 4 * p + 3)
    vh, vrd, vsb, voff0, vzero, vrdb = ("v%d" % x for x in (VH, VRD, VSB, VOFF0, VZERO, VRDB))
    vctmax = "v%d" % (MISC0 + 18)
    vgoff = lambda j: "v%d" % (VGOFF0 + j)
    atmp = lambda i: "v%d" % (ATMP0 + (i & 1))

    def slot_addr(dst, idx_sgpr):
        return ["s_and_b32 %s, %s, %d" % (dst, idx_sgpr, NSLOT - 1), "s_mul_i32 %s, %s, %d" % (dst, dst, SS), "s_add_u32 %s, %s, %%[ring]" % (dst, dst)]

    def frag_read(k, ib, i, base):
        off = (" offset:%d" % (16 * 2 * D)) if ib else ""
        if k == 0:
            return ["ds_read_b128 %s, %s%s" % (frag(0, ib), base, off)]
        return ["v_xor_b32 %s, %d, %s" % (atmp(i), 64 * k, base), "ds_read_b128 %s, %s%s" % (frag(k, ib), atmp(i), off)]

    def pointers_from_scratch():
        return ["s_sub_u32 s81, %[hend], 1", "s_min_u32 s81, %[issued], s81", "s_lshr_b32 s82, s81, 1", "s_mul_i32 s82, s82, %[nsplit]", "s_add_u32 s82, s82, %[t0]",
                "s_lshl_b32 s82, s82, 1", "s_and_b32 s81, s81, 1", "s_add_u32 s82, s82, s81",
                "s_mul_hi_u32 s85, s82, %d" % HB, "s_mul_i32 s84, s82, %d" % HB, "s_add_u32 s84, s84, %[imglo]", "s_addc_u32 s85, s85, %[imghi]",
                "s_lshl_b32 s82, s82, 4", "s_add_u32 s88, %[metalo], s82", "s_addc_u32 s89, %[metahi], 0"]

    def dma_ops(x_sgpr):
        Gs = [slot_addr("s83", x_sgpr)]
        for j in range(PW):
            Gs.append([("s_add_u32 m0, s83, %[w1024]" if j == 0 else "s_add_u32 m0, m0, 4096"), "s_nop 0", "global_load_lds_dwordx4 %s, s[84:85]" % vgoff(j)])
        Gs.append(["s_add_u32 m0, s83, %d" % HB, "s_mov_b64 exec, 1", "global_load_lds_dwordx4 %s, s[88:89]" % vzero, "s_mov_b64 exec, -1"])
        return Gs

    flat = lambda Gs: [l for g in Gs for l in g]

    EV = [dict(), dict()]
    LOAD = [[0] * n_half, [0] * n_half]

    def ev(p, s, kind, tag, lines):
        if not lines:
            return
        p, s = (p + s // n_half) % 2, s % n_half
        EV[p].setdefault(s, []).append((kind, tag, lines))
        LOAD[p][s] += len(lines)

    def spread(p, lo, hi, groups):
        cur_ = lo
        for g in groups:
            best = min(range(cur_, hi + 1), key=lambda s: (LOAD[p][s], s))
            ev(p, best, "valu", None, g)
            cur_ = best
        return cur_

    def spread_least(p, lo, hi, groups):
        """the groups, in order, into the len(groups) least loaded slots of [lo, hi]"""
        slots = sorted(sorted(range(lo, hi + 1), key=lambda s: (LOAD[p][s], s))[:len(groups)])
        assert len(slots) == len(groups), (D, p, lo, hi)
        for s_, g in zip(slots, groups):
            ev(p, s_, "valu", None, g)

    slot_of = lambda g, k, ib, j: ((g * NK + k) * IB + ib) * GU + j
    EMITS = []
    AHEAD = []
    for p in range(2):
        q = 1 - p
        ev(p, 0, "valu", None, ["v_mov_b32 %s, %s" % (vrd, vrdb)])
        ev(p, 1, "valu", None, ["v_add_u32 %s, s95, %s" % (vrdb, voff0), "v_mov_b32 %s, s95" % vsb])
        if p == 1:
            ev(p, A_SLOT, "valu", None, ["s_waitcnt vmcnt(0)", "s_barrier"])
        ev(p, A_SLOT + 1, "lds", ("meta", q), ["ds_read_b128 %s, %s offset:%d" % (metapair(q), vsb, HB)])
        if not maxmode:
            ev(p, S_VH, "valu", None, ["v_mov_b32 %s, %%[h]" % vh])
        n_rd = 0
        for k in range(NK):
            for ib in range(IB):
                if k >= NSETK:
                    s = slot_of(G - 1, k - NSETK, ib, GU - 1) + 1
                    assert 1 <= s < slot_of(0, k, ib, 0) - 4
                    ev(p, s, "lds", ("frag", p, k, ib), frag_read(k, ib, n_rd, vrd))
                else:
                    s = slot_of(G - 1, NK - NSETK + k, ib, GU - 1) + 1
                    if s == n_half:
                        ev(p, 0, "lds", ("frag", p, k, ib), frag_read(k, ib, n_rd, vrd))
                    else:
                        assert s > A_SLOT + 1
                        ev(p, s, "lds", ("frag", q, k, ib), frag_read(k, ib, n_rd, vrdb))
                        if p == 0:
                            AHEAD.append((k, ib))
                n_rd += 1
        EMITS.append([])
        for g in range(G):
            for j in range(GU):
                u = g * GU + j
                c0, c1 = slot_of(g, NK - 1, 0, j), slot_of(g, NK - 1, 1, j)
                r0, r1 = n_half + slot_of(g, 0, 0, j), n_half + slot_of(g, 0, 1, j)
                a, b = (lambda r: accr(u, 0, r)), (lambda r: accr(u, 1, r))
                rd = RD if n_half - slot_of(0, NK - 1, 0, 0) - RD >= 12 else 2
                # the emission: behind both chains' last MFMAs, in front of chain 0's restart; and where the header register holds THIS
                # half-tile's index: slots (S_VH, n_half) of its own half-tile or [0, S_VH) of the next one
                e_lo = max(c1 + rd, S_VH + 1) if c1 + rd < n_half else c1 + rd
                e_hi = min(r0 - 1 - SPLIT_EMIT, n_half + S_VH - 1 - SPLIT_EMIT)
                if e_lo >= n_half:
                    e_hi = min(e_hi, e_lo + GU + 6)      # (in front of the next half-tile's wait at A: their stores want ~50 slots to be acknowledged)
                elif e_lo <= A_SLOT:
                    e_lo = A_SLOT + 1
                assert e_lo <= e_hi and (e_lo >= n_half or e_lo > S_VH), (D, u, e_lo, e_hi)
                mx = [(c0 + rd, r0 - 1, ["v_max3_f32 %s, %s, %s, %s" % (mt(u), a(0), a(1), a(2))]),
                      (c0 + rd, r0 - 1, ["v_max_f32 %s, %s, %s" % (mt(u), mt(u), a(3))]),
                      (c1 + rd, r1 - 1, ["v_max3_f32 %s, %s, %s, %s" % (mt(u), mt(u), b(0), b(1))]),
                      (c1 + rd, r1 - 1, ["v_max3_f32 %s, %s, %s, %s" % (mt(u), mt(u), b(2), b(3))])]
                mx = [(lo, min(hi, e_hi - (4 - i)), ln) for i, (lo, hi, ln) in enumerate(mx)]
                o = ENTRY * u
                em = ["v_cmpx_gt_f32 vcc, %s, %s" % (mt(u), thr(u)),
                      "buffer_store_dwordx4 %s, %s, %%[rsrc], 0 offen offset:%d" % (acc(u, 0), cur(u), o),
                      "buffer_store_dwordx4 %s, %s, %%[rsrc], 0 offen offset:%d" % (acc(u, 1), cur(u), o + 16),
                      "buffer_store_dword %s, %s, %%[rsrc], 0 offen offset:%d" % (vh, cur(u), o + 32),
                      "v_add_u32 %s, %d, %s" % (cur(u), ESTRIDE, cur(u)),
                      "s_mov_b64 exec, -1"]
                if BRANCH:
                    em = ["v_cmp_gt_f32 vcc, %s, %s" % (mt(u), thr(u)), "s_cbranch_vccz 7f", "s_mov_b64 exec, vcc"] + em[1:] + ["7:"]
                if maxmode:
                    em = ["v_min_f32 %s, %s, %s" % (vh, cur(u), mt(u)), "v_max_f32 %s, %s, %s" % (cur(u), cur(u), mt(u)), "v_max_f32 %s, %s, %s" % (thr(u), thr(u), vh)]
                if ONLY_U >= 0 and u != ONLY_U:
                    em = ["s_nop 0"]
                EMITS[p].append(mx + [(e_lo, e_hi, em)])

    # the maxima and emissions: every operation into the least loaded slot of its window, in order within a user block; the emission of a
    # block is one event (the exec mask is narrowed inside it: nothing else may sit between its lines) -- or two in consecutive slots
    for p in range(2):
        for seq in sorted(EMITS[p], key=lambda q_: q_[0][0]):
            prev = -1
            for i, (lo, hi, lines) in enumerate(seq):
                lo = max(lo, prev + 1)
                hi = min(seq[t][1] - (t - i) for t in range(i, len(seq)))
                assert lo <= hi, (D, p, i, lo, hi)
                load = lambda sl: LOAD[(p + sl // n_half) % 2][sl % n_half]
                last = i == len(seq) - 1
                if last and SPLIT_EMIT:
                    best = min(range(lo, hi + 1), key=lambda sl: (load(sl) + load(sl + 1), sl))
                    ev(p, best, "emitA", None, lines[:3])
                    ev(p, best + 1, "emitB", None, lines[3:])
                else:
                    best = min(range(lo, hi + 1), key=lambda sl: (load(sl), sl))
                    ev(p, best, "emit" if last else "valu", None, lines)
                prev = best

    for p in range(2):
        q = 1 - p
        s0 = 10
        if p == 1:
            ev(p, s0, "valu", None, ["s_cmp_gt_u32 %[h], %[hend]", "s_cbranch_scc1 92f"])
        # (behind the read of the meta entry at A_SLOT + 1: with one group of user blocks -- UB = 8 -- the slot below lies in front of it, and ct would
        # be formed from the entry read two half-tiles ago)
        sq = max(slot_of(G - 1, 0, 1, GU - 1) + 2, A_SLOT + 4)
        # ct = pmax + A nmax + B rmax (A, B: the wave's rounding-residual and norm maxima -- sweep7_kernel)
        ev(p, sq, "check", ("meta", q), ["v_fma_f32 %s, %%[eu], %s, %s" % (ctr(q, 0), metan(q), metap(q)),
                                         "v_fma_f32 %s, %%[eu2], %s, %s" % (ctr(q, 0), metar(q), ctr(q, 0))])
        ev(p, sq + 1, "valu", None, ["v_mov_b32 %s, %s" % (ctr(q, r), ctr(q, 0)) for r in (1, 2, 3)] +
           (["v_max_f32 %s, %s, %s" % (vctmax, vctmax, ctr(q, 0))] if maxmode else []))
        s1 = spread(p, s0 + 1, s0 + 4, [["s_add_u32 s97, %[h], 2"] + slot_addr("s95", "s97")])
        x_odd = (p + PFD) & 1
        step = ["s_add_u32 s80, %%[h], %d" % (PFD + 1), "s_cmp_lt_u32 s80, %[hend]", "s_cselect_b32 s86, %s, 0" % ("s90" if x_odd else "%d" % HB),
                "s_cselect_b32 s87, %s, 0" % ("s91" if x_odd else "16")]
        adv = [["s_add_u32 s84, s84, s86", "s_addc_u32 s85, s85, 0"], ["s_add_u32 s88, s88, s87", "s_addc_u32 s89, s89, 0", "s_mov_b32 %[issued], s80"]]
        # the pieces of h + PFD: even half-tiles early (their wait is the next half-tile's), odd ones behind their own wait
        d_lo, d_hi = (s1, A_SLOT - 8) if (p == 0 and not os.environ.get("V7_DMA_LATE")) else (A_SLOT + 2, n_half - 3)
        assert d_hi > d_lo + 4
        spread_least(p, d_lo, d_hi, [["s_add_u32 s81, %%[h], %d" % PFD] + dma_ops("s81")[0]] + dma_ops("s81")[1:] + [step, flat(adv)])

    def order_events(evs):
        """an open emission (emitA) closes the slot; its second half (emitB) opens the next one"""
        first = [e for e in evs if e[0] == "emitB"]
        last = [e for e in evs if e[0] == "emitA"]
        assert len(first) <= 1 and len(last) <= 1
        return first + [e for e in evs if e[0] not in ("emitA", "emitB")] + last

    def build_body(state_in):
        lg = list(state_in)
        out = []

        def wait_for(tag):
            if tag in lg:
                pos = len(lg) - 1 - lg[::-1].index(tag)
                out.append("s_waitcnt lgkmcnt(%d)" % min(15, len(lg) - 1 - pos))
                del lg[:pos + 1]

        for p in range(2):
            out.append("2%d:" % p)
            for g in range(G):
                for k in range(NK):
                    for ib in range(IB):
                        for j in range(GU):
                            s, u = slot_of(g, k, ib, j), g * GU + j
                            if g == 0 and j == 0:
                                wait_for(("frag", p, k, ib))
                            # (fp16 operands: the funnel's image and user fragments are halves -- pda_item_prep7_*, uprep5_kernel<.., F16 = true>)
                            out.append("v_mfma_f32_16x16x32_f16 %s, %s, %s, %s" % (acc(u, ib), frag(k, ib), usr(u, k), ctq(p) if k == 0 else acc(u, ib)))
                            for kind, tag, lines in order_events(EV[p].get(s, [])):
                                if kind == "lds":
                                    out.extend(lines)
                                    lg.append(tag)
                                elif kind == "check":
                                    wait_for(tag)
                                    out.extend(lines)
                                else:
                                    out.extend(lines)
            out += (["s_waitcnt vmcnt(0)", "s_barrier"] if BAR_EVERY else []) + ["s_add_u32 %[h], %[h], 1"]
        out.append("s_branch 20b")
        return out, lg

    _, st1 = build_body([])
    b2, st2 = build_body(st1)
    b3, st3 = build_body(st2)
    assert st2 == st3 and b2 == b3, D
    assert 2 <= PFD <= NSLOT - 4

    P = []
    P += ["s_mov_b32 %[m0save], m0", "s_waitcnt vmcnt(0) lgkmcnt(0)", "v_mov_b32 %s, 0" % vzero]
    for j in range(PW):
        P.append("v_add_u32 %s, %d, %%[lane16]" % (vgoff(j), 4096 * j))
        P.append("v_add_u32 %s, %%[w1024], %s" % (vgoff(j), vgoff(j)))
    t0r, t1r = atmp(0), atmp(1)
    P += ["v_lshrrev_b32 %s, 4, %%[lane16]" % t0r, "v_and_b32 %s, 15, %s" % (t1r, t0r), "v_lshrrev_b32 %s, 4, %s" % (t0r, t0r)]
    if D >= 128:
        P.append("v_and_b32 %s, 15, %s" % (voff0, t1r))
    else:
        P += ["v_lshrrev_b32 %s, 1, %s" % (voff0, t1r), "v_and_b32 %s, 7, %s" % (voff0, voff0)]
    P += ["v_xor_b32 %s, %s, %s" % (voff0, voff0, t0r), "v_lshlrev_b32 %s, 4, %s" % (voff0, voff0),
          "v_lshl_add_u32 %s, %s, %d, %s" % (voff0, t1r, (2 * D).bit_length() - 1, voff0)]
    P += ["s_mul_i32 s90, %%[nsplit], %d" % (2 * HB), "s_sub_u32 s90, s90, %d" % HB, "s_lshl_b32 s91, %[nsplit], 5", "s_sub_u32 s91, s91, 16"]
    # the cursors: the lane's first entry of every user block (lane16 = 16 lane); the accumulators and maxima: -inf (what the first slots
    # of the entry half-tile test belongs to no half-tile)
    if maxmode:
        P += ["v_mov_b32 %s, 0xff800000" % cur(u) for u in range(UB)] + ["v_mov_b32 %s, 0xff800000" % thr(u) for u in range(UB)] + ["v_mov_b32 %s, 0" % vctmax]
    else:
        P.append("v_mul_u32_u24 %s, %d, %%[lane16]" % (cur(0), LSTRIDE // 16))
        P += ["v_mov_b32 %s, %s" % (cur(u), cur(0)) for u in range(1, UB)]
    P += ["v_mov_b32 v%d, 0xff800000" % r for r in range(ACC0, ACC0 + 8 * UB)]
    P += ["v_mov_b32 %s, 0xff800000" % mt(u) for u in range(UB)]
    P.append("v_mov_b32 %s, -1" % vh)
    P.append("s_mov_b64 s[88:89], %[ufrag]")
    for i in range(UB * NK):
        if i % 4 == 0 and i > 0:
            P += ["s_add_u32 s88, s88, 4096", "s_addc_u32 s89, s89, 0"]
        P.append("global_load_dwordx4 a[%d:%d], %%[lane16], s[88:89] offset:%d" % (4 * i, 4 * i + 3, 1024 * (i % 4)))
    P.append("s_waitcnt vmcnt(0)")
    P += ["5:", "s_add_u32 s97, %%[h], %d" % PFD, "s_cmp_ge_u32 %[issued], s97", "s_cbranch_scc1 6f"]
    P += pointers_from_scratch() + flat(dma_ops("%[issued]")) + ["s_add_u32 %[issued], %[issued], 1", "s_branch 5b", "6:"]
    P += pointers_from_scratch()
    P += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
    P += slot_addr("s97", "%[h]") + ["v_add_u32 %s, s97, %s" % (vrdb, voff0), "v_mov_b32 %s, s97" % vsb, "s_add_u32 s97, %[h], 1"] + slot_addr("s95", "s97")
    P += ["ds_read_b128 %s, %s offset:%d" % (metapair(0), vsb, HB), "s_waitcnt lgkmcnt(0)", "v_fma_f32 %s, %%[eu], %s, %s" % (ctr(0, 0), metan(0), metap(0)),
          "v_fma_f32 %s, %%[eu2], %s, %s" % (ctr(0, 0), metar(0), ctr(0, 0))]
    P += ["v_mov_b32 %s, %s" % (ctr(0, r), ctr(0, 0)) for r in (1, 2, 3)] + ["v_mov_b32 %s, %s" % (ctr(1, r), ctr(0, 0)) for r in range(4)]
    if maxmode:
        P.append("v_max_f32 %s, %s, %s" % (vctmax, vctmax, ctr(0, 0)))
    P += ["v_mov_b32 %s, %s" % (metap(1), metap(0))]
    for i, (k, ib) in enumerate(AHEAD):
        P += frag_read(k, ib, i, vrdb)
    P += ["s_bitcmp1_b32 %[h], 0", "s_cbranch_scc1 9f"]
    for par in range(2):
        P += ["s_waitcnt lgkmcnt(0)", "s_branch 2%df" % par]
        if par == 0:
            P.append("9:")
    # ---- exit: the sweep is over (half-tile hend + 1; what hend and hend + 1 wrote carries a header >= hend: the consumer drops it).
    # The cursors go to the wave's count words: [user block][lane]
    E = ["92:", "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15", "s_mov_b64 exec, -1", "s_mov_b32 m0, %[m0save]",
         "v_lshrrev_b32 %s, 2, %%[lane16]" % atmp(0)]
    E += ["global_store_dword %s, %s, %%[cnt] offset:%d" % (atmp(0), cur(u), 256 * u) for u in range(UB)]
    if maxmode:      # behind the largest maxima [user block][lane]: the second largest, then the wave's largest ct (every lane the same)
        E += ["v_add_u32 %s, %d, %s" % (atmp(0), 256 * UB, atmp(0))]
        E += ["global_store_dword %s, %s, %%[cnt] offset:%d" % (atmp(0), thr(u), 256 * u) for u in range(UB)]
        E += ["v_add_u32 %s, %d, %s" % (atmp(0), 256 * UB, atmp(0)), "global_store_dword %s, %s, %%[cnt]" % (atmp(0), vctmax)]
    E += ["s_waitcnt vmcnt(0)"]
    if os.environ.get("V5_LOADS"):
        for p in range(2):
            print("D=%d UB=%d parity %d fillers per slot: %s" % (D, UB, p, " ".join("%d" % x for x in LOAD[p])), file=sys.stderr)
    return P + b2 + E, THR0, LO_CLOBBER, CUR0


def emit(D, UB, maxmode=False):
    L, THR0, LO, CUR0 = gen(D, UB, maxmode)
    out = []
    out.append("template <>")
    out.append("struct %s<%d, %d> {" % ("Loop7M" if maxmode else "Loop7", D, UB))
    out.append("    static constexpr int kSlotBytes = %d, kPfd = %d, kEntry = %d, kLaneStride = %d, kEntryStride = %d;" %
               (64 * D + (512 if D == 256 else 256), PFD, ENTRY, UB * ENTRY, 64 * UB * ENTRY))
    out.append("    // h: the local half-tile to start at (even).  hend: half-tiles of the launch (even).  rsrc: the wave's list region as a raw buffer; cnt: its")
    out.append("    // count words [user block][lane] (the cursors: lane * kLaneStride + entries * kEntryStride).")
    out.append("    static __device__ __forceinline__ void run(unsigned h, unsigned issued, unsigned hend, unsigned ring, unsigned w1024, unsigned t0, unsigned nsplit,")
    out.append("                                               unsigned imglo, unsigned imghi, unsigned metalo, unsigned metahi, float eu, float eu2, const void* ufrag,")
    if maxmode:
        out.append("                                               float* cnt, unsigned lane16) {     // cnt: [2][user block][lane] the two largest maxima, then [lane] the largest ct")
    else:
        out.append("                                               u32x4 rsrc, unsigned* cnt, const float (&thr)[%d], unsigned lane16) {" % UB)
    out.append("#if defined(__HIP_DEVICE_COMPILE__)")
    out.append("        unsigned m0save;")
    out.append("        asm volatile(")
    for l in L:
        out.append('            "%s\\n\\t"' % l)
    out.append('            : [h] "+&s"(h), [issued] "+&s"(issued), [m0save] "=&s"(m0save)')
    ins = ['[hend] "s"(hend)', '[ring] "s"(ring)', '[w1024] "s"(w1024)', '[t0] "s"(t0)', '[nsplit] "s"(nsplit)',
           '[imglo] "s"(imglo)', '[imghi] "s"(imghi)', '[metalo] "s"(metalo)', '[metahi] "s"(metahi)', '[eu] "s"(eu)', '[eu2] "s"(eu2)', '[ufrag] "s"(ufrag)',
           '[cnt] "s"(cnt)', '[lane16] "v"(lane16)']
    if not maxmode:
        ins += ['[rsrc] "s"(rsrc)'] + ['"{v%d}"(thr[%d])' % (THR0 + u, u) for u in range(UB)]
    out.append("            : " + ", ".join(ins))
    hi = 16 * UB
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"s%d"' % r for r in range(80, 100)] + \
           ['"v%d"' % r for r in range(LO, hi) if maxmode or not THR0 <= r < THR0 + UB] + ['"a%d"' % r for r in range(4 * UB * (D // 32))] + \
           (['"v%d"' % r for r in range(128, 148)] if D == 256 else [])
    out.append("            : " + ", ".join(clob) + ");")
    out.append("#endif")
    out.append("    }")
    out.append("};")
    return "\n".join(out)


def main():
    print("// GENERATED by tools/gen_v7_emit_loop_asm.py -- do not edit.")
    print("#pragma once")
    print("template <int D, int UB> struct Loop7;")
    print("template <int D, int UB> struct Loop7M;")
    for D in (64, 128):
        print(emit(D, 16))
        print(emit(D, 16, True))
    print(emit(256, 8))
    print(emit(256, 8, True))


if __name__ == "__main__":
    main()
