#!/usr/bin/env python
"""Generates pda_amd/csrc/pda_v5_loop_asm.h: the main loop of the "huge" geometry of the sweep (sweep5_kernel, pda_v5_sweep.h) as ONE
inline-asm statement per (re-)entry -- d = 64 / 128, popularity head, dense sweep in visiting order.

The mapping (measured first in tools/ubench/mfma_struct5.hip, V4): four waves per workgroup, one per SIMD, 512 registers
each.  A wave owns 256 users: their bf16 rows sit in AGPRs as eight B operands of 32 users (a[0 .. 4 UA NK)), the eight accumulators
in v[128:255].  Every item fragment read from the LDS (one ds_read_b128) feeds EIGHT MFMAs, and the product is transposed -- A = item
fragment, B = user fragment -- so that a lane of an accumulator holds 16 items of ONE user: the threshold test is a per-lane compare
(7 v_max3 + v_max + v_add + v_cmp per 32 x 32 block, placed in the MFMA shadow), and the folded test k-step of generation 4 (1/9 of
all MFMAs at d = 128, 1/5 at d = 64) is gone.  The waves load the tiles themselves (LDS-DMA, two 1 KiB pieces per wave and half-tile
at d = 128), hand them over through per-wave "landed" / "released" words in the LDS, and leave the statement when a half-tile raised a
flag (the C++ around it rescores that half-tile exactly and re-enters) or when the sweep is over.

    python tools/gen_v5_loop_asm.py > pda_amd/csrc/pda_v5_loop_asm.h
"""

UA = 8
ACC0, FRAG0 = 128, 112
THR0, M0T, SW0 = 80, 88, 96          # thr[8], m[8], sw[NK <= 8]
CT0, META0, ATMP0 = 104, 106, 110    # ct[2], meta pairs (pmax, nmax)[2], address temporaries[2]
POLL0, VREL, VGOFF0 = 72, 76, 77     # poll quad, value register for sync-word stores, DMA lane offsets [2], v79 = zero
VZERO = 79
LO_CLOBBER = 72
NSLOT = 8
# hard SGPRs (clobbered): temporaries s84..s91, flag s[92:93], s94 cur slot, s95 next slot, s96 spin counter, s97..s99 temporaries
SPIN_MAX = 1 << 24


def acc(u):
    return "v[%d:%d]" % (ACC0 + 16 * u, ACC0 + 16 * u + 15)


def accr(u, r):
    return "v%d" % (ACC0 + 16 * u + r)


def gen(D):
    NK = D // 16
    HB = 32 * 2 * D                      # one half-tile: 32 rows of 2 D bytes, 16-byte chunks XOR-swizzled (no padding)
    HBL = HB.bit_length() - 1
    R = 4                                # fragment registers: slot k % 4 (a fragment is read PF = 3 steps ahead of its eight MFMAs)
    PF = 3                               # steps between a fragment's read and its first MFMA
    PW = HB // 1024 // 4                 # LDS-DMA pieces per wave and half-tile (2 at d = 128, 1 at d = 64)
    CD = 1 if NK == 8 else 2             # at half-tile h the wave confirms its pieces of h + CD and polls for everybody's
    PFD = CD + 1                         # ... and issues the pieces of h + PFD
    OPS = PW + 1                         # vector-memory operations per half-tile: the pieces and the meta pair of h + 1
    usr = lambda u, k: "a[%d:%d]" % (4 * (u * NK + k), 4 * (u * NK + k) + 3)
    fslot = lambda p, k: (p * NK + k) % R
    frag = lambda p, k: "v[%d:%d]" % (FRAG0 + 4 * fslot(p, k), FRAG0 + 4 * fslot(p, k) + 3)
    thr = lambda u: "v%d" % (THR0 + u)
    mt = lambda u: "v%d" % (M0T + u)
    sw = lambda k: "v%d" % (SW0 + k)
    ct = lambda p: "v%d" % (CT0 + p)
    metap = lambda p: "v%d" % (META0 + 2 * p)
    metan = lambda p: "v%d" % (META0 + 2 * p + 1)
    metapair = lambda p: "v[%d:%d]" % (META0 + 2 * p, META0 + 2 * p + 1)
    poll = "v[%d:%d]" % (POLL0, POLL0 + 3)
    pollr = lambda i: "v%d" % (POLL0 + i)
    vrel, vzero = "v%d" % VREL, "v%d" % VZERO
    vgoff = lambda j: "v%d" % (VGOFF0 + j)
    n_half = NK * UA                     # MFMA slots per half-tile

    # ---- helpers emitting instruction GROUPS (a group stays together; groups of one event are spread over consecutive slots) --------
    # hard SGPRs: s80 x, s81 x', s82 T, s83 LDS piece base, s[84:85] piece source | s86 y', s87 T, s[88:89] meta source | s[92:93] flag |
    # s94 / s95 current / next slot | s96 spin count | s97 scratch | s98 / s99 poll: minimum / needed
    def slot_addr(dst, idx_sgpr):
        # dst = ring + (idx & 7) * HB
        return ["s_and_b32 %s, %s, %d" % (dst, idx_sgpr, NSLOT - 1), "s_lshl_b32 %s, %s, %d" % (dst, dst, HBL), "s_add_u32 %s, %s, %%[ring]" % (dst, dst)]

    def tile_of(dst, xc, x):
        # xc = min(x, hend - 1); dst = T0 + (xc >> 1) S   (the 64-item tile of local half-tile x, clamped to the split's last)
        return ["s_sub_u32 %s, %%[hend], 1" % xc, "s_min_u32 %s, %s, %s" % (xc, x, xc), "s_lshr_b32 %s, %s, 1" % (dst, xc),
                "s_mul_i32 %s, %s, %%[nsplit]" % (dst, dst), "s_add_u32 %s, %s, %%[t0]" % (dst, dst)]

    def dma_issue(x_lines):
        """the wave's PW pieces of local half-tile s80 (set by x_lines) into slot s80 & 7; s80 may run past the end (clamped source)"""
        G = [x_lines + tile_of("s82", "s81", "s80")]
        G.append(["s_mul_hi_u32 s85, s82, %d" % (2 * HB), "s_mul_i32 s84, s82, %d" % (2 * HB), "s_and_b32 s81, s81, 1", "s_lshl_b32 s81, s81, %d" % HBL])
        G.append(["s_add_u32 s84, s84, s81", "s_addc_u32 s85, s85, 0", "s_add_u32 s84, s84, %[imglo]", "s_addc_u32 s85, s85, %[imghi]"])
        G.append(slot_addr("s83", "s80") + ["s_add_u32 s83, s83, %[w1024]"])
        for j in range(PW):
            G.append(["s_add_u32 m0, s83, %d" % (4096 * j), "s_nop 0", "global_load_lds_dwordx4 %s, s[84:85]" % vgoff(j)])
        return G

    def meta_issue(y_lines, p):
        """(pmax, nmax) of local half-tile s97 (set by y_lines) into the meta pair of parity p"""
        G = [y_lines + tile_of("s87", "s86", "s97")]
        G.append(["s_lshl_b32 s87, s87, 1", "s_and_b32 s86, s86, 1", "s_add_u32 s87, s87, s86", "s_lshl_b32 s87, s87, 3"])
        G.append(["s_add_u32 s88, %[metalo], s87", "s_addc_u32 s89, %[metahi], 0", "global_load_dwordx2 %s, %s, s[88:89]" % (metapair(p), vzero)])
        return G

    flat = lambda G: [l for g in G for l in g]
    SPINS = []                # out-of-line spin loops

    def poll_read(word_off):
        return ["ds_read_b128 %s, %%[syncv] offset:%d" % (poll, word_off)]

    def poll_min_lines():
        return ["v_min_u32 %s, %s, %s" % (pollr(0), pollr(0), pollr(1)), "v_min3_u32 %s, %s, %s, %s" % (pollr(0), pollr(0), pollr(2), pollr(3)),
                "v_readfirstlane_b32 s98, %s" % pollr(0)]

    def poll_check(word_off, need_expr_lines, lbl):
        """(behind the counted wait for the poll's read) min over the four waves' words >= s99?  else the out-of-line spin `lbl`, which
        comes back to `lbl + 1` once it is (or leaves through 90 when its bounded count runs out)"""
        L = list(need_expr_lines) + poll_min_lines() + ["s_cmp_lt_u32 s98, s99", "s_cbranch_scc1 %df" % lbl, "%d:" % (lbl + 1)]
        SPINS.append(["%d:" % lbl, "s_mov_b32 s96, 0", "%d:" % (lbl + 2), "s_sleep 1", "ds_read_b128 %s, %%[syncv] offset:%d" % (poll, word_off),
                      "s_waitcnt lgkmcnt(0)"] + poll_min_lines() + ["s_cmp_ge_u32 s98, s99", "s_cbranch_scc1 %db" % (lbl + 1), "s_add_u32 s96, s96, 1",
                      "s_cmp_lt_u32 s96, %d" % SPIN_MAX, "s_cbranch_scc1 %db" % (lbl + 2), "s_branch 90f"])
        return L

    def poll_blocking(word_off, need_expr_lines, lbl):
        """the same, waiting in line (prologue)"""
        return list(need_expr_lines) + ["s_mov_b32 s96, 0", "%d:" % lbl, "ds_read_b128 %s, %%[syncv] offset:%d" % (poll, word_off), "s_waitcnt lgkmcnt(0)"] + \
            poll_min_lines() + ["s_cmp_ge_u32 s98, s99", "s_cbranch_scc1 %df" % (lbl + 1), "s_sleep 1", "s_add_u32 s96, s96, 1",
                                "s_cmp_lt_u32 s96, %d" % SPIN_MAX, "s_cbranch_scc1 %db" % lbl, "s_branch 90f", "%d:" % (lbl + 1)]

    # ---- the loop body: parity 0 half-tile, parity 1 half-tile --------------------------------------------------------------
    EV = [dict(), dict()]      # events per parity and slot

    def ev(p, s, kind, tag, lines):
        assert 0 <= s < n_half, (D, s)
        EV[p].setdefault(s, []).append((kind, tag, lines))

    def spread(p, s0, groups):
        for i, g in enumerate(groups):
            ev(p, s0 + i, "valu", None, g)
        return s0 + len(groups)

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
            for k in range(NK):
                for u in range(UA):
                    s = k * UA + u
                    wait_for(("frag", p, k))
                    out.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc(u), frag(p, k), usr(u, k), "0" if k == 0 else acc(u)))
                    fill = []
                    # fragment reads: (p, k2) is first used at step k2, read PF steps earlier, behind the step's second MFMA
                    if u == 1:
                        k2 = k + PF
                        if k2 < NK:
                            fill.append(("lds", ("frag", p, k2), ["v_add_u32 v%d, s94, %s" % (ATMP0 + (k & 1), sw(k2)),
                                                                "ds_read_b128 %s, v%d" % (frag(p, k2), ATMP0 + (k & 1))]))
                        else:                                 # the next half-tile's first fragments, from its slot (s95)
                            fill.append(("lds", ("frag", 1 - p, k2 - NK), ["v_add_u32 v%d, s95, %s" % (ATMP0 + (k & 1), sw(k2 - NK)),
                                                                         "ds_read_b128 %s, v%d" % (frag(1 - p, k2 - NK), ATMP0 + (k & 1))]))
                    fill += EV[p].get(s, [])
                    for kind, tag, lines in fill:
                        if kind == "lds":
                            out.extend(lines)
                            lg.append(tag)
                        elif kind == "check":               # the poll's read must have returned
                            wait_for(tag)
                            out.extend(lines)
                        else:
                            out.extend(lines)
            # end of the half-tile: the next one becomes current
            out += ["s_add_u32 %[h], %[h], 1", "s_mov_b32 s94, s95", "s_add_u32 s97, %[h], 1"] + slot_addr("s95", "s97")
        out.append("s_branch 20b")
        return out, lg

    for p in range(2):
        q = 1 - p
        # tests: chain u is final behind slot (NK - 1) UA + u of this half-tile and restarts UA slots later: the eight maxima (which read the
        # accumulator) in the six slots behind the next-but-one MFMA; the add, compare and OR (which do not) one chain per slot from slot
        # UA of the next half-tile on.  (>= 2 MFMAs between an accumulator's last MFMA and its first VALU read: the XDL write has landed.)
        for u in range(UA):
            s = (NK - 1) * UA + u
            ops = [["v_max3_f32 %s, %s, %s, %s" % (mt(u), accr(u, 0), accr(u, 1), accr(u, 2))]]
            for r in range(3, 15, 2):
                ops.append(["v_max3_f32 %s, %s, %s, %s" % (mt(u), mt(u), accr(u, r), accr(u, r + 1))])
            ops.append(["v_max_f32 %s, %s, %s" % (mt(u), mt(u), accr(u, 15))])
            first, nwin = s + 2, UA - 2
            for i, o in enumerate(ops):
                sl = first + (i * nwin) // len(ops)
                if sl < n_half:
                    ev(p, sl, "valu", None, o)
                else:
                    ev(q, sl - n_half, "valu", None, o)          # ... spilling into the first slots of the next half-tile
            # flag <=> max(m + ct, pmax) > thr: the product could reach the threshold, or (clamp) a popularity of the half-tile beats it --
            # a head below 1 x pop may qualify whatever the product says
            ev(q, UA + u, "valu", None, ["v_add_f32 %s, %s, %s" % (mt(u), mt(u), ct(p)), "v_max_f32 %s, %s, %s" % (mt(u), mt(u), metap(p)),
                                         "v_cmp_gt_f32 vcc, %s, %s" % (mt(u), thr(u)), "s_or_b64 s[92:93], s[92:93], vcc"])
        # the flags of the previous half-tile are complete: leave if one is set, else release it
        s = 2 * UA + 1
        chk = ["s_cmp_lg_u64 s[92:93], 0", "s_cbranch_scc1 91f"]
        if p == 0:
            chk += ["s_cmp_ge_u32 %[h], %[hend]", "s_cbranch_scc1 92f"]          # (parity 0 only: the sweep ends on a whole 64-item tile)
        chk += ["v_mov_b32 %s, %%[h]" % vrel]
        ev(p, s, "valu", None, chk)
        ev(p, s, "lds", ("rel", p), ["ds_write_b32 %%[syncw], %s offset:16" % vrel])
        # is the slot of half-tile h + PFD free?  (everybody has released h + PFD - NSLOT): the read now, the look at it a step later
        ev(p, s + 1, "lds", ("pollr", p), poll_read(16))
        # ct of THIS half-tile from its meta pair (pmax, nmax): the slack between the bf16 product and a bound of the exact head
        ev(p, s + 2, "valu", None, ["s_waitcnt vmcnt(%d)" % PW, "v_fma_f32 %s, %%[eu], %s, %s" % (ct(p), metan(p), metap(p))])
        # the meta pair of h + 1, then (behind the check that their slot is free) the pieces of h + PFD
        s = spread(p, s + 3, meta_issue(["s_add_u32 s97, %[h], 1"], q))
        need = ["s_add_u32 s99, %%[h], %d" % (PFD + 1), "s_sub_u32 s99, s99, %d" % NSLOT, "s_max_i32 s99, s99, 0"]
        s = max(s, 2 * UA + 2 + (5 if NK == 8 else 4))
        ev(p, s, "check", ("pollr", p), poll_check(16, need, 30 + 4 * p))
        G = dma_issue(["s_add_u32 s80, %%[h], %d" % PFD])
        G[-1] = G[-1] + ["s_add_u32 %[issued], s80, 1"]
        s = spread(p, s + 1, G)
        # own pieces of h + CD have landed (everything older than the last PFD - CD rounds): say so, then look at everybody's
        ev(p, s, "valu", None, ["s_waitcnt vmcnt(%d)" % (OPS * (PFD - CD)), "s_add_u32 s97, %%[h], %d" % (CD + 1), "v_mov_b32 %s, s97" % vrel])
        ev(p, s, "lds", ("lan", p), ["ds_write_b32 %%[syncw], %s" % vrel])
        ev(p, s + 1, "lds", ("polll", p), poll_read(0))
        s2 = min(n_half - 1, s + 1 + (UA if NK == 8 else 3))
        print('// d = %d parity %d: landed check at slot %d of %d' % (D, p, s2, n_half)) if False else None
        if NK == 8:
            assert s2 <= (NK - PF) * UA, s2             # (CD = 1: before the first read of the next half-tile)
        ev(p, s2, "check", ("polll", p), poll_check(0, ["s_add_u32 s99, %%[h], %d" % (CD + 1)], 40 + 4 * p))

    # steady state of the counted LDS waits
    _, st1 = build_body([])
    b2, st2 = build_body(st1)
    b3, st3 = build_body(st2)
    assert st2 == st3 and b2 == b3, D

    # ---- prologue (every entry) ------------------------------------------------------------------------------------------------
    P = []
    P += ["s_mov_b32 %[m0save], m0", "s_waitcnt vmcnt(0) lgkmcnt(0)", "v_mov_b32 %s, 0" % vzero]
    for u in range(UA):
        P.append("v_mov_b32 %s, %%[thr%d]" % (thr(u), u))
    for j in range(PW):
        P.append("v_add_u32 %s, %d, %%[lane16]" % (vgoff(j), 4096 * j))
        P.append("v_add_u32 %s, %%[w1024], %s" % (vgoff(j), vgoff(j)))
    # swizzled fragment offsets of the lane: row r = lane & 31, half hh = lane >> 5, chunk c = 2 k + hh -> r * 2D + ((c ^ swz(r)) << 4)
    P += ["v_lshrrev_b32 v%d, 4, %%[lane16]" % ATMP0, "v_and_b32 v%d, 31, v%d" % (ATMP0 + 1, ATMP0), "v_lshrrev_b32 v%d, 5, v%d" % (ATMP0, ATMP0)]
    if D >= 128:
        P.append("v_and_b32 %s, 15, v%d" % (pollr(0), ATMP0 + 1))                      # swz(r) = r & 15
    else:
        P += ["v_lshrrev_b32 %s, 1, v%d" % (pollr(0), ATMP0 + 1), "v_and_b32 %s, 7, %s" % (pollr(0), pollr(0))]      # (r >> 1) & 7
    P.append("v_lshlrev_b32 v%d, %d, v%d" % (ATMP0 + 1, (2 * D).bit_length() - 1, ATMP0 + 1))                         # r * 2D
    for k in range(NK):
        P += ["v_add_u32 %s, %d, v%d" % (sw(k), 2 * k, ATMP0), "v_xor_b32 %s, %s, %s" % (sw(k), sw(k), pollr(0)),
              "v_lshl_add_u32 %s, %s, 4, v%d" % (sw(k), sw(k), ATMP0 + 1)]
    # the wave's user fragments -> AGPRs, at EVERY entry: the compiler uses AGPRs as spill space between the statements
    P.append("s_mov_b64 s[88:89], %[ufrag]")
    for i in range(UA * NK):
        if i % 4 == 0 and i > 0:
            P += ["s_add_u32 s88, s88, 4096", "s_addc_u32 s89, s89, 0"]
        P.append("global_load_dwordx4 a[%d:%d], %%[lane16], s[88:89] offset:%d" % (4 * i, 4 * i + 3, 1024 * (i % 4)))
    P.append("s_waitcnt vmcnt(0)")
    # everything this wave has issued has landed (vmcnt(0) above): say so
    P += ["v_mov_b32 %s, %%[issued]" % vrel, "ds_write_b32 %%[syncw], %s" % vrel]
    # catch up with the loads: the pieces of half-tiles issued .. h + PFD - 1 (first entry: all of them)
    P += ["5:", "s_add_u32 s97, %%[h], %d" % PFD, "s_cmp_ge_u32 %[issued], s97", "s_cbranch_scc1 6f"]
    need3 = ["s_add_u32 s99, %[issued], 1", "s_sub_u32 s99, s99, %d" % NSLOT, "s_max_i32 s99, s99, 0"]
    P += poll_blocking(16, need3, 50)
    P += flat(dma_issue(["s_mov_b32 s80, %[issued]"])) + ["s_add_u32 %[issued], %[issued], 1", "s_branch 5b", "6:"]
    # the meta pair of h into the pair of its parity (that of h + 1 is requested by half-tile h itself)
    P += ["s_bitcmp1_b32 %[h], 0", "s_cbranch_scc1 7f"]
    P += flat(meta_issue(["s_mov_b32 s97, %[h]"], 0)) + ["s_branch 8f", "7:"]
    P += flat(meta_issue(["s_mov_b32 s97, %[h]"], 1)) + ["8:"]
    P += ["s_waitcnt vmcnt(0)", "v_mov_b32 %s, %%[issued]" % vrel, "ds_write_b32 %%[syncw], %s" % vrel]
    need4 = ["s_add_u32 s99, %%[h], %d" % (CD + 1), "s_min_u32 s99, s99, %[issued]"]
    P += poll_blocking(0, need4, 52)
    P += slot_addr("s94", "%[h]") + ["s_add_u32 s97, %[h], 1"] + slot_addr("s95", "s97")
    P += ["s_mov_b64 s[92:93], 0"]
    # the fragments the steady-state body expects in flight on entry, for either parity (then drained: the counted waits are merely
    # conservative in the first pass)
    P += ["s_bitcmp1_b32 %[h], 0", "s_cbranch_scc1 9f"]
    for par in range(2):
        for tag in st2:
            if tag[0] == "frag":
                # st2 is the state in front of the parity-0 half-tile; in front of the parity-1 half-tile it is the same with parities flipped
                pp = tag[1] ^ par
                P += ["v_add_u32 v%d, s94, %s" % (ATMP0, sw(tag[2])), "ds_read_b128 %s, v%d" % (frag(pp, tag[2]), ATMP0)]
        # (the first slots of the entry half-tile carry the tail of the PREVIOUS half-tile's tests: -inf makes them fail)
        P += ["v_mov_b32 %s, 0xff800000" % ct(1 - par), "v_mov_b32 %s, 0xff800000" % metap(1 - par), "s_waitcnt lgkmcnt(0)", "s_branch 2%df" % par]
        if par == 0:
            P.append("9:")
    # ---- exits ------------------------------------------------------------------------------------------------------------------
    E = []
    for lines in SPINS[:4]:
        E += lines
    drain = ["s_waitcnt vmcnt(0) lgkmcnt(0)", "v_mov_b32 %s, %%[issued]" % vrel, "ds_write_b32 %%[syncw], %s" % vrel, "s_waitcnt lgkmcnt(0)",
             "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15", "s_mov_b32 m0, %[m0save]"]
    E += ["90:", "s_mov_b32 %[reason], 2"] + drain + ["s_branch 99f"]                    # a bounded spin ran out: protocol error
    E += ["91:", "s_mov_b32 %[reason], 1"] + drain + ["s_branch 99f"]                    # half-tile h - 1 raised a flag (h is the one in progress)
    E += ["92:", "s_mov_b32 %[reason], 0"] + drain + ["99:"]                             # the sweep is over
    return P + b2 + E, st2


def emit(D):
    L, _ = gen(D)
    NK = D // 16
    out = []
    out.append("template <>")
    out.append("struct Loop5<%d> {" % D)
    out.append("    // h: the local half-tile to run next (in: where to (re)start; out: the half-tile in progress when the statement left).")
    out.append("    // issued: half-tiles whose pieces this wave has issued.  reason: 0 = the sweep is over, 1 = half-tile h - 1 raised a flag, 2 = a spin ran out.")
    out.append("    static __device__ __forceinline__ void run(unsigned& h, unsigned& issued, unsigned& reason, unsigned hend, unsigned ring, unsigned syncv, unsigned syncw, unsigned w1024,")
    out.append("                                               unsigned t0, unsigned nsplit, unsigned imglo, unsigned imghi, unsigned metalo, unsigned metahi, float eu, const void* ufrag,")
    out.append("                                               const float (&thr)[8], unsigned lane16) {")
    out.append("#if defined(__HIP_DEVICE_COMPILE__)")
    out.append("        unsigned m0save;")
    out.append("        asm volatile(")
    for l in L:
        out.append('            "%s\\n\\t"' % l)
    out.append('            : [h] "+s"(h), [issued] "+s"(issued), [reason] "=&s"(reason), [m0save] "=&s"(m0save)')
    ins = ['[hend] "s"(hend)', '[ring] "s"(ring)', '[syncv] "v"(syncv)', '[syncw] "v"(syncw)', '[w1024] "s"(w1024)', '[t0] "s"(t0)', '[nsplit] "s"(nsplit)',
           '[imglo] "s"(imglo)', '[imghi] "s"(imghi)', '[metalo] "s"(metalo)', '[metahi] "s"(metahi)', '[eu] "s"(eu)', '[ufrag] "s"(ufrag)', '[lane16] "v"(lane16)']
    ins += ['[thr%d] "v"(thr[%d])' % (u, u) for u in range(8)]
    out.append("            : " + ", ".join(ins))
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"s%d"' % r for r in range(80, 100)] + ['"v%d"' % r for r in range(LO_CLOBBER, 256)] + \
           ['"a%d"' % r for r in range(4 * UA * NK)]
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
