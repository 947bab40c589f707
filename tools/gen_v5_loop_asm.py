#!/usr/bin/env python
"""Generates pda_amd/csrc/pda_v5_loop_asm.h: the main loop of the "huge" geometry of the sweep (sweep5_kernel, pda_v5_sweep.h) as ONE
inline-asm statement per (re-)entry -- d = 64 / 128, popularity head, dense sweep in visiting order.

The mapping (measured first in tools/ubench/mfma_struct5.hip, V4): four waves per workgroup, one per SIMD, 512 registers
each.  A wave owns 256 users: their bf16 rows sit in AGPRs as eight B operands of 32 users (a[0 .. 4 UA NK)), the eight accumulators
in v[128:255].  Every item fragment read from the LDS (one ds_read_b128) feeds EIGHT MFMAs, and the product is transposed -- A = item
fragment, B = user fragment -- so that a lane of an accumulator holds 16 items of ONE user: the threshold test is a per-lane compare
(7 v_max3 + v_max, then v_add + v_max + v_cmp per 32 x 32 block, placed in the MFMA shadow), and the folded test k-step of generation
4 (1/9 of all MFMAs at d = 128, 1/5 at d = 64) is gone.

The four waves run in step, ONE s_barrier per 32-item half-tile:
  * half-tile h: every wave issues its LDS-DMA pieces of half-tile h + 3 (and the half-tile's 16-byte meta entry), runs the 8 NK MFMAs
    of h (the first three fragments were read during h - 1), tests the accumulators of h - 1 in the shadow, ORs "h - 1 raised a flag
    in one of my lanes" into a shared LDS word, waits for its own pieces of h + 2 (a counted vmcnt: all of the loop's vector-memory
    operations are LDS-DMA, in order among themselves) and meets the others at the barrier: behind it h + 2 has landed for everybody,
    and everybody is done reading h;
  * the shared flag word of h - 2 is read behind the barrier: when it is set, ALL four waves leave the statement at the same place;
    the C++ around it scores h - 2 and h - 1 again with compiler-visible MFMAs (h - 1's flags were still in flight), rescores the
    candidates exactly and re-enters at h.  (The first version let every wave leave on its own, with per-wave "landed" / "released"
    words and polls in the stream: half-tiles were read before a slower wave's pieces had landed.)

    python tools/gen_v5_loop_asm.py > pda_amd/csrc/pda_v5_loop_asm.h
"""

UA = 8
ACC0, FRAG0 = 128, 112
THR0, M0T, SW0 = 80, 88, 96          # thr[8], m[8], sw[NK <= 8]
CT0, META0, ATMP0 = 104, 106, 110    # ct[2], meta pairs (pmax, nmax)[2], address temporaries[2]
VFLAG, VTMP, VGOFF0, VZERO = 72, 73, 77, 79   # the shared flag word read back, a scratch register, DMA lane offsets [2], zero
LO_CLOBBER = 72
NSLOT = 8
PFD = 3                              # half-tile h issues the pieces of h + PFD


def acc(u):
    return "v[%d:%d]" % (ACC0 + 16 * u, ACC0 + 16 * u + 15)


def accr(u, r):
    return "v%d" % (ACC0 + 16 * u + r)


def gen(D):
    NK = D // 16
    HB = 32 * 2 * D                      # one half-tile: 32 rows of 2 D bytes, 16-byte chunks XOR-swizzled (no padding)
    HBL = HB.bit_length() - 1
    R = 4                                # fragment registers: slot k % 4 (a fragment is read PF = 3 steps ahead of its eight MFMAs)
    PF = 3
    PW = HB // 1024 // 4                 # LDS-DMA pieces per wave and half-tile (2 at d = 128, 1 at d = 64)
    OPS = PW + 1                         # vector-memory operations per half-tile, ALL of them LDS-DMA (in order among themselves)
    usr = lambda u, k: "a[%d:%d]" % (4 * (u * NK + k), 4 * (u * NK + k) + 3)
    frag = lambda k: "v[%d:%d]" % (FRAG0 + 4 * (k % R), FRAG0 + 4 * (k % R) + 3)
    thr = lambda u: "v%d" % (THR0 + u)
    mt = lambda u: "v%d" % (M0T + u)
    sw = lambda k: "v%d" % (SW0 + k)
    ct = lambda p: "v%d" % (CT0 + p)
    metap = lambda p: "v%d" % (META0 + 2 * p)
    metan = lambda p: "v%d" % (META0 + 2 * p + 1)
    metapair = lambda p: "v[%d:%d]" % (META0 + 2 * p, META0 + 2 * p + 1)
    vflag, vtmp, vzero = "v%d" % VFLAG, "v%d" % VTMP, "v%d" % VZERO
    vgoff = lambda j: "v%d" % (VGOFF0 + j)
    n_half = NK * UA                     # MFMA slots per half-tile

    # hard SGPRs: s80 x, s81 x', s82 T, s83 LDS piece base, s[84:85] piece source, s86 / s87 scratch, s[88:89] meta source | s[92:93] the
    # wave's own flags | s94 / s95 current / next slot | s97 scratch | s98 the shared flag word
    def slot_addr(dst, idx_sgpr):
        return ["s_and_b32 %s, %s, %d" % (dst, idx_sgpr, NSLOT - 1), "s_lshl_b32 %s, %s, %d" % (dst, dst, HBL), "s_add_u32 %s, %s, %%[ring]" % (dst, dst)]

    def tile_of(dst, xc, x):
        # xc = min(x, hend - 1); dst = T0 + (xc >> 1) S   (the 64-item tile of local half-tile x, clamped to the split's last)
        return ["s_sub_u32 %s, %%[hend], 1" % xc, "s_min_u32 %s, %s, %s" % (xc, x, xc), "s_lshr_b32 %s, %s, 1" % (dst, xc),
                "s_mul_i32 %s, %s, %%[nsplit]" % (dst, dst), "s_add_u32 %s, %s, %%[t0]" % (dst, dst)]

    def dma_issue(x_lines):
        """the wave's PW pieces of local half-tile s80 (set by x_lines) into slot s80 & 7, and the half-tile's 16-byte meta entry (pmax,
        nmax, 0, 0) into the LDS meta ring (every wave loads it: the same bytes to the same place -- all waves issue the same number of
        operations); s80 may run past the end (clamped source)"""
        G = [x_lines + tile_of("s82", "s81", "s80")]
        G.append(["s_mul_hi_u32 s85, s82, %d" % (2 * HB), "s_mul_i32 s84, s82, %d" % (2 * HB), "s_lshl_b32 s86, s82, 1", "s_and_b32 s81, s81, 1",
                  "s_add_u32 s86, s86, s81", "s_lshl_b32 s86, s86, 4", "s_add_u32 s88, %[metalo], s86", "s_addc_u32 s89, %[metahi], 0", "s_lshl_b32 s81, s81, %d" % HBL])
        G.append(["s_add_u32 s84, s84, s81", "s_addc_u32 s85, s85, 0", "s_add_u32 s84, s84, %[imglo]", "s_addc_u32 s85, s85, %[imghi]"] +
                 slot_addr("s83", "s80") + ["s_add_u32 s83, s83, %[w1024]"])
        for j in range(PW):
            G.append(["s_add_u32 m0, s83, %d" % (4096 * j), "s_nop 0", "global_load_lds_dwordx4 %s, s[84:85]" % vgoff(j)])
        G.append(["s_and_b32 s87, s80, %d" % (NSLOT - 1), "s_lshl_b32 s87, s87, 4", "s_add_u32 m0, s87, %[metalds]", "s_mov_b64 exec, 1",
                  "global_load_lds_dwordx4 %s, s[88:89]" % vzero, "s_mov_b64 exec, -1", "s_add_u32 %[issued], s80, 1"])
        return G

    def meta_read(idx_sgpr, p):
        """(pmax, nmax) of local half-tile idx_sgpr from the LDS meta ring into the meta pair of parity p (a broadcast read)"""
        return ["s_and_b32 s97, %s, %d" % (idx_sgpr, NSLOT - 1), "s_lshl_b32 s97, s97, 4", "s_add_u32 s97, s97, %[metalds]", "v_mov_b32 v%d, s97" % (ATMP0 + 1),
                "ds_read_b64 %s, v%d" % (metapair(p), ATMP0 + 1)]

    def flag_addr(delta):
        """s97 := LDS address of the shared flag word of half-tile h + delta (a ring of four words)"""
        return ["s_add_u32 s97, %%[h], %d" % (delta + 8), "s_and_b32 s97, s97, 3", "s_lshl_b32 s97, s97, 2", "s_add_u32 s97, s97, %[flags]"]

    flat = lambda G: [l for g in G for l in g]

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
                    out.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc(u), frag(k), usr(u, k), "0" if k == 0 else acc(u)))
                    fill = []
                    # fragment reads: (p, k2) is first used at step k2, read PF steps earlier, behind the step's second MFMA
                    if u == 1:
                        k2 = k + PF
                        if k2 < NK:
                            fill.append(("lds", ("frag", p, k2), ["v_add_u32 v%d, s94, %s" % (ATMP0 + (k & 1), sw(k2)),
                                                                "ds_read_b128 %s, v%d" % (frag(k2), ATMP0 + (k & 1))]))
                        else:                                 # the next half-tile's first fragments, from its slot (s95)
                            fill.append(("lds", ("frag", 1 - p, k2 - NK), ["v_add_u32 v%d, s95, %s" % (ATMP0 + (k & 1), sw(k2 - NK)),
                                                                         "ds_read_b128 %s, v%d" % (frag(k2 - NK), ATMP0 + (k & 1))]))
                    fill += EV[p].get(s, [])
                    for kind, tag, lines in fill:
                        if kind == "lds":
                            out.extend(lines)
                            lg.append(tag)
                        elif kind == "check":               # an LDS read must have returned
                            wait_for(tag)
                            out.extend(lines)
                        else:
                            out.extend(lines)
            # end of the half-tile: my pieces of h + 2 have landed (everything but the OPS operations of h + 3), and so will everybody's
            # behind the barrier; everybody is done reading h.  The next half-tile becomes current.
            out += ["s_waitcnt vmcnt(%d)" % OPS, "s_barrier", "s_add_u32 %[h], %[h], 1", "s_mov_b32 s94, s95", "s_add_u32 s97, %[h], 1"] + slot_addr("s95", "s97")
        out.append("s_branch 20b")
        return out, lg

    for p in range(2):
        q = 1 - p
        # tests: chain u is final behind slot (NK - 1) UA + u of this half-tile and restarts UA slots later: the eight maxima (which read the
        # accumulator) in the six slots behind the next-but-one MFMA; the add, maximum, compare and OR (which do not) one chain per slot from
        # slot UA of the next half-tile on.  (>= 2 MFMAs between an accumulator's last MFMA and its first VALU read: the XDL write has landed.)
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
        # slot 0: the shared flag word of h - 2 (complete behind the barrier that ended h - 1); slot 3: the meta pair of h
        ev(p, 0, "lds", ("flag", p), flag_addr(-2) + ["v_mov_b32 %s, s97" % vtmp, "ds_read_b32 %s, %s" % (vflag, vtmp)])
        ev(p, 3, "lds", ("meta", p), meta_read("%[h]", p))
        # slot UA: set -> everybody leaves here (h - 2 and h - 1 are scored again outside); parity 1 only: the sweep is over once the flags
        # of its last half-tile (hend - 1, looked at in hend + 1) have been seen
        chk = ["v_readfirstlane_b32 s98, %s" % vflag, "s_cmp_lg_u32 s98, 0", "s_cbranch_scc1 91f"]
        if p == 1:
            chk += ["s_cmp_gt_u32 %[h], %[hend]", "s_cbranch_scc1 92f"]
        ev(p, UA, "check", ("flag", p), chk)
        # slot 2 UA + 1: my own flags of h - 1 are complete: publish them (an OR into the shared word of h - 1), start afresh; the word of
        # h - 3 (everybody has looked at it during h - 1) is cleared for h + 1
        s = 2 * UA + 1
        ev(p, s, "valu", None, ["s_cmp_lg_u64 s[92:93], 0", "s_cselect_b32 s98, 1, 0", "s_mov_b64 s[92:93], 0", "v_mov_b32 %s, s98" % vflag] +
           flag_addr(-1) + ["v_mov_b32 %s, s97" % vtmp])
        ev(p, s, "lds", ("or", p), ["ds_or_b32 %s, %s" % (vtmp, vflag)])
        ev(p, s + 1, "valu", None, flag_addr(-3) + ["v_mov_b32 %s, s97" % vtmp])
        ev(p, s + 1, "lds", ("clr", p), ["ds_write_b32 %s, %s" % (vtmp, vzero)])
        # ct of THIS half-tile from its meta pair (pmax, nmax): the slack between the bf16 product and a bound of the exact head
        ev(p, s + 2, "check", ("meta", p), ["v_fma_f32 %s, %%[eu], %s, %s" % (ct(p), metan(p), metap(p))])
        # the pieces (and the meta entry) of h + PFD: their slot held h + PFD - 8, which everybody left long ago
        spread(p, s + 3, dma_issue(["s_add_u32 s80, %%[h], %d" % PFD]))

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
        P.append("v_and_b32 %s, 15, v%d" % (vtmp, ATMP0 + 1))                      # swz(r) = r & 15
    else:
        P += ["v_lshrrev_b32 %s, 1, v%d" % (vtmp, ATMP0 + 1), "v_and_b32 %s, 7, %s" % (vtmp, vtmp)]      # (r >> 1) & 7
    P.append("v_lshlrev_b32 v%d, %d, v%d" % (ATMP0 + 1, (2 * D).bit_length() - 1, ATMP0 + 1))            # r * 2D
    for k in range(NK):
        P += ["v_add_u32 %s, %d, v%d" % (sw(k), 2 * k, ATMP0), "v_xor_b32 %s, %s, %s" % (sw(k), sw(k), vtmp),
              "v_lshl_add_u32 %s, %s, 4, v%d" % (sw(k), sw(k), ATMP0 + 1)]
    # the wave's user fragments -> AGPRs, at EVERY entry: the compiler uses AGPRs as spill space between the statements
    P.append("s_mov_b64 s[88:89], %[ufrag]")
    for i in range(UA * NK):
        if i % 4 == 0 and i > 0:
            P += ["s_add_u32 s88, s88, 4096", "s_addc_u32 s89, s89, 0"]
        P.append("global_load_dwordx4 a[%d:%d], %%[lane16], s[88:89] offset:%d" % (4 * i, 4 * i + 3, 1024 * (i % 4)))
    P.append("s_waitcnt vmcnt(0)")
    # the shared flag words of h - 2 and h - 1 are dealt with outside: clear them (every wave; the barrier below orders it)
    for dlt in (-2, -1):
        P += flag_addr(dlt) + ["v_mov_b32 %s, s97" % vtmp, "ds_write_b32 %s, %s" % (vtmp, vzero)]
    # catch up with the loads: the pieces of half-tiles issued .. h + PFD - 1 (first entry: all of them; their slots are free)
    P += ["5:", "s_add_u32 s97, %%[h], %d" % PFD, "s_cmp_ge_u32 %[issued], s97", "s_cbranch_scc1 6f"]
    P += flat(dma_issue(["s_mov_b32 s80, %[issued]"])) + ["s_branch 5b", "6:"]
    # everything issued has landed; behind the barrier everybody's has
    P += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
    # the meta pair of h into the pair of its parity
    P += ["s_bitcmp1_b32 %[h], 0", "s_cbranch_scc1 7f"] + meta_read("%[h]", 0) + ["s_branch 8f", "7:"] + meta_read("%[h]", 1) + ["8:"]
    P += slot_addr("s94", "%[h]") + ["s_add_u32 s97, %[h], 1"] + slot_addr("s95", "s97")
    P += ["s_mov_b64 s[92:93], 0"]
    P += ["s_bitcmp1_b32 %[h], 0", "s_cbranch_scc1 9f"]
    for par in range(2):
        # every fragment the body reads AHEAD of the half-tile it belongs to (k < PF: during the last PF steps of the previous half-tile)
        for k in range(PF):
            P += ["v_add_u32 v%d, s94, %s" % (ATMP0, sw(k)), "ds_read_b128 %s, v%d" % (frag(k), ATMP0)]
        # (the first slots of the entry half-tile carry the tail of the PREVIOUS half-tile's tests: -inf makes them fail)
        P += ["v_mov_b32 %s, 0xff800000" % ct(1 - par), "v_mov_b32 %s, 0xff800000" % metap(1 - par), "s_waitcnt lgkmcnt(0)", "s_branch 2%df" % par]
        if par == 0:
            P.append("9:")
    # ---- exits ------------------------------------------------------------------------------------------------------------------
    drain = ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15", "s_mov_b32 m0, %[m0save]"]
    E = []
    E += ["91:", "s_mov_b32 %[reason], 1"] + drain + ["s_branch 99f"]          # half-tile h - 2 (or h - 1, not looked at yet) raised a flag in some wave
    E += ["92:", "s_mov_b32 %[reason], 0"] + drain + ["99:"]                   # the sweep is over
    return P + b2 + E


def emit(D):
    L = gen(D)
    out = []
    out.append("template <>")
    out.append("struct Loop5<%d> {" % D)
    out.append("    // h: the local half-tile to run next (in: where to (re)start; out: the half-tile in progress when the statement left).")
    out.append("    // issued: half-tiles whose pieces this wave has issued.  reason: 0 = the sweep is over, 1 = half-tile h - 2 raised a flag in some")
    out.append("    // wave of the workgroup (all four leave together; h - 1 has not been looked at).")
    out.append("    static __device__ __forceinline__ void run(unsigned& h, unsigned& issued, unsigned& reason, unsigned hend, unsigned ring, unsigned flags, unsigned w1024,")
    out.append("                                               unsigned t0, unsigned nsplit, unsigned imglo, unsigned imghi, unsigned metalo, unsigned metahi, unsigned metalds, float eu, const void* ufrag,")
    out.append("                                               const float (&thr)[8], unsigned lane16) {")
    out.append("#if defined(__HIP_DEVICE_COMPILE__)")
    out.append("        unsigned m0save;")
    out.append("        asm volatile(")
    for l in L:
        out.append('            "%s\\n\\t"' % l)
    out.append('            : [h] "+s"(h), [issued] "+s"(issued), [reason] "=&s"(reason), [m0save] "=&s"(m0save)')
    ins = ['[hend] "s"(hend)', '[ring] "s"(ring)', '[flags] "s"(flags)', '[w1024] "s"(w1024)', '[t0] "s"(t0)', '[nsplit] "s"(nsplit)',
           '[imglo] "s"(imglo)', '[imghi] "s"(imghi)', '[metalo] "s"(metalo)', '[metahi] "s"(metahi)', '[metalds] "s"(metalds)', '[eu] "s"(eu)', '[ufrag] "s"(ufrag)', '[lane16] "v"(lane16)']
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
