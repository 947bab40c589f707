#!/usr/bin/env python
"""Generates tools/ubench/loop5.h: the WHOLE main loop of an MFMA wave as one inline-asm statement, for the structural
micro-benchmark of round 4 (tools/ubench/mfma_struct5.hip) -- which machine mapping of the sweep can keep the matrix pipe fed when

  * the product is TRANSPOSED (A = item fragment from the LDS, B = user fragment in registers): an accumulator lane then holds 16
    items of ONE user, so the threshold test is a per-lane compare (7 v_max3 + v_max + v_add + v_cmp per 32 x 32 block) and the
    folded test k-step (1/9 of all MFMAs at d = 128, 1/5 at d = 64) disappears;
  * a wave carries UA user fragments per item fragment (UA MFMAs per ds_read_b128): UA = 2 (today's wide geometry), 4, 8.

Every register is a hard register (the statement clobbers them); the stream is placed by hand: one MFMA per slot with its fillers
behind it (MI355X_MICROARCH.md: <= 5 single-issue fillers hide in a 32-cycle MFMA gap).

    python tools/ubench/gen_loop5.py > tools/ubench/loop5.h
"""


class Cfg:
    def __init__(self, name, UA, NK, G, R, users_in_agpr, acc0, usr0, frag0, tmp0, lo_clobber, hi_clobber, self_pieces, MW, pfd=2):
        self.name, self.UA, self.NK, self.G, self.R = name, UA, NK, G, R
        self.users_in_agpr = users_in_agpr
        self.acc0, self.usr0, self.frag0, self.tmp0 = acc0, usr0, frag0, tmp0
        self.lo_clobber, self.hi_clobber = lo_clobber, hi_clobber
        self.self_pieces = self_pieces          # LDS-DMA pieces this wave issues per half-tile (0: loader waves do it)
        self.MW = MW                            # MFMA waves per workgroup (piece p of a half-tile belongs to wave p % MW)
        self.pfd = pfd                          # DMA prefetch distance in half-tiles
        self.lag = NK // G
        self.PF = max(1, min(3, R - (G - 1) * self.lag - 1))      # steps between a fragment's read and its first MFMA
        assert (2 * NK) % R == 0 and (G - 1) * self.lag + self.PF < R


def gen(c):
    UA, NK, G, lag, PF, R = c.UA, c.NK, c.G, c.lag, c.PF, c.R
    D = 16 * NK
    RB = 2 * D + 16
    HB = 32 * RB
    per_g = UA // G
    acc = lambda u: "v[%d:%d]" % (c.acc0 + 16 * u, c.acc0 + 16 * u + 15)
    accr = lambda u, r: "v%d" % (c.acc0 + 16 * u + r)
    usr = lambda u, k: ("a[%d:%d]" if c.users_in_agpr else "v[%d:%d]") % (c.usr0 + 4 * (u * NK + k), c.usr0 + 4 * (u * NK + k) + 3)
    frag = lambda k: "v[%d:%d]" % (c.frag0 + 4 * (k % R), c.frag0 + 4 * (k % R) + 3)
    T = c.tmp0
    thr = lambda u: "v%d" % (T + u)
    mt = lambda u: "v%d" % (T + UA + u)
    X = T + 2 * UA
    cpair = lambda p: "v[%d:%d]" % (X + 2 * p, X + 1 + 2 * p)
    cval = lambda p: "v%d" % (X + 2 * p)
    base_cur, base_nxt, caddr = "v%d" % (X + 4), "v%d" % (X + 5), "v%d" % (X + 6)
    poll = "v[%d:%d]" % (X + 8, X + 11)
    one = "v%d" % (X + 7)
    voff = lambda j: "v%d" % (X + 12 + j)
    n_tmp = 2 * UA + 12 + max(1, c.self_pieces)
    assert T + n_tmp <= min(c.frag0, c.acc0, c.usr0 if not c.users_in_agpr else 10 ** 9), (c.name, T + n_tmp)
    GS = "s[90:91]"
    n_slots = 2 * NK * UA
    slot_of = lambda hh, k, pos: ((hh * NK + k) * UA + pos) % n_slots

    # ---------------- the body's instruction placement ----------------
    mf = []
    for hh in range(2):
        for k in range(NK):
            pos = 0
            for g in range(G):
                kg = (k - g * lag) % NK
                for uu in range(per_g):
                    mf.append((hh, k, pos, g * per_g + uu, kg))
                    pos += 1
    fill = [[] for _ in range(n_slots)]          # (kind, tag, [lines])
    for hh in range(2):
        for k in range(NK):
            step = (hh * NK + k - PF) % (2 * NK)       # fragment (hh, k): first used by group 0 at step (hh, k), read PF steps earlier
            read_hh = step // NK
            b = base_cur if read_hh == hh else base_nxt
            fill[step * UA + min(1, UA - 1)].append(("lds", ("frag", hh, k), ["ds_read_b128 %s, %s offset:%d" % (frag(k), b, 32 * k)]))
    for hh in range(2):
        s0 = slot_of(hh, 0, 0)
        # half-tile hh begins: cur <- nxt, nxt <- nxt + HB (wrapping inside the tile ring); everything read for the NEXT half-tile uses nxt
        fill[s0].insert(0, ("valu", None, ["v_mov_b32 %s, %s" % (base_cur, base_nxt), "v_add_u32 %s, %d, %s" % (base_nxt, HB, base_nxt)]))
        fill[(s0 + 1) % n_slots].append(("valu", None, ["v_cmp_ge_u32 vcc, %s, %%[tend]" % base_nxt,
                                                         "v_cndmask_b32 %s, %s, %%[tbase], vcc" % (base_nxt, base_nxt),
                                                         "v_sub_u32 %s, %s, %%[hoff]" % (caddr, base_nxt)]))
        # the tile-uniform slack c_t of the NEXT half-tile (its parity 1 - hh): a broadcast read of the tile's pad
        fill[(s0 + 2) % n_slots].append(("lds", ("c", 1 - hh), ["ds_read_b64 %s, %s offset:%d" % (cpair(1 - hh), caddr, 2 * D)]))
        # hand-over words: poll "landed", publish "released"
        fill[(s0 + 3) % n_slots].append(("lds", ("poll", hh), ["ds_read_b128 %s, %%[sync]" % poll]))
        fill[slot_of(hh, NK - 1, UA - 1)].append(("lds", ("rel", hh), ["ds_write_b32 %%[sync], %s offset:64" % one]))
        if c.self_pieces:
            for j in range(c.self_pieces):
                sl = slot_of(hh, 1 + j, min(2, UA - 1))
                fill[sl].append(("vmem", None, ["s_add_u32 m0, %%[dmadst], %d" % (j * c.MW * 1024), "s_nop 0",
                                                "global_load_lds_dwordx4 %s, %s" % (voff(j), GS)]))
            sl = slot_of(hh, 1 + c.self_pieces, min(2, UA - 1))
            fill[sl].append(("salu", None, ["s_add_u32 s90, s90, %d" % HB, "s_addc_u32 s91, s91, 0",
                                            "s_waitcnt vmcnt(%d)" % (c.self_pieces * (c.pfd - 1))]))
            fill[(sl + 1) % n_slots].append(("lds", ("lan", hh), ["ds_write_b32 %%[sync], %s offset:128" % one]))
    # the threshold tests: chain u's accumulator of a half-tile is final behind its MFMA at kg = NK - 1 and restarts UA slots later
    for (hh, k, pos, u, kg) in mf:
        if kg != NK - 1:
            continue
        s = (hh * NK + k) * UA + pos
        g = u // per_g
        par = hh if k >= g * lag else 1 - hh            # the half-tile this chain has just finished
        m = mt(u)
        ops = [["v_max3_f32 %s, %s, %s, %s" % (m, accr(u, 0), accr(u, 1), accr(u, 2))]]
        for r in range(3, 15, 2):
            ops.append(["v_max3_f32 %s, %s, %s, %s" % (m, m, accr(u, r), accr(u, r + 1))])
        ops.append(["v_max_f32 %s, %s, %s" % (m, m, accr(u, 15))])
        ops.append(["v_add_f32 %s, %s, %s" % (m, m, cval(par))])
        ops.append(["v_cmp_gt_f32 vcc, %s, %s" % (m, thr(u)), "s_or_b64 %[flag], %[flag], vcc"])
        if UA >= 4:
            first, last = s + 2, s + UA - 1             # >= 2 MFMAs behind the final one (XDL write -> VALU read), before the restart
            nwin = last - first + 1
            for i, o in enumerate(ops):
                tag = ("testc", par) if "v_add_f32" in o[0] else None
                fill[(first + (i * nwin) // len(ops)) % n_slots].append(("valu", tag, o))
        else:
            # two chains per wave: no room between an accumulator's last MFMA and its restart -- the wave drains (two such waves share
            # a SIMD and cover for each other), like today's block statement
            sl = (hh * NK + k) * UA + UA - 1
            if pos == 0:
                fill[sl].append(("valu", None, ["s_nop 15", "s_nop 3"]))
            for o in ops:
                fill[sl].append(("valu", ("testc", par) if "v_add_f32" in o[0] else None, o))

    def body(state):
        L, lg = [], list(state)

        def wait_for(tag):
            if tag in lg:
                pos = len(lg) - 1 - lg[::-1].index(tag)
                L.append("s_waitcnt lgkmcnt(%d)" % min(15, len(lg) - 1 - pos))
                del lg[:pos + 1]
        for (hh, k, pos, u, kg) in mf:
            g = u // per_g
            ftile = hh if k >= g * lag else 1 - hh
            wait_for(("frag", ftile, kg))
            L.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc(u), frag(kg), usr(u, kg), "0" if kg == 0 else acc(u)))
            for kind, tag, lines in fill[(hh * NK + k) * UA + pos]:
                if kind == "lds":
                    L.extend(lines)
                    lg.append(tag)
                else:
                    if tag is not None:
                        wait_for(("c", tag[1]))
                    L.extend(lines)
        return L, lg

    # steady state: the waits of a body are counted against what the PREVIOUS body left in flight
    _, st1 = body([])
    L2, st2 = body(st1)
    L3, st3 = body(st2)
    assert st2 == st3 and L2 == L3, c.name

    P = []
    P.append("s_mov_b32 %[m0save], m0")
    P.append("s_mov_b64 %s, %%[gsrc]" % GS)
    P.append("v_mov_b32 %s, %%[tbase]" % base_nxt)
    P.append("v_sub_u32 %s, %s, %%[hoff]" % (caddr, base_nxt))
    P.append("v_mov_b32 %s, 1" % one)
    for u in range(UA):
        P.append("v_mov_b32 %s, %%[thr0]" % thr(u))
    for j in range(c.self_pieces):
        P.append("v_add_u32 %s, %d, %%[goff]" % (voff(j), j * c.MW * 1024))
    n = 0
    for u in range(UA):
        for k in range(NK):
            P.append("ds_read_b128 %s, %%[ubase] offset:%d" % (usr(u, k), 1024 * (u * NK + k)))
            n += 1
            if n % 12 == 0:
                P.append("s_waitcnt lgkmcnt(0)")
    P.append("s_waitcnt lgkmcnt(0)")
    for u in range(UA):
        P.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, 0" % (acc(u), usr(u, 0), usr(u, 0)))      # defined accumulators for the lagging groups' first pass
    # everything the steady-state body expects to have been ISSUED on entry (then drained: the body's counted waits are merely conservative
    # in the first pass)
    for tag in st2:
        if tag[0] == "frag":
            P.append("ds_read_b128 %s, %s offset:%d" % (frag(tag[2]), base_nxt, 32 * tag[2]))
        elif tag[0] == "c":
            P.append("ds_read_b64 %s, %s offset:%d" % (cpair(tag[1]), caddr, 2 * D))
    P.append("ds_read_b64 %s, %s offset:%d" % (cpair(0), caddr, 2 * D))
    P.append("ds_read_b64 %s, %s offset:%d" % (cpair(1), caddr, 2 * D))
    P.append("s_waitcnt lgkmcnt(0)")
    P.append("s_mov_b64 %[flag], 0")
    P.append("1:")
    E = ["s_sub_u32 %[ntile], %[ntile], 1", "s_cmp_lg_u32 %[ntile], 0", "s_cbranch_scc1 1b", "s_waitcnt lgkmcnt(0)", "s_waitcnt vmcnt(0)",
         "s_nop 15", "s_nop 3", "s_mov_b32 m0, %[m0save]"]
    for u in range(UA):
        E.append("v_max_f32 %%[sink], %%[sink], %s" % accr(u, 0))
    n_mfma = len(mf)
    return P + L2 + E, n_mfma


def emit_struct(c):
    L, n_mfma = gen(c)
    out = []
    out.append("struct %s {" % c.name)
    out.append("    static constexpr int UA = %d, NK = %d, SELF = %d, HB = %d, RB = %d, MFMA_PER_BODY = %d, AGPR = %d;" %
               (c.UA, c.NK, c.self_pieces, 32 * (32 * c.NK + 16), 32 * c.NK + 16, n_mfma, 1 if c.users_in_agpr else 0))
    out.append("    // one body = two 32-item half-tiles against 32 UA users; n_body bodies")
    out.append("    static __device__ __forceinline__ void run(float& sink, unsigned long long& flag, unsigned ubase, unsigned tbase, unsigned tend, unsigned hoff, unsigned sync,")
    out.append("                                               float thr0, unsigned n_body, const void* gsrc, unsigned goff, unsigned dmadst) {")
    out.append("#if defined(__HIP_DEVICE_COMPILE__)")
    out.append("        unsigned m0save;")
    out.append("        asm volatile(")
    for l in L:
        out.append('            "%s\\n\\t"' % l)
    out.append('            : [sink] "+v"(sink), [flag] "=&s"(flag), [ntile] "+s"(n_body), [m0save] "=&s"(m0save)')
    out.append('            : [ubase] "v"(ubase), [tbase] "v"(tbase), [tend] "v"(tend), [hoff] "v"(hoff), [sync] "v"(sync), [thr0] "v"(thr0), [goff] "v"(goff), [dmadst] "s"(dmadst), [gsrc] "s"(gsrc)')
    clob = ['"memory"', '"vcc"', '"scc"', '"s90"', '"s91"'] + ['"v%d"' % r for r in range(c.lo_clobber, c.hi_clobber + 1)]
    if c.users_in_agpr:
        clob += ['"a%d"' % r for r in range(0, 4 * c.UA * c.NK)]
    out.append("            : " + ", ".join(clob) + ");")
    out.append("#endif")
    out.append("    }")
    out.append("};")
    return "\n".join(out)


CFGS = [
    # name, UA, NK, G, R, agpr users, acc0, usr0, frag0, tmp0, clobber range, self-load pieces, MFMA waves
    # V1: today's wide mapping without the test k-step: 8 MFMA waves x 64 users (168 VGPRs, 3 waves per SIMD), loader waves beside them
    Cfg("LoopV1", 2, 8, 1, 4, False, 136, 72, 56, 24, 24, 167, 0, 8),
    # V2: 4 MFMA waves x 128 users, one per SIMD, + 2 loader + 2 idle waves (8 waves at 256 VGPRs)
    Cfg("LoopV2", 4, 8, 1, 4, False, 192, 64, 48, 16, 16, 255, 0, 4),
    # V3: 8 MFMA waves x 128 users, two per SIMD (256 VGPRs), loading their own tiles
    Cfg("LoopV3", 4, 8, 1, 4, False, 192, 64, 48, 16, 16, 255, 1, 8),
    # V4: 4 MFMA waves x 256 users, one per SIMD (512 registers: the user fragments in AGPRs), loading their own tiles; two skew groups
    Cfg("LoopV4", 8, 8, 2, 8, True, 128, 0, 96, 64, 64, 255, 2, 4),
    Cfg("LoopV4g1", 8, 8, 1, 8, True, 128, 0, 96, 64, 64, 255, 2, 4),
    # d = 64 (C1 / C2): V4 with four k-steps
    Cfg("LoopV4d64", 8, 4, 2, 8, True, 128, 0, 96, 64, 64, 255, 1, 4),
]


def main():
    print("// GENERATED by tools/ubench/gen_loop5.py -- do not edit.")
    for c in CFGS:
        print(emit_struct(c))


if __name__ == "__main__":
    main()
