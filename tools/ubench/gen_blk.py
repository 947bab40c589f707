# emits asm blocks for the ubench: UA=1 (18 MFMAs) and UA=2 (36 MFMAs) at D=128, F fragments in flight
def gen(D, UA, F):
    NM = D // 16; RB = 2*D+48; HB = 32*RB; S = 2*NM
    loads = [("frag", (s % 2)*HB + 32*(s//2)) for s in range(S)] + [("frag", cb*HB + 2*D) for cb in range(2)] + [("pi", HB//512)]
    n_steps = S + 2; n_loads = len(loads); lines = []
    def issue(i):
        kind, off = loads[i]
        lines.append(("ds_read_b128 %%[t%d], %%[addr] offset:%d" % (i % F, off)) if kind == "frag" else ("ds_read2st64_b64 %%[pi], %%[addrpi] offset1:%d" % off))
    for i in range(min(F, n_loads)): issue(i)
    for s in range(n_steps):
        issued = min(s + F, n_loads)
        lines.append("s_waitcnt lgkmcnt(%d)" % (issued - (s + 1)))
        m, cb = (s//2, s % 2) if s < S else (NM, s - S)
        for u in range(UA):
            a = ("%%[a%d_%d]" % (u, m)) if s < S else ("%%[aex%d]" % u)
            acc = "%%[acc%d_%d]" % (u, cb)
            lines.append("v_mfma_f32_32x32x16_bf16 %s, %s, %%[t%d], %s" % (acc, a, s % F, "0" if m == 0 else acc))
        if s + F < n_loads: issue(s + F)
    lines.append("s_waitcnt lgkmcnt(0)")
    outs = ['[acc%d_%d] "=&v"(acc[%d][%d])' % (u, cb, u, cb) for u in range(UA) for cb in range(2)] + ['[pi] "=&v"(pi)'] + ['[t%d] "=&v"(t%d)' % (i, i) for i in range(F)]
    ins = ['[a%d_%d] "v"(a[%d][%d])' % (u, m, u, m) for u in range(UA) for m in range(NM)] + ['[aex%d] "v"(aex[%d])' % (u, u) for u in range(UA)] + ['[addr] "v"(addr)', '[addrpi] "v"(addr_pi)']
    body = "\n".join('        "%s\\n\\t"' % l for l in lines)
    return ("template <> struct Blk<%d> {\n  static __device__ __forceinline__ void run(f32x16 (&acc)[%d][2], u32x4& pi, const u32x4 (&a)[%d][%d], const u32x4 (&aex)[%d], unsigned addr, unsigned addr_pi) {\n"
            "    u32x4 %s;\n    asm volatile(\n%s\n        : %s\n        : %s\n        : \"memory\");\n  }\n};\n") % (UA, UA, UA, NM, UA, ", ".join("t%d" % i for i in range(F)), body, ", ".join(outs), ", ".join(ins))
import sys
F = int(sys.argv[1]) if len(sys.argv) > 1 else 6
print("template <int UA> struct Blk;")
print(gen(128, 1, F)); print(gen(128, 2, F))
