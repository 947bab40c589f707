import sys, torch
sys.path.insert(0, '.')
from pda_amd import ops
dev = torch.device('cuda')
g = torch.Generator(device=dev); g.manual_seed(5)
U = torch.randn(200_000, 256, device=dev, generator=g) * 0.07
I = torch.randn(100_000, 256, device=dev, generator=g) * 0.07
pop = (torch.rand(100_000, device=dev, generator=g) ** 4).contiguous()
users = torch.arange(0, 32768, dtype=torch.int32, device=dev)
def t(impl, prune):
    k = ops.score_topk_keys(U, I, users, 50, 1, pop, None, impl=impl, prune=prune); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): k = ops.score_topk_keys(U, I, users, 50, 1, pop, None, impl=impl, prune=prune)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3, ops.topk_merge(k, want="keys")
a, ka = t("v1", False)
b, kb = t("v2", "order")
print("d=256 f32: v1 %.2f ms   prepped (env kernel) dense ordered %.2f ms   same=%s" % (a, b, torch.equal(ka, kb)))
