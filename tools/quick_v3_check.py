import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from pda_amd import ops
dev = torch.device('cuda')
g = torch.Generator(device=dev); g.manual_seed(1)
U = torch.randn(300, 128, device=dev, generator=g) * 0.1
I = torch.randn(3000, 128, device=dev, generator=g) * 0.1
users = torch.arange(300, dtype=torch.int32, device=dev)
t0 = time.time()
a = ops.topk_merge(ops.score_topk_keys(U, I, users, 50, 0, None, None, impl="v1"), want="keys")
b = ops.topk_merge(ops.score_topk_keys(U, I, users, 50, 0, None, None, prune=(sys.argv[1] == "ord")), want="keys")
torch.cuda.synchronize()
print("natural v3 == v1:", torch.equal(a, b), "%.2f s" % (time.time() - t0))
