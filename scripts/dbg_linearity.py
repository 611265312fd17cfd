import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from polars_ols_amd import Engine
eng = Engine(0)
groups, rows, k = 10_000, 1_000, 8
g = torch.Generator(device="cuda").manual_seed(0)
N = groups * rows
cols = [torch.randn(N, generator=g, device="cuda", dtype=torch.float32) for _ in range(k)]
y = sum(cols) + 0.1 * torch.randn(N, generator=g, device="cuda", dtype=torch.float32)
offs = np.arange(groups + 1, dtype=np.int64) * rows
for engine in ("valu", "mfma"):
    eng.set_option("K1_ENGINE", engine)
    a = eng.least_squares(y, cols, offs, want=("coef", "pred"))
    b = eng.least_squares(y, cols, offs, want=("coef", "pred"))
    torch.cuda.synchronize()
    print(engine, eng.last_kernel, "same-input bitwise equal:", torch.equal(a["pred"], b["pred"]), torch.equal(a["coef"], b["coef"]))
    y2 = 2.0 * y
    c = eng.least_squares(y2, cols, offs, want=("coef", "pred"))
    torch.cuda.synchronize()
    d = (c["pred"] - 2 * a["pred"]).abs()
    dc = (c["coef"] - 2 * a["coef"]).abs()
    print("  2y: max|dpred|=%.3e at %d  max|dcoef|=%.3e  n_bad_groups=%d" % (d.max().item(), d.argmax().item(), dc.max().item(), (dc.max(1).values > 0).sum().item()))
    bad = (dc.max(1).values > 0).nonzero().flatten()[:8].tolist()
    print("  first differing groups:", bad)
