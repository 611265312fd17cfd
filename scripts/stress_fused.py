"""Stress the fused fix-up: many launches with randomly placed degenerate groups; fused result must equal the separate-launch result."""
import os, sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from polars_ols_amd import Engine
eng = Engine(0)
rng = np.random.default_rng(0)
G, n, k = 4000, 1000, 8
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    cols = [torch.randn(G * n, device="cuda") for _ in range(k)]
    y = sum(cols) + 0.1 * torch.randn(G * n, device="cuda")
    flagged = rng.choice(G, size=int(rng.integers(0, 40)), replace=False)
    for g in flagged:                                     # two identical columns -> rank deficient -> flagged
        cols[5][g * n:(g + 1) * n] = cols[2][g * n:(g + 1) * n]
    offs = np.arange(G + 1, dtype=np.int64) * n
    eng.set_option("FUSED_FIXUP", "1")
    out = eng.least_squares(y, cols, offs, want=("coef", "pred", "status"))
    kern = eng.last_kernel
    torch.cuda.synchronize()
    eng.set_option("FUSED_FIXUP", None)
    ref = eng.least_squares(y, cols, offs, want=("coef", "pred", "status"))
    torch.cuda.synchronize()
    st = out["status"].cpu().numpy()
    ok = (np.array_equal(st, ref["status"].cpu().numpy()) and set(np.nonzero(st == 1)[0]) == set(flagged.tolist())
          and torch.equal(out["coef"], ref["coef"]) and torch.equal(out["pred"], ref["pred"]))
    if not ok:
        bad += 1
        print("MISMATCH it", it, "flagged", len(flagged), "status1", int((st == 1).sum()),
              "coef maxdiff", float((out["coef"] - ref["coef"]).abs().max()), "pred maxdiff", float((out["pred"] - ref["pred"]).abs().nan_to_num().max()))
print("stress done, kernel", kern, "bad =", bad)
