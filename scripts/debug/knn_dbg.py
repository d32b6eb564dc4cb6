import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    sys.path.insert(0, p)
import torch
from svcmi import Ops
ops = Ops()
torch.manual_seed(0)
for (t, n, d) in ((300, 4097, 256), (2510, 50001, 256), (2510, 50000, 1280), (2510, 8192, 256), (512, 50000, 256)):
    c = torch.randn(8, d) * 2
    bank = (c[torch.randint(0, 8, (n,))] + torch.randn(n, d)).cuda()
    x = (c[torch.randint(0, 8, (t,))] + torch.randn(t, d)).cuda()
    ldd = (n + 3) // 4 * 4
    dots = torch.zeros(t, ldd, device="cuda")
    ops.conv(x[None], bank, out=dots[None], n_out=n)
    ref = x.double() @ bank.double().t()
    err = (dots[:, :n].double() - ref).abs()
    print(f"t={t} n={n} d={d}: dots max err {err.max().item():.3e}; bad rows {(err.max(1).values > 1e-2).sum().item()} bad cols {(err.max(0).values > 1e-2).sum().item()}", flush=True)
    if err.max() > 1e-2:
        bad = (err > 1e-2).nonzero()
        print("  first bad", bad[:5].tolist(), "last bad", bad[-5:].tolist())
    bsq = ops.row_sqnorm(bank)
    print("  sqnorm err", (bsq.double() - (bank.double() ** 2).sum(1)).abs().max().item())
    out = ops.knn_blend(x, bank, dots, bsq, 3, 0.5)
    d2 = (x.double() ** 2).sum(1)[:, None] + (bank.double() ** 2).sum(1)[None] - 2 * ref
    sc, ids = torch.topk(d2, 3, dim=1, largest=False)
    nb = bank[ids]                      # [t, 3, d]
    ex = ((x[:, None].double() - nb.double()) ** 2).sum(-1).float()
    w = (1 / ex) ** 2
    w = w / w.sum(1, keepdim=True)
    want = 0.5 * x + 0.5 * (nb * w[:, :, None]).sum(1)
    e2 = (out - want).abs().max(1).values
    print(f"  blend max err {e2.max().item():.3e}; bad rows {(e2 > 1e-3).sum().item()} of {t}; first bad rows {(e2 > 1e-3).nonzero()[:8].flatten().tolist()}", flush=True)
import time
from svcmi.feature_retrieval import KnnFeatureIndex
for (t, n, d) in ((2510, 200000, 1280), (2510, 200000, 256)):
    idx = KnnFeatureIndex(torch.randn(n, d), 0.5, 3, ops=ops)
    x = torch.randn(t, d, device="cuda")
    for _ in range(2):
        idx.retriv(x)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5):
        idx.retriv(x)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print(f"retriv t={t} n={n} d={d}: {dt * 1e3:.2f} ms ({2.0 * t * n * d / dt / 1e12:.1f} TFLOP/s on the score GEMM)")
