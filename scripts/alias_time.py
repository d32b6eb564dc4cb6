import sys, os, torch, time
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "whisper-vits-svc_amd"))
from svcmi import Ops
from workload import weights as W
ops = Ops(); filt = W.kaiser_sinc_filter().view(-1).to("cuda")
g = torch.Generator().manual_seed(1)
for cp, n in ((40, 24000), (80, 12000), (160, 6000)):
    xs = [torch.randn(1, n, cp, generator=g).to("cuda") for _ in range(3)]
    al = [(torch.randn(cp, generator=g) * 0.3).to("cuda") for _ in range(3)]; be = [(torch.randn(cp, generator=g) * 0.3).to("cuda") for _ in range(3)]
    outs = [torch.empty_like(x) for x in xs]
    for _ in range(20): ops.snake_alias_group(xs, al, be, filt, outs)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): ops.snake_alias_group(xs, al, be, filt, outs)
    e1.record(); torch.cuda.synchronize()
    print(f"alias group {cp} ch n={n}: {e0.elapsed_time(e1) / 200 * 1000:.2f} us per launch", flush=True)
