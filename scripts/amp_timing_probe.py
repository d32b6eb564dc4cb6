"""Throw-away phase timing of the fused half-step kernel (narrow generator stages): needs a library built with -DSVCMI_AMP_TIMING
(scripts/build_variant.sh timing -DSVCMI_AMP_TIMING on an instrumented csrc/amp_fused.hip; SVCMI_LIB points at it).  Per wave: cycles in the
SnakeAlias phase, at the barrier behind it, in the convolution, in the epilogue; per SnakeAlias work item: cycles until its 18 x loads have
arrived, cycles of arithmetic."""
import ctypes
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from svcmi import Ops  # noqa: E402
from svcmi import weights as PW  # noqa: E402
from workload import weights as W  # noqa: E402


def main():
    ops = Ops()
    lib = ops.lib
    lib.svcmi_amp_timing_read.restype = ctypes.c_int
    lib.svcmi_amp_timing_read.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    NS = 8192
    buf = (ctypes.c_uint64 * (NS * 8))()
    filt = W.kaiser_sinc_filter().view(-1).cuda()
    import numpy as np
    for (c, ld, L) in ((20, 20, 160000), (10, 12, 320000)):
        for B in (1, 4):
            for amp_u in (-1, 1):
                g = torch.Generator().manual_seed(c)
                x = torch.zeros(B, L, ld)
                x[..., :c] = torch.randn(B, L, c, generator=g)
                x = x.cuda()
                probs = []
                for k, d in ((3, 1), (7, 3), (11, 5)):
                    w = PW.pack_conv(torch.randn(c, c, k, generator=g) / math.sqrt(c * k), ld, ld).cuda()
                    al, be = torch.zeros(ld), torch.zeros(ld)
                    al[:c], be[:c] = torch.randn(c, generator=g) * 0.3, torch.randn(c, generator=g) * 0.3
                    probs.append(dict(x=x, alpha_log=al.cuda(), beta_log=be.cuda(), w=w, bias=PW.pad_vec(torch.randn(c, generator=g), ld).cuda(), ksize=k,
                                      dilation=d, res=x, out=torch.empty_like(x)))
                lib.svcmi_tune_set(b"amp_u", amp_u)
                for _ in range(3):
                    ops.snake_conv_group(probs, filt, c=c)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.snake_conv_group(probs, filt, c=c)
                e1.record()
                torch.cuda.synchronize()
                lib.svcmi_amp_timing_read(buf, NS * 8)
                a = np.frombuffer(buf, dtype=np.uint64).reshape(NS, 8).astype(np.float64)
                a = a[a[:, 5] > 0]
                ph = a[:, :4]
                span = (a[:, 5].max() - a[:, 4].min())
                tot = ph.sum(1)
                print(f"c={c} B={B} amp_u={amp_u}: launch {e0.elapsed_time(e1) * 1e3:7.1f} us, {len(a)} blocks sampled | median s_memtime ticks (100 MHz?) per block (wave 0): snake {np.median(ph[:, 0]):7.0f}  barrier {np.median(ph[:, 1]):7.0f}  "
                      f"conv {np.median(ph[:, 2]):7.0f}  epilogue {np.median(ph[:, 3]):7.0f}  total {np.median(tot):7.0f} | shares {ph[:, 0].sum() / tot.sum():.2f} {ph[:, 1].sum() / tot.sum():.2f} {ph[:, 2].sum() / tot.sum():.2f} {ph[:, 3].sum() / tot.sum():.2f} | first start -> last end {span:9.0f} ticks", flush=True)
    lib.svcmi_tune_set(b"amp_u", 0)


if __name__ == "__main__":
    main()
