"""scripts/inflight_trace.py on a synthetic rocprofv3 kernel trace: two queues, known overlap -> the concurrency shares it must report."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_concurrency_shares_of_a_known_trace(tmp_path):
    rows, t0 = [], 10 ** 9
    gemm = "void (anonymous namespace)::conv_gemm_kernel<1, 5, 0, true, 0, 2>((anonymous namespace)::ConvArgs)"
    attn = "void (anonymous namespace)::attention_kernel<64, 4>((anonymous namespace)::AttnArgs)"
    clip = "void (anonymous namespace)::pitch_prefix_kernel((anonymous namespace)::PitchArgs)"
    # queue 0: a GEMM for the first 60 us of every 100 us; queue 1: an attention launch for the first 30 us of every 100 us, from 20 us on
    for i in range(4000):
        b = t0 + i * 100_000
        rows.append((0, gemm, b, b + 60_000))
        rows.append((1, attn if i % 10 else clip, b + 20_000, b + 50_000))
    d = tmp_path / "trace"
    d.mkdir()
    with open(d / "x_kernel_trace.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Queue_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Workgroup_Size_X", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"])
        for q, n, s, e in rows:
            w.writerow(["KERNEL_DISPATCH", q, n, s, e, 256, 256 * 512, 1, 1])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "inflight_trace.py"), str(d), "100", "10"], check=True, capture_output=True, text=True).stdout
    lines = {l.split(":")[0].strip(): l for l in out.splitlines() if ":" in l}
    shares = dict(kv.split(": ") for kv in lines["kernels executing at once"].split(": ", 1)[1].split("  ")[:3])
    # per 100 us: 40 us idle, 30 us one kernel, 30 us two kernels
    assert abs(float(shares["0"]) - 0.40) < 0.01 and abs(float(shares["1"]) - 0.30) < 0.01 and abs(float(shares["2"]) - 0.30) < 0.01
    assert "NO implicit GEMM among them: 0.000" in out
    assert "mean kernels resident): 0.90" in out
    assert "conv_gemm_kernel<1,5,0,true,0,2>,512" in out
