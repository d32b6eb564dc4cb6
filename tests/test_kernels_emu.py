"""Kernel logic under the CPU SIMT emulator (tests/emu): the SAME csrc/*.hip sources compiled with
g++ -DSVCMI_EMU, called through the C ABI, checked against torch / the oracle on small shapes.
This is what `-m "not gpu"` can verify about the HIP kernels without a GPU: tiling, indexing,
masking, epilogues.  The gpu-marked tests run the identical checks on the real library."""
import pytest
import torch

from tests import kernel_cases as K
from tests.emu import emu_ops


@pytest.fixture(scope="module")
def ops():
    return emu_ops()


@pytest.mark.parametrize("case", K.CONV_CASES_SMALL, ids=lambda c: c["id"])
def test_conv_gemm(ops, case):
    K.check_conv(ops, case, device="cpu")


@pytest.mark.parametrize("case", K.CONV_CASES_LP_SMALL, ids=lambda c: c["id"])
def test_conv_gemm_reduced_precision(ops, case):
    K.check_conv(ops, case, device="cpu")


@pytest.mark.parametrize("c", [32, 192, 1280])
def test_layernorm(ops, c):
    K.check_layernorm(ops, c, device="cpu")


@pytest.mark.parametrize("case", K.ATTN_CASES_SMALL + K.ATTN_CASES_Q32 + K.ATTN_CASES_LDS + K.ATTN_CASES_WIDE, ids=lambda c: c["id"])
def test_attention(ops, case):
    K.check_attention(ops, case, device="cpu")


@pytest.mark.parametrize("n,c", [(3, 4), (7, 12), (23, 4), (61, 12), (130, 20)])
def test_snake_alias(ops, n, c):
    K.check_snake(ops, n, c, device="cpu")


@pytest.mark.parametrize("n,c,amp,alpha", [(61, 12, 40.0, 2.5), (130, 20, 300.0, 3.0)])
def test_snake_alias_outlier_scale(ops, n, c, amp, alpha):
    K.check_snake(ops, n, c, device="cpu", amp=amp, alpha_mean=alpha)


@pytest.mark.parametrize("case", K.SNAKE_CONV_CASES, ids=lambda c: c["id"])
def test_snake_conv_fused(ops, case):
    K.check_snake_conv(ops, case, device="cpu")


@pytest.mark.parametrize("S,c", [(1, 128), (3, 1280), (8, 2048)])
def test_splitk_layernorm(ops, S, c):
    K.check_splitk_layernorm(ops, "cpu", S=S, c=c)


@pytest.mark.parametrize("case", K.UPNOISE_CASES, ids=lambda c: c["id"])
def test_upsample_noise_fused(ops, case):
    K.check_upsample_noise(ops, case, device="cpu")


@pytest.mark.parametrize("T,c", [(5, 8), (700, 32)] + ([(64000, 512)] if "cpu" == "cuda" else []))
def test_channel_norm_gelu(ops, T, c):
    K.check_channel_norm_gelu(ops, "cpu", T=T, c=c)


@pytest.mark.parametrize("t,n,d,k,ratio", [(9, 70, 32, 3, 0.5), (5, 300, 256, 1, 1.0), (3, 9, 16, 8, 0.25)])
def test_knn_blend(ops, t, n, d, k, ratio):
    K.check_knn_blend(ops, "cpu", t, n, d, k, ratio)


@pytest.mark.parametrize("t,n,d,k,ratio,nlist", [(9, 200, 16, 3, 0.5, 5), (6, 120, 32, 1, 1.0, 4), (5, 150, 16, 8, 0.25, 3)])
def test_ivf_index(ops, tmp_path, t, n, d, k, ratio, nlist):
    K.check_ivf_index(ops, "cpu", t, n, d, k, ratio, nlist, tmp_path=tmp_path)


@pytest.mark.parametrize("n,d,blobs,n_ivf,exact", [(120, 8, 6, None, True)])
def test_ivf_train(ops, n, d, blobs, n_ivf, exact):
    K.check_ivf_train(ops, "cpu", n, d, blobs, n_ivf, exact)


@pytest.mark.parametrize("n,c", [(150, 40), (70, 80), (90, 16), (40, 32)])
def test_grouped_launches(ops, n, c):
    K.check_grouped_launches(ops, "cpu", B=2, n=n, c=c, ld=c)


@pytest.mark.parametrize("n,c,prec", [(150, 40, "bf16x3"), (70, 80, "bf16"), (90, 16, "f16")])
def test_grouped_launches_reduced_precision(ops, n, c, prec):
    K.check_grouped_launches(ops, "cpu", B=2, n=n, c=c, ld=c, prec=prec)


@pytest.mark.parametrize("c,ld,n", [(10, 12, 150), (20, 20, 150), (20, 20, 900)])
def test_snake_conv_group(ops, c, ld, n):
    K.check_snake_conv_group(ops, "cpu", c=c, ld=ld, B=2 if n < 500 else 1, n=n)


@pytest.mark.parametrize("c,ld,n,precision", [(10, 12, 150, "f16"), (20, 20, 300, "f16w2"), (20, 20, 40, "f16"), (10, 12, 290, "f16w2"),
                                              (20, 20, 1100, "f16")])       # (1100 rows: tiles 1 and 2 of 5 take the interior fast path of the U tile)
def test_snake_conv_group_on_the_fp16_matrix_cores(ops, c, ld, n, precision):
    K.check_snake_conv_group_lp(ops, "cpu", c=c, ld=ld, B=2 if n < 200 else 1, n=n, precision=precision)


@pytest.mark.parametrize("n", [5, 700])
def test_snake_post(ops, n):
    K.check_snake_post(ops, "cpu", B=2, n=n)


@pytest.mark.parametrize("jumps", [False, True])
def test_viterbi_decode(ops, jumps):
    K.check_viterbi(ops, "cpu", frames=40, batch_frames=16, jumps=jumps)


def test_flow_glue(ops):
    K.check_flow_glue(ops, device="cpu")


def test_prior_glue(ops):
    K.check_prior_glue(ops, device="cpu")


def test_layout_bridges(ops):
    K.check_bridges(ops, device="cpu")


@pytest.mark.parametrize("T,B", [(5, 2), (1100, 1)])
def test_pitch2source(ops, T, B):
    K.check_pitch2source(ops, T, B, device="cpu", hop=16 if T > 100 else 320)


def test_source2wav(ops):
    K.check_source2wav(ops, device="cpu")


def test_argument_errors_are_reported(ops):
    x = torch.zeros(1, 8, 8)
    w = torch.zeros(4, 8)
    with pytest.raises(Exception):
        ops.conv(x, w, ksize=2)          # ldw < ksize*c_in
    with pytest.raises(Exception):
        ops.attention(torch.zeros(1, 4, 3 * 20), heads=1, scale=1.0)   # head_dim 20 unsupported


def test_svc_train_retrieval_cli_writes_loadable_indexes(ops, tmp_path, monkeypatch):
    """svc_train_retrieval.py of the reference end to end on the emulated kernels: data_svc/{hubert,whisper}/<spk>/*.npy ->
    data_svc/indexes/<spk>/{hubert,whisper}.index (faiss layout) -> load_retrieve_index -> retriv."""
    import numpy as np
    from svcmi import feature_retrieval as FR, ivf_index as IV, svc_train_retrieval as TR
    rng = np.random.default_rng(11)
    base = tmp_path / "data_svc"
    (base / "whisper" / "spk0").mkdir(parents=True)
    for j in range(2):
        np.save(base / "whisper" / "spk0" / f"{j}.npy", rng.standard_normal((40, 16)).astype(np.float32))
    (base / "indexes").mkdir()
    TR.create_index("whisper", "p_", "spk0", base, base / "indexes", 200_000, 10_000, 1, device="cpu", ops=ops)
    f = base / "indexes" / "spk0" / "p_whisper.index"
    raw = IV.read_faiss_ivf_flat(f)
    assert raw["ntotal"] == 80 and raw["nlist"] == IV.ivf_list_count(80) == 2 and raw["nprobe"] == 1
    with pytest.raises(FileExistsError):
        TR.create_index("whisper", "p_", "spk0", base, base / "indexes", 200_000, 10_000, 1, device="cpu", ops=ops)
    index = FR.load_retrieve_index(f, 0.5, 2, device="cpu", ops=ops)
    assert isinstance(index, IV.IvfFlatFeatureIndex)
    x = rng.standard_normal((5, 16)).astype(np.float32)
    y = index.retriv(x)
    assert y.shape == x.shape and np.isfinite(y).all() and not np.allclose(y, x)
    assert TR.build_parser().parse_args([]).compress_features_after == 200_000


def test_outputs16(ops):
    K.check_outputs16(ops, "cpu")


@pytest.mark.parametrize("case", K.ATTN16_CASES, ids=lambda c: c["id"])
def test_attention16(ops, case):
    K.check_attention16(ops, case, "cpu")


@pytest.mark.parametrize("order", ["reverse", "shuffle:5"])
def test_kernels_do_not_depend_on_the_thread_order_inside_a_barrier_interval(ops, order, monkeypatch):
    """The emulator runs the fibers of a block in thread order between barriers; SVCMI_EMU_ORDER makes it run them backwards / shuffled
    (tests/emu/hip_emu.cpp).  A kernel with a missing __syncthreads (a thread reading LDS another thread writes in the same interval)
    gives different results then: the LDS-heavy kernels -- the half-step kernels in all forms, GEMM, attention,
    split-K LayerNorm -- are checked against their references under both orders."""
    monkeypatch.setenv("SVCMI_EMU_ORDER", order)
    K.check_snake_conv_group_lp(ops, "cpu", c=20, ld=20, B=1, n=300, precision="f16w2")
    K.check_snake_conv_group_lp(ops, "cpu", c=10, ld=12, B=2, n=150, precision="f16")
    K.check_snake_conv_group(ops, "cpu", c=20, ld=20, B=1, n=300)
    K.check_snake_conv_group(ops, "cpu", c=10, ld=12, B=2, n=150)
    K.check_snake_post(ops, "cpu", B=1, n=300)
    K.check_conv(ops, K.CONV_CASES_SMALL[1], "cpu")
    K.check_conv(ops, K.CONV_CASES_SMALL[3], "cpu")
    K.check_attention(ops, K.ATTN_CASES_SMALL[0], "cpu")
    K.check_attention(ops, K.ATTN_CASES_LDS[0], "cpu")
    K.check_splitk_layernorm(ops, "cpu", B=1, S=3, T=5, c=1280)


@pytest.mark.parametrize("tile,n,cin,k,T", [(1, 70, 64, 5, 150), (6, 80, 64, 5, 150), (4, 40, 40, 3, 300), (1, 70, 32, 1, 90), (3, 70, 64, 3, 150), (10, 150, 64, 3, 200)])
def test_conv_gemm_two_deep_ring_is_bit_identical(ops, tile, n, cin, k, T):
    K.check_conv_ring2(ops, "cpu", tile, n, cin=cin, k=k, T=T)


@pytest.mark.parametrize("n,cin,k,T,partials", [(200, 64, 3, 300, False), (70, 40, 5, 150, False), (160, 128, 1, 260, True)])
def test_conv_gemm_eight_wave_tile_equals_the_four_wave_tile(ops, n, cin, k, T, partials):
    K.check_conv_w8(ops, "cpu", n, cin=cin, k=k, T=T, partials=partials)
