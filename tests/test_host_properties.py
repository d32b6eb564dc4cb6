"""Property tests (hypothesis) of the host-side arithmetic of the path: the chunk schedule with its halo trim, the Whisper /
HuBERT window plans, the pitch shift and CSV round trip, and the LPT sharding -- the product's copies against the oracle's
restatements and against the invariants the reference's loops rely on."""
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import svc_oracle as O
from svcmi import dist as D
from svcmi.hubert.inference import window_plan as hubert_plan
from svcmi.pitch import inference as PI
from svcmi.svc_inference import chunk_schedule, shift_pitch
from svcmi.whisper.inference import window_plan as whisper_plan


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 20000), st.sampled_from([160, 320]))
def test_chunk_schedule_tiles_the_output_exactly(T, hop):
    plan = chunk_schedule(T, hop)
    assert plan == O.chunk_schedule(T, hop)
    if T > 2500 and 1 <= T % 2500 <= 10:
        # quirk of svc_inference.py:108-115 kept as is: when T ends within the 10-frame halo past a chunk boundary, the chunk
        # before the last already runs to the end, and the last chunk repeats those <= 10 frames
        assert plan[-2][1] == plan[-1][1] == T
        return
    kept = 0
    for i, (cs, ce, cso, ceo) in enumerate(plan):
        assert 0 <= cs < ce <= T and ce - cs <= 2500 + 2 * 10
        n = (ce - cs) * hop                         # samples the generator makes for the chunk
        lo, hi = cso, n + ceo                       # python slice [cso:ceo] with negative ceo
        assert 0 <= lo < hi <= n
        # the kept samples of chunk i start where chunk i-1's stopped (svc_inference.py:101-131)
        assert cs * hop + lo == kept
        kept = cs * hop + hi
    assert kept == T * hop - 1                       # the reference drops the very last sample ([..:-1] on the last chunk)


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 16000 * 70))
def test_window_plans_cover_the_audio(n):
    wp = whisper_plan(n)
    assert all(k == (b - a) // 320 for a, b, k in wp)            # kept PPG frames per window (whisper/inference.py:40)
    for plan, win in (([w[:2] for w in wp], 15 * 16000), (hubert_plan(n), 20 * 16000)):
        assert plan[0][0] == 0 and plan[-1][1] == n
        for (a, b), (c, d) in zip(plan, plan[1:]):
            assert b == c and b - a == win           # full windows, back to back
        assert 0 < plan[-1][1] - plan[-1][0] <= win


@settings(max_examples=100, deadline=None)
@given(f0=st.lists(st.integers(0, 1200), min_size=1, max_size=50).filter(lambda v: any(x > 0 for x in v)), shift=st.integers(-12, 12))
def test_pitch_shift_and_csv_round_trip(tmp_path_factory, f0, shift):
    pit = np.asarray(f0, dtype=np.float64)
    out = shift_pitch(pit, shift)
    if shift == 0:
        assert np.array_equal(out, pit)
    else:
        assert np.allclose(out, pit * 2 ** (shift / 12))          # svc_inference.py:185-200
        assert np.array_equal(out == 0, pit == 0)                  # unvoiced frames stay unvoiced
    path = tmp_path_factory.mktemp("csv") / "p.csv"
    PI.save_csv_pitch(pit, str(path))
    back = PI.load_csv_pitch(str(path))
    assert back == [int(v) for v in pit]                           # pitch/inference.py:102-119: integer Hz per 10 ms frame


@settings(max_examples=200, deadline=None)
@given(st.lists(st.integers(1, 10 ** 6), min_size=0, max_size=60), st.integers(1, 8))
def test_lpt_sharding_invariants(lengths, world):
    shards = D.shard_utterances(lengths, world)
    assert len(shards) == world
    assert sorted(i for s in shards for i in s) == list(range(len(lengths)))
    if lengths:
        loads = [sum(lengths[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(lengths)


@settings(max_examples=100, deadline=None)
@given(st.lists(st.floats(0.0, 1200.0, allow_nan=False, width=32), min_size=1, max_size=40))
def test_batch_path_quantises_f0_like_the_csv_round_trip(tmp_path_factory, f0):
    """svc_inference_batch must hand the synthesizer the same F0 as svc_inference (which round-trips the pitch
    CSV, pitch/inference.py:102-119): int() truncation per frame."""
    path = tmp_path_factory.mktemp("csv") / "q.csv"
    PI.save_csv_pitch(f0, str(path))
    assert PI.quantize_pitch_like_csv(f0) == PI.load_csv_pitch(str(path))


def test_batch_path_rejects_nan_f0_like_the_reference():
    import pytest
    with pytest.raises(ValueError):
        PI.quantize_pitch_like_csv([220.0, float("nan")])          # int(nan): the reference's save_csv_pitch fails the same way


@settings(max_examples=100, deadline=None)
@given(st.lists(st.one_of(st.floats(50.0, 1000.0, width=32), st.just(float("nan")), st.just(0.0)), min_size=1, max_size=30),
       st.sampled_from([3, 5, 9]))
def test_mean_filter_keeps_length_and_matches_the_oracle(x, win):
    """clips shorter than the window must keep their length (crepe/filter.py:10-57 pads, it does not grow)."""
    import torch
    from oracle import crepe_oracle as CO
    got = PI._mean_filter_np(np.asarray(x, dtype=np.float32), win)
    want = CO.mean_filter(torch.tensor(x, dtype=torch.float32)[None], win)[0].numpy()
    assert got.shape == want.shape == (len(x),)
    assert np.allclose(got, want, rtol=1e-6, atol=0, equal_nan=True)


def test_streaming_decoder_tiles_reassemble_on_the_emulator():
    """The host side of SynthesizerInfer.stream_frames (BASELINE.json configs[4]) lives in the C++ stage host (csrc/host_stages.hip:
    generator_fwd): tiles of N frames + STREAM_HALO-frame halos, clamped at the chunk ends, copied in / out with svcmi_copy2d_f32.
    Here on the CPU emulator at a size it finishes in seconds (the halo exceeds the clip, so every tile recomputes the clip and keeps
    its own slice: offsets, clamping and the strided copies are what is checked); interior tiles at full size are the GPU test
    test_streaming_decoder_is_bit_identical_and_matches_oracle_on_30s_chunk."""
    import torch
    from tests import engine_cases as E
    from tests.emu import emu_ops
    from workload import config as C, inputs as I
    ops = emu_ops()
    hp = C.tiny_hp()
    m, _ = E.make_model(hp, ops, "cpu")
    d = I.synth_clip(T=4, hp=hp, seed=5, B=1)
    lens = d["lengths"].clone()
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    run = lambda: m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], lens, src, noise=d["enc_noise"])
    m.stream_frames = 100                    # one tile: the untiled generator with split-K off
    whole = run()
    m.stream_frames = 3                      # tiles [0, 3) and [3, 4)
    assert torch.equal(run(), whole)
    m.stream_frames = None
    assert float((run() - whole).abs().max()) <= 2e-6


def test_graph_lanes_refuse_to_run_without_a_gpu():
    """Clips in flight are HIP streams + graphs: no CPU stand-in."""
    import torch
    from svcmi.lanes import GraphLanes, want_hw_queues
    want_hw_queues(8)
    import os
    assert os.environ["GPU_MAX_HW_QUEUES"].isdigit()
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        GraphLanes([lambda: None])


@settings(max_examples=40, deadline=None)
@given(st.lists(st.integers(min_value=0, max_value=5), min_size=1, max_size=12), st.sampled_from([4, 8, 12]), st.integers(min_value=0, max_value=2 ** 31))
def test_faiss_ivf_flat_files_round_trip(sizes, d, seed):
    """write_faiss_ivf_flat -> read_faiss_ivf_flat for arbitrary list occupancies (both size-table encodings: 'full' when more than half of
    the lists are non-empty, 'sprs' otherwise), ids that are not row numbers, and the direct-map variants a faiss file may carry."""
    import struct
    import tempfile
    from svcmi import ivf_index as IV
    rng = np.random.default_rng(seed)
    nlist = len(sizes)
    cent = rng.standard_normal((nlist, d)).astype(np.float32)
    vecs = [rng.standard_normal((m, d)).astype(np.float32) for m in sizes]
    ids = [rng.integers(0, 2 ** 40, size=m).astype(np.int64) for m in sizes]
    with tempfile.TemporaryDirectory() as tmp:
        f = os.path.join(tmp, "x.index")
        IV.write_faiss_ivf_flat(f, cent, vecs, ids)
        raw = open(f, "rb").read()
        assert (b"full" in raw) == (sum(1 for m in sizes if m) > nlist // 2)
        r = IV.read_faiss_ivf_flat(f)
        assert (r["d"], r["nlist"], r["ntotal"], r["nprobe"]) == (d, nlist, sum(sizes), 1)
        assert np.array_equal(r["centroids"], cent)
        for (v, i), v0, i0 in zip(r["lists"], vecs, ids):
            assert np.array_equal(v, v0) and np.array_equal(i, i0)
        # a direct map of type Array (1) with ntotal entries, as faiss writes when the index maintains one: skipped by the reader
        dm = raw.index(b"ilar") - 9                                   # the 1 + 8 bytes of the empty NoMap record
        n = sum(sizes)
        g = os.path.join(tmp, "y.index")
        with open(g, "wb") as out:
            out.write(raw[:dm] + struct.pack("<bQ", 1, n) + np.arange(n, dtype="<i8").tobytes() + raw[dm + 9:])
        r2 = IV.read_faiss_ivf_flat(g)
        assert all(np.array_equal(a[1], b[1]) for a, b in zip(r2["lists"], r["lists"]))
