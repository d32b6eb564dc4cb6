"""The whole svcmi facade (weight packing, Flip folding, polyphase transposed convs, chunk driver)
executed on the CPU SIMT emulator at the tiny configuration and compared with the golden vectors of
the real reference.  Slow-ish (fibers), sized to stay within the CPU test budget."""
import pytest

from workload import config as C
from tests import engine_cases as E
from tests.emu import emu_ops


@pytest.fixture(scope="module")
def ops():
    return emu_ops()


def test_vits_tiny_ragged_matches_reference_golden(ops):
    errs = E.check_vits_golden(ops, "cpu", "vits_tiny_ragged", C.tiny_hp())
    print(errs)


def test_whisper_tiny_matches_reference_golden(ops):
    print(E.check_whisper_golden(ops, "cpu", "whisper_tiny", C.WHISPER_TINY_TEST))


def test_logmel_frontend_matches_reference_golden(ops):
    print(E.check_logmel_golden(ops, "cpu"))


def test_hubert_tiny_matches_oracle(ops):
    print(E.check_hubert_against_oracle(ops, "cpu", C.HUBERT_TINY_TEST, n=4000, heads=4))


def test_crepe_tiny_matches_oracle(ops):
    print(E.check_crepe_against_oracle(ops, "cpu", "tiny", n=1600))


def test_crepe_tiny_f16_activation_chain(ops):
    print(E.check_crepe_precision(ops, "cpu", "tiny", 1600, "f16", 2e-2))


def test_svc_infer_with_knn_retrieval(ops):
    print(E.check_svc_infer_retrieval(ops, "cpu", T=10, check_changed=False))


def test_generator_base_widths_match_oracle(ops):
    print(E.check_generator_widths_against_oracle(ops, "cpu", T=3, B=2))


def test_generator_base_widths_bf16x3_within_parity_bar(ops):
    """The reduced-precision GEMM path through the whole facade (single, PARTIALS and grouped launches): split-bf16 keeps the
    waveform inside the 1e-3 bar."""
    print(E.check_generator_widths_against_oracle(ops, "cpu", T=3, B=2, tol=2e-4, precision="bf16x3"))


def test_generator_base_widths_f16_with_16bit_activations(ops):
    """fp16 mode: on the wide stages SnakeAlias writes 16-bit rows and the grouped GEMMs run the _A16 kernels (K-step 64) -- error in
    the fp16 class against the fp32 oracle."""
    print(E.check_generator_widths_against_oracle(ops, "cpu", T=3, B=1, tol=4e-3, precision="f16"))


def test_mixed_precision_policy_switches_modes_per_layer_class(ops):
    print(E.check_mixed_precision_policy(ops, "cpu"))


def test_generator_base_widths_f16_activations_split_f16_weights(ops):
    """SVCMI_PREC_F16W2: fp16 activation rows, (hi, lo) fp16 weight pairs in the launches that read them -- closer to fp32 than plain fp16."""
    e_w2 = E.check_generator_widths_against_oracle(ops, "cpu", T=3, B=1, tol=4e-3, precision="f16w2")
    print(e_w2)


def test_whisper_tiny_f16_operands(ops):
    """fp16 operands (what the reference's `.half()` accelerator path uses, whisper/inference.py:22-23) on the tiny encoder:
    error in the fp16 class, far from fp32's 1e-6 but bounded."""
    print(E.check_whisper_golden(ops, "cpu", "whisper_tiny", C.WHISPER_TINY_TEST, tol=2e-2, precision="f16"))


def test_whisper_tiny_bf16x3_split_activations(ops):
    """bf16x3 mode: the producers hand the GEMMs split (hi, lo) bf16 rows and the _BF16X3_A16 kernel runs -- same products as the
    in-register split, fp32-class error."""
    print(E.check_whisper_golden(ops, "cpu", "whisper_tiny", C.WHISPER_TINY_TEST, tol=1e-4, precision="bf16x3"))


def test_whisper_batched_windows_flatten_their_rows(ops):
    """B * tw rows above small_m_rows: one M = B * tw matrix per projection, items bit-identical to their solo runs."""
    print(E.check_whisper_batched_rows_flattened(ops, "cpu", C.WHISPER_TINY_TEST))


def test_outlier_stress_weights_whisper_and_generator(ops):
    """parity must not rest on N(0, sigma) weights only."""
    print(E.check_whisper_stress(ops, "cpu", C.WHISPER_TINY_TEST, n=120))
    print(E.check_generator_widths_against_oracle(ops, "cpu", T=3, B=1, stress=True))
