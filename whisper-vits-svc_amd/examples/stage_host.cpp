// stage_host.cpp -- a host WITHOUT Python: loads a packed synthesizer (python -m svcmi.tools pack), uploads it, and runs
// Generator.pitch2source + SynthesizerInfer.inference (vits_decoder/generator.py:160-165, vits/models.py:251-256) through the stage-level
// entry points of libsvcmi.so.  What a C / C++ / Go (cgo) / Rust (FFI) serving process would do; tests/test_gpu_engine.py runs it and
// compares the waveform with the Python facade bit for bit.
//
//   stage_host <model.svcmi> <inputs.bin> <wave_out.bin>
// inputs.bin: int32 B, T, then float32 ppg[B][T][ppg_dim], vec[B][T][vec_dim], pit[B][T], spk[B][spk_dim], int32 lengths[B],
//             float32 rand_ini[B][11], src_noise[B][T*hop][11], enc_noise[B][inter][T]      (the explicit draws of the path)
// wave_out.bin: float32 [B][T*hop]
// With a packed Whisper encoder (python -m svcmi.tools pack --whisper ...; kind 2) the same binary runs AudioEncoder.forward
// (whisper/model.py:147-163 with the `mel + 0.1 * noise` of whisper/inference.py:46):
//   stage_host <whisper.svcmi> <mel.bin> <ppg_out.bin>
// mel.bin: int32 B, n_frames, then float32 mel[B][n_mels][n_frames], noise[B][n_mels][n_frames];  ppg_out.bin: float32 [B][tw][n_state]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../include/svcmi.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_SV(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d\n", #x, r_); return 3; } } while (0)

static std::vector<char> read_file(const char* path) {
    std::vector<char> v;
    FILE* f = fopen(path, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize(n);
    if (fread(v.data(), 1, n, f) != (size_t)n) v.clear();
    fclose(f);
    return v;
}

template <typename T>
static T* upload(const char*& p, size_t count) {
    T* d = nullptr;
    if (hipMalloc(&d, count * sizeof(T) + 256) != hipSuccess) return nullptr;
    if (hipMemcpy(d, p, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    p += count * sizeof(T);
    return d;
}

// the Whisper stage: mel (+ noise) -> PPG through svcmi_whisper_encoder_fwd
static int run_whisper(const std::vector<char>& file, const void* arena, const std::vector<char>& in, const char* out_path) {
    static svcmi_whisper_model model;       // (a few KB of block descriptors)
    CHECK_SV(svcmi_packed_model_bind(file.data(), (int64_t)file.size(), arena, &model, sizeof(model)));
    const char* p = in.data();
    const int32_t B = reinterpret_cast<const int32_t*>(p)[0], n = reinterpret_cast<const int32_t*>(p)[1];
    p += 8;
    const size_t mel_count = (size_t)B * model.n_mels * n;
    if (B <= 0 || n <= 0 || in.size() != 8 + 2 * mel_count * sizeof(float)) { fprintf(stderr, "mel.bin does not match B = %d, n_frames = %d\n", B, n); return 1; }
    float* mel = upload<float>(p, mel_count);
    float* noise = upload<float>(p, mel_count);
    if (!mel || !noise) return 2;
    const int tw = (n - 1) / 2 + 1;
    float* ppg = nullptr;
    CHECK_HIP(hipMalloc(&ppg, (size_t)B * tw * model.n_state * sizeof(float)));
    const int64_t ws_bytes = svcmi_whisper_workspace_bytes(&model, B, n);
    if (ws_bytes < 0) { fprintf(stderr, "svcmi_whisper_workspace_bytes -> %lld\n", (long long)ws_bytes); return 3; }
    void* ws = nullptr;
    CHECK_HIP(hipMalloc(&ws, ws_bytes));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_SV(svcmi_whisper_encoder_fwd(&model, mel, noise, 0.1f, B, n, ppg, ws, ws_bytes, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    std::vector<float> out((size_t)B * tw * model.n_state);
    CHECK_HIP(hipMemcpy(out.data(), ppg, out.size() * sizeof(float), hipMemcpyDeviceToHost));
    FILE* f = fopen(out_path, "wb");
    if (!f || fwrite(out.data(), sizeof(float), out.size(), f) != out.size()) { fprintf(stderr, "cannot write %s\n", out_path); return 1; }
    fclose(f);
    printf("stage_host: Whisper encoder, B = %d, %d mel frames -> [%d][%d][%d], workspace %.1f MB\n", B, n, B, tw, model.n_state, ws_bytes / 1e6);
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s model.svcmi inputs.bin wave_out.bin\n", argv[0]); return 1; }
    if (svcmi_abi_version() != SVCMI_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    const std::vector<char> file = read_file(argv[1]), in = read_file(argv[2]);
    if (file.empty() || in.size() < 8) { fprintf(stderr, "cannot read inputs\n"); return 1; }
    int32_t kind = 0;
    int64_t arena_off = 0, arena_bytes = 0;
    CHECK_SV(svcmi_packed_model_info(file.data(), (int64_t)file.size(), &kind, &arena_off, &arena_bytes));
    if (kind != 1 && kind != 2) { fprintf(stderr, "unknown model kind %d\n", kind); return 1; }
    void* arena = nullptr;
    CHECK_HIP(hipMalloc(&arena, arena_bytes));
    CHECK_HIP(hipMemcpy(arena, file.data() + arena_off, arena_bytes, hipMemcpyHostToDevice));
    if (kind == 2) return run_whisper(file, arena, in, argv[3]);
    svcmi_synth_model model;
    CHECK_SV(svcmi_packed_model_bind(file.data(), (int64_t)file.size(), arena, &model, sizeof(model)));

    const char* p = in.data();
    const int32_t B = reinterpret_cast<const int32_t*>(p)[0], T = reinterpret_cast<const int32_t*>(p)[1];
    p += 8;
    const int64_t L = (int64_t)T * model.hop;
    {       // the whole expected size from B and T BEFORE any upload: a short inputs.bin must not be read past its end
        if (B <= 0 || T <= 0 || B > (1 << 20) || T > (1 << 24)) { fprintf(stderr, "inputs.bin: bad B = %d, T = %d\n", B, T); return 1; }
        const unsigned long long bt = (unsigned long long)B * T;
        const unsigned long long want = 8ull + 4ull * (bt * model.ppg_dim + bt * model.vec_dim + bt + (unsigned long long)B * model.spk_dim +
                                                       (unsigned long long)B + 11ull * B + 11ull * bt * model.hop + bt * model.inter);
        if (want != (unsigned long long)in.size()) {
            fprintf(stderr, "inputs.bin does not match B = %d, T = %d (%llu bytes expected, %zu found)\n", B, T, want, in.size());
            return 1;
        }
    }
    float* ppg = upload<float>(p, (size_t)B * T * model.ppg_dim);
    float* vec = upload<float>(p, (size_t)B * T * model.vec_dim);
    float* pit = upload<float>(p, (size_t)B * T);
    float* spk = upload<float>(p, (size_t)B * model.spk_dim);
    int32_t* lengths = upload<int32_t>(p, (size_t)B);
    float* rand_ini = upload<float>(p, (size_t)B * 11);
    float* src_noise = upload<float>(p, (size_t)B * L * 11);
    float* enc_noise = upload<float>(p, (size_t)B * model.inter * T);
    if (!ppg || !vec || !pit || !spk || !lengths || !rand_ini || !src_noise || !enc_noise || p != in.data() + in.size()) {
        fprintf(stderr, "inputs.bin does not match B = %d, T = %d\n", B, T);
        return 1;
    }
    float *source = nullptr, *wave = nullptr;
    CHECK_HIP(hipMalloc(&source, B * L * sizeof(float)));
    CHECK_HIP(hipMalloc(&wave, B * L * sizeof(float)));
    const int64_t ws_bytes = svcmi_synth_workspace_bytes(&model, B, T, 0);
    if (ws_bytes < 0) { fprintf(stderr, "svcmi_synth_workspace_bytes -> %lld\n", (long long)ws_bytes); return 3; }
    void* ws = nullptr;
    CHECK_HIP(hipMalloc(&ws, ws_bytes));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));

    CHECK_SV(svcmi_pitch2source_fwd(&model, pit, rand_ini, src_noise, B, T, source, ws, ws_bytes, stream));
    svcmi_synth_io io = {};
    io.ppg = ppg; io.vec = vec; io.pit = pit; io.spk = spk; io.lengths = lengths; io.source = source; io.noise = enc_noise;
    io.batch = B; io.t = T; io.wave = wave;
    CHECK_SV(svcmi_synth_infer_fwd(&model, &io, ws, ws_bytes, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    std::vector<float> out((size_t)B * L);
    CHECK_HIP(hipMemcpy(out.data(), wave, out.size() * sizeof(float), hipMemcpyDeviceToHost));
    FILE* f = fopen(argv[3], "wb");
    if (!f || fwrite(out.data(), sizeof(float), out.size(), f) != out.size()) { fprintf(stderr, "cannot write %s\n", argv[3]); return 1; }
    fclose(f);
    double acc = 0.0;
    for (float v : out) acc += v < 0 ? -v : v;
    printf("stage_host: B = %d, T = %d, %lld samples, workspace %.1f MB, mean |wave| = %.6f\n", B, T, (long long)(B * L), ws_bytes / 1e6, acc / out.size());
    return 0;
}
