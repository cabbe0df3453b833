"""Audio front-end: waveform -> one Wav2Vec2 feature vector per video frame.

Mirror of `GaussianAudioDataset._extract_audio_features` (src/datasets/dataset_speech_vid.py:37-116), which the
reference runs on the CPU at dataset construction: mono mix, crop to the video length at its hard-coded 30 fps
(:54-62), clips of `clip_length` frames through `Wav2Vec2Model` (:68-89; the processor's per-clip zero-mean /
unit-variance normalisation included), mean-pool `time_steps // frames` hidden states per frame (:94-105), pad or
trim to the frame count (:107-113).  On-wire format `[T, 768]` fp32 -- what `AudioTriplaneNet.forward` consumes.

Here the model runs on the MI355X through PyTorch-ROCm library kernels (MIOpen convolutions, rocBLAS GEMMs): this is
plumbing around the hot path, not one of its hand-written kernels.  The checkpoint `facebook/wav2vec2-base-960h` is
not available offline; `build_wav2vec2()` constructs the same architecture with random weights (or loads a local
directory when one is given).

Resampling (`torchaudio.transforms.Resample(sr, 16000)`, dataset_speech_vid.py:40-42) and WAV reading
(`torchaudio.load`, :39) are restated here because torchaudio is absent from this image: `resample()` is torchaudio's
default band-limited sinc interpolation (Hann window, lowpass_filter_width 6, rolloff 0.99) as one strided
convolution on the device, `load_wav()` reads PCM WAV files through the standard library.  PARITY UNPINNED for both
(no torchaudio to compare with); pinned instead by known answers (tests/test_audio_frontend.py: identity at equal
rates, a sine keeps its frequency and amplitude, band-limiting, length = ceil(n * new / orig)).
"""
import math

import torch
import torch.nn.functional as F


def load_wav(path):
    """PCM WAV (8/16/24/32-bit integer) -> (waveform [channels, samples] float32 in [-1, 1), sample_rate), like
    torchaudio.load(path) with its default normalisation (dataset_speech_vid.py:39)."""
    import wave

    import numpy as np

    with wave.open(path, "rb") as w:
        ch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 1:
        data = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 2:
        data = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        data = np.where(v >= 1 << 23, v - (1 << 24), v).astype(np.float32) / float(1 << 23)
    elif width == 4:
        data = np.frombuffer(raw, dtype="<i4").astype(np.float32) / float(1 << 31)
    else:
        raise ValueError(f"{path}: unsupported sample width {width}")
    return torch.from_numpy(data.reshape(-1, ch).T.copy()), sr


def _sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width, rolloff, device):
    """The polyphase filter bank of torchaudio.functional.resample (sinc_interp_hann): [new, 1, 2*width + orig]."""
    base_freq = min(orig_freq, new_freq) * rolloff
    width = math.ceil(lowpass_filter_width * orig_freq / base_freq)
    idx = torch.arange(-width, width + orig_freq, dtype=torch.float64, device=device)[None, None] / orig_freq
    t = torch.arange(0, -new_freq, -1, dtype=torch.float64, device=device)[:, None, None] / new_freq + idx
    t = (t * base_freq).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base_freq / orig_freq)
    return kernels.to(torch.float32), width


def resample(waveform, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """waveform [..., samples] at orig_freq Hz -> [..., ceil(samples * new / orig)] at new_freq Hz."""
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq <= 0 or new_freq <= 0:
        raise ValueError("sample rates must be positive")
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(orig_freq, new_freq)
    o, n = orig_freq // g, new_freq // g
    kernel, width = _sinc_resample_kernel(o, n, lowpass_filter_width, rolloff, waveform.device)
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1]).to(torch.float32)
    length = x.shape[1]
    x = F.pad(x, (width, width + o))
    y = F.conv1d(x[:, None], kernel, stride=o)              # [rows, n, frames]
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    target = int(math.ceil(n * length / o))
    return y[:, :target].reshape(shape[:-1] + (target,))


def build_wav2vec2(model_path=None, device="cuda", seed=0):
    from transformers import Wav2Vec2Config, Wav2Vec2Model

    if model_path:
        model = Wav2Vec2Model.from_pretrained(model_path, local_files_only=True)
    else:
        torch.manual_seed(seed)
        model = Wav2Vec2Model(Wav2Vec2Config())  # base-960h architecture: 7 conv layers, 12 x 768 transformer
    return model.to(device).eval()


def normalize_clip(clip):
    """Wav2Vec2FeatureExtractor(do_normalize=True), the audio half of the Wav2Vec2Processor the reference calls
    (dataset_speech_vid.py:48,88): (x - mean) / sqrt(var + 1e-7) per clip (pinned against the class itself in
    tests/test_audio_frontend.py)."""
    return (clip - clip.mean()) / torch.sqrt(clip.var(unbiased=False) + 1e-7)


_normalize = normalize_clip


@torch.no_grad()
def extract_audio_features(waveform, sr, frames_count, model, clip_length=8, sample_rate=16000,
                           estimated_frame_rate=30):
    """waveform [channels, samples] (or [samples]) at `sr` Hz -> features [frames_count, hidden] on the model's device."""
    device = next(model.parameters()).device
    waveform = waveform.to(device=device, dtype=torch.float32)
    if waveform.dim() == 1:
        waveform = waveform[None]
    if sr != sample_rate:                                                    # :40-42
        waveform = resample(waveform, sr, sample_rate)
    if waveform.shape[0] > 1:
        waveform = waveform.mean(dim=0, keepdim=True)                       # :44-45
    audio_duration = waveform.shape[1] / sample_rate
    video_duration = frames_count / estimated_frame_rate                     # :54-55 (30 fps is hard-coded)
    if audio_duration > video_duration:                                      # :57-62
        waveform = waveform[:, : int(video_duration * sample_rate)]
        audio_duration = video_duration
    frame_duration = audio_duration / frames_count
    # clip boundaries as the reference cuts them (:68-89) ...
    clips = []
    for start_idx in range(0, frames_count, clip_length):
        end_idx = min(start_idx + clip_length, frames_count)
        start_sample = int(start_idx * frame_duration * sample_rate)
        end_sample = min(int(end_idx * frame_duration * sample_rate), waveform.shape[1])
        if start_sample >= end_sample:
            start_sample = max(0, waveform.shape[1] - int((end_idx - start_idx) * frame_duration * sample_rate))
            end_sample = waveform.shape[1]
        clips.append((start_sample, end_sample, end_idx - start_idx))
    # ... but clips of equal length go through the model as ONE batch (the reference feeds them one by one on the
    # CPU): the base model has no cross-sample operation -- GroupNorm in the feature extractor and the positional
    # convolution are per sample, there is no padding and hence no attention mask -- so each row of the batch is the
    # clip's own result up to GEMM summation order.
    hidden_of = [None] * len(clips)
    by_length = {}
    for ci, (a, b, _) in enumerate(clips):
        by_length.setdefault(b - a, []).append(ci)
    for members in by_length.values():
        batch = torch.stack([_normalize(waveform[0, clips[ci][0]:clips[ci][1]]) for ci in members])
        hidden = model(batch).last_hidden_state                              # [clips, steps, hidden]
        for row, ci in enumerate(members):
            hidden_of[ci] = hidden[row]
    features = []
    for ci, (_, _, frames_in_clip) in enumerate(clips):                      # :94-105
        hidden = hidden_of[ci]
        steps = hidden.shape[0]
        per = max(1, steps // frames_in_clip)
        for i in range(frames_in_clip):
            a, b = min(i * per, steps - 1), min((i + 1) * per, steps)
            features.append(hidden[a:b].mean(dim=0) if a < b else hidden[a])
    if len(features) < frames_count:                                         # :107-113
        features.extend([features[-1]] * (frames_count - len(features)))
    return torch.stack(features[:frames_count])
