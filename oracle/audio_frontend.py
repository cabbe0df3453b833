"""CPU restatement of the reference's audio feature extraction (src/datasets/dataset_speech_vid.py:37-116), on the
reference's own dependency (`transformers.Wav2Vec2Model`, importable here; the pretrained weights are not, so tests
hand in a seeded random-weight model).  The reference goes through `Wav2Vec2Processor` and numpy; the arithmetic is
the same per-clip normalisation, model call and per-frame mean pooling.  Test infrastructure only.
"""
import numpy as np
import torch


def normalize_clip(clip):
    """Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm on one clip (numpy, as the processor does it)."""
    return (clip - clip.mean()) / np.sqrt(clip.var() + 1e-7)


def extract_audio_features(waveform, sr, frames_count, model, clip_length=8, sample_rate=16000):
    assert sr == sample_rate, "resampling (torchaudio) is not restated"
    if waveform.shape[0] > 1:
        waveform = torch.mean(waveform, dim=0, keepdim=True)
    audio_duration = waveform.shape[1] / sample_rate
    estimated_frame_rate = 30
    estimated_video_duration = frames_count / estimated_frame_rate
    if audio_duration > estimated_video_duration:
        waveform = waveform[:, : int(estimated_video_duration * sample_rate)]
        audio_duration = estimated_video_duration
    frame_duration = audio_duration / frames_count
    features = []
    for start_idx in range(0, frames_count, clip_length):
        end_idx = min(start_idx + clip_length, frames_count)
        start_sample = int(start_idx * frame_duration * sample_rate)
        end_sample = min(int(end_idx * frame_duration * sample_rate), waveform.shape[1])
        if start_sample >= end_sample:
            start_sample = max(0, waveform.shape[1] - int((end_idx - start_idx) * frame_duration * sample_rate))
            end_sample = waveform.shape[1]
        clip = waveform[:, start_sample:end_sample].numpy().squeeze()
        clip = normalize_clip(clip)
        with torch.no_grad():
            hidden_states = model(torch.from_numpy(clip)[None].float()).last_hidden_state
        time_steps = hidden_states.shape[1]
        frames_in_clip = end_idx - start_idx
        steps_per_frame = max(1, time_steps // frames_in_clip)
        for i in range(frames_in_clip):
            frame_start = min(i * steps_per_frame, time_steps - 1)
            frame_end = min((i + 1) * steps_per_frame, time_steps)
            if frame_start < frame_end:
                features.append(hidden_states[:, frame_start:frame_end, :].mean(dim=1).squeeze().numpy())
            else:
                features.append(hidden_states[:, frame_start:frame_start + 1, :].squeeze().numpy())
    if len(features) < frames_count:
        features.extend([features[-1]] * (frames_count - len(features)))
    return np.stack(features[:frames_count])
