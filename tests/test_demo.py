"""Demo driver: frame writer + the command line counterpart of `src.main2 --mode demo`."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch


def test_frame_writer_raw_file_sidecar_and_pipe(tmp_path):
    """uint8 frames go out byte for byte as rgb24, to a file (+ JSON side-car) or into any pipe; an .mp4 target is
    refused loudly when ffmpeg is not installed (main2.py:342-384 needs cv2 + ffmpeg; this image has neither)."""
    from audio_motion_avatar_amd.demo import FrameWriter

    frames = torch.randint(0, 256, (5, 8, 12, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0))
    path = tmp_path / "clip.rgb"
    with FrameWriter(str(path), 8, 12, fps=24.0) as w:
        w.write(frames[:2])
        w.write(frames[2:])
    assert np.array_equal(np.fromfile(path, dtype=np.uint8).reshape(5, 8, 12, 3), frames.numpy())
    assert json.load(open(str(path) + ".json")) == {"pix_fmt": "rgb24", "width": 12, "height": 8, "fps": 24.0, "frames": 5}
    # a pipe into another process (what an encoder would be)
    sink = tmp_path / "piped.rgb"
    with open(sink, "wb") as fh:
        proc = subprocess.Popen(["cat"], stdin=subprocess.PIPE, stdout=fh)
        w = FrameWriter(proc.stdin, 8, 12)
        w.write(frames)
        w.close()
        proc.stdin.close()
        assert proc.wait() == 0
    assert np.array_equal(np.fromfile(sink, dtype=np.uint8), frames.numpy().ravel())
    if shutil.which("ffmpeg") is None:
        with pytest.raises(RuntimeError, match="ffmpeg"):
            FrameWriter(str(tmp_path / "clip.mp4"), 8, 12)


def test_frame_writer_drives_an_encoder_process(tmp_path):
    """An .mp4 target starts `ffmpeg` with rawvideo rgb24 on stdin, the clip's audio as second input (main2.py:366-382)
    and the frames piped in; a stand-in executable records its command line and stdin (ffmpeg is not in this image)."""
    from audio_motion_avatar_amd.demo import FrameWriter

    fake = tmp_path / "ffmpeg"
    fake.write_text('#!/bin/sh\nprintf "%s\\n" "$@" > "$(dirname "$0")/args.txt"\ncat > "$(dirname "$0")/stdin.bin"\n')
    fake.chmod(0o755)
    frames = torch.randint(0, 256, (3, 4, 6, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
    with FrameWriter(str(tmp_path / "clip.mp4"), 4, 6, fps=24.0, audio_path="speech.wav", ffmpeg=str(fake)) as w:
        w.write(frames)
    args = (tmp_path / "args.txt").read_text().split("\n")
    assert args[:10] == ["-y", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", "6x4", "-r", "24.0", "-i"]
    assert "speech.wav" in args and "aac" in args and "-shortest" in args and args[-2] == str(tmp_path / "clip.mp4")
    assert np.array_equal(np.fromfile(tmp_path / "stdin.bin", dtype=np.uint8), frames.numpy().ravel())
    failing = tmp_path / "ffmpeg_bad"
    failing.write_text("#!/bin/sh\ncat > /dev/null\nexit 3\n")
    failing.chmod(0o755)
    w = FrameWriter(str(tmp_path / "bad.mp4"), 4, 6, ffmpeg=str(failing))
    w.write(frames)
    with pytest.raises(RuntimeError, match="ffmpeg failed"):
        w.close()


@pytest.mark.gpu
def test_demo_command_line_renders_a_clip(tmp_path):
    """`python -m audio_motion_avatar_amd.demo`: seeded tokens + a WAV file at 22.05 kHz -> resample -> Wav2Vec2 ->
    chained windows -> frames on disk; the frames equal harness.rollout quantised like main2.py:351."""
    import wave

    from audio_motion_avatar_amd import demo, ops

    g = torch.Generator().manual_seed(1)
    pcm = (torch.randn(1, 22050, generator=g) * 0.1 * 32768).clamp(-32768, 32767).short()
    with wave.open(str(tmp_path / "a.wav"), "wb") as w:
        w.setnchannels(1), w.setsampwidth(2), w.setframerate(22050)
        w.writeframes(pcm.numpy().tobytes())
    out = tmp_path / "clip.rgb"
    rc = demo.main(["--frames", "7", "--image-size", "64", "48", "--audio", str(tmp_path / "a.wav"), "--out", str(out)])
    assert rc == 0
    meta = json.load(open(str(out) + ".json"))
    assert meta["frames"] == 7 and (meta["height"], meta["width"]) == (64, 48)
    data = np.fromfile(out, dtype=np.uint8).reshape(7, 64, 48, 3)
    assert data.min() < 250 and (data == 255).mean() > 0.2  # an avatar on a white background
    # the interleaved form (two chains zipped) through the same entry point
    out2 = tmp_path / "clip2.rgb"
    assert demo.main(["--frames", "12", "--image-size", "64", "48", "--interleave", "--out", str(out2)]) == 0
    assert os.path.getsize(out2) == 12 * 64 * 48 * 3
