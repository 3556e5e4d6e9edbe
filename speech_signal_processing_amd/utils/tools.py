"""Host-side file I/O of the reference's ``utils/tools.py`` that the hot path's callers use (GMM_UBM.load_files / MFCC_DTW.load_train read
every utterance through ``read``): wav files in, (sampling rate, int16 / float samples) out, plus the timing helper.  Pure host code —
no GPU work happens here; recording / playback (pyaudio, simpleaudio) are out of scope."""
from __future__ import annotations

import time
import wave

import numpy as np


def get_time(start_time=None):
    """utils/tools.py:31-35: the current time, or the seconds elapsed since ``start_time``."""
    if start_time is None:
        return time.time()
    return time.time() - start_time


def wave_read(filename='test.wav'):
    """utils/tools.py:39-41: the wave.Wave_read object of a .wav file."""
    return wave.open(filename, mode='rb')


def read(filename='test.wav'):
    """utils/tools.py:45-47: ``(sampling_freq, audio)`` as scipy.io.wavfile.read returns them — int16 / int32 / uint8 / float32 samples,
    (N,) for mono and (N, channels) otherwise.  PCM and IEEE-float RIFF files are parsed here (RIFF chunk walk; 24-bit PCM comes back as
    int32 left-justified like scipy)."""
    with open(filename, 'rb') as f:
        raw = f.read()
    if len(raw) < 12 or raw[:4] != b'RIFF' or raw[8:12] != b'WAVE':
        raise ValueError("File format %r not understood. Only 'RIFF' WAVE files are supported." % raw[:4])
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(raw):
        cid, size = raw[pos:pos + 4], int.from_bytes(raw[pos + 4:pos + 8], 'little')
        body = raw[pos + 8:pos + 8 + size]
        if cid == b'fmt ':
            fmt = body
        elif cid == b'data':
            data = body
            break
        pos += 8 + size + (size & 1)
    if fmt is None or data is None or len(fmt) < 16:
        raise ValueError("wav file without a 'fmt ' / 'data' chunk")
    tag = int.from_bytes(fmt[0:2], 'little')
    channels = int.from_bytes(fmt[2:4], 'little')
    rate = int.from_bytes(fmt[4:8], 'little')
    bits = int.from_bytes(fmt[14:16], 'little')
    if tag == 0xFFFE and len(fmt) >= 26:  # WAVE_FORMAT_EXTENSIBLE: the sub-format's first two bytes are the real tag
        tag = int.from_bytes(fmt[24:26], 'little')
    if tag == 1:
        if bits == 8:
            audio = np.frombuffer(data, dtype=np.uint8)
        elif bits == 16:
            audio = np.frombuffer(data, dtype='<i2')
        elif bits == 32:
            audio = np.frombuffer(data, dtype='<i4')
        elif bits == 24:
            b = np.frombuffer(data[:len(data) - len(data) % 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            audio = ((b[:, 0] << 8) | (b[:, 1] << 16) | (b[:, 2] << 24)).astype(np.int32)
        else:
            raise ValueError("unsupported PCM bit depth %d" % bits)
    elif tag == 3:
        audio = np.frombuffer(data, dtype='<f4' if bits == 32 else '<f8')
    else:
        raise ValueError("Unknown wave file format: tag %#x" % tag)
    audio = audio.copy()
    if channels > 1:
        audio = audio[:len(audio) - len(audio) % channels].reshape(-1, channels)
    return rate, audio


def save_wave_file(filename, data, channels=1, sampwidth=2, framerate=8000):
    """utils/tools.py:51-58: write the byte strings in ``data`` as one PCM .wav file."""
    wf = wave.open(filename, 'wb')
    wf.setnchannels(channels)
    wf.setsampwidth(sampwidth)
    wf.setframerate(framerate)
    wf.writeframes(b"".join(data))
    wf.close()
