"""Wire framing of the service wrapper (the step after `to_pcm16`): the int16 stream of one
utterance is cut into frames of `round(frame_length * rate)` samples (20 ms by default,
tts_vits.py:36-38) and every frame travels as the base64 text of its little-endian bytes
(tts_vits.py:219-226; the last frame is simply shorter).  Host-side by nature: the payload is text.

Resampling to a target rate (`librosa.resample`, tts_vits.py:199-200) is NOT provided: librosa /
resampy / soxr are not part of this build, and a resampler of our own could not be pinned to the
reference's output (parity unpinned) — a caller that needs another rate resamples `o` before
`to_pcm16`, as the reference does.
"""
import base64

import numpy as np


def chunk_size(rate, frame_length=0.02):
    """Samples per frame, as tts_vits.py:38 computes it."""
    return int(round(frame_length * rate))


def frame_pcm16(pcm, rate, frame_length=0.02, valid_samples=None):
    """pcm: 1-D int16 (numpy array or torch tensor, any device) of ONE utterance -> list of base64
    strings, one per frame.  `valid_samples` (e.g. 256 * y_lengths[b]) trims the zero padding of a
    batched row first."""
    if hasattr(pcm, "detach"):
        pcm = pcm.detach().cpu().numpy()
    pcm = np.ascontiguousarray(pcm)
    if pcm.dtype != np.int16 or pcm.ndim != 1:
        raise ValueError("pcm must be a 1-D int16 array (the output row of to_pcm16)")
    if valid_samples is not None:
        pcm = pcm[:int(valid_samples)]
    n = chunk_size(rate, frame_length)
    if n <= 0:
        raise ValueError("frame_length * rate must be at least one sample")
    pcm = pcm.astype("<i2", copy=False)                    # ndarray.tobytes() of the reference runs on little-endian hosts
    return [base64.b64encode(pcm[t:t + n].tobytes()).decode("utf-8") for t in range(0, len(pcm), n)]
