"""TEST INFRASTRUCTURE ONLY -- float64 restatement of the reference's log-mel feature
(data_loader/lmdb_data_loader.py:216-218):

    melspec     = librosa.feature.melspectrogram(y=audio_padded, sr=16000, n_fft=1024, hop_length=1096, power=2)
    log_melspec = librosa.power_to_db(melspec, ref=np.max).T

PARITY PINNED BY INDEPENDENT IMPLEMENTATIONS, NOT BY LIBROSA ITSELF: librosa (pinned at 0.8.1 by the reference,
requirements_HOP:35) is a third-party dependency that is neither under /root/reference nor importable in the build container,
and the reference holds no fixture of this feature.  This file restates the published librosa 0.8.1 definitions --
stft(center=True, pad_mode='reflect', window='hann' = scipy get_window('hann', 1024, fftbins=True), win_length = n_fft),
filters.mel(htk=False, norm='slaney', fmin=0, fmax=sr/2), power_to_db(amin=1e-10, top_db=80.0) -- in float64 numpy; only tests
import it.  tests/test_mel_pin.py holds every stage against an implementation written by someone else that IS importable here:
the framing / window / reflect padding / FFT half (`stft_power`) against torch.stft and scipy's window, the Slaney filter bank
(`mel_basis`) against transformers.audio_utils.mel_filter_bank (Hugging Face's numpy implementation of librosa.filters.mel), the
whole mel power against transformers.audio_utils.spectrogram, the dB stage (`power_to_db`) against its closed form.
"""
import numpy as np


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    out = f / f_sp
    big = f >= min_log_hz
    out = np.where(big, min_log_mel + np.log(np.where(big, f, min_log_hz) / min_log_hz) / logstep, out)
    return out


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sr=16000, n_fft=1024, n_mels=128):
    freqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    w = np.zeros((n_mels, freqs.size))
    for i in range(n_mels):
        lower = (freqs - mel_f[i]) / (mel_f[i + 1] - mel_f[i])
        upper = (mel_f[i + 2] - freqs) / (mel_f[i + 2] - mel_f[i + 1])
        w[i] = np.maximum(0.0, np.minimum(lower, upper)) * (2.0 / (mel_f[i + 2] - mel_f[i]))
    return w.astype(np.float32).astype(np.float64)          # librosa returns float32 weights


def hann_periodic(n_fft=1024):
    k = np.arange(n_fft)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n_fft)


def stft_power(y, n_fft=1024, hop=1096):
    """|STFT|^2 of librosa.stft(y, n_fft, hop_length=hop, center=True, pad_mode='reflect', window='hann'): (frames, 1 + n_fft/2)."""
    y = np.asarray(y, dtype=np.float64)
    ypad = np.pad(y, n_fft // 2, mode="reflect")
    window = hann_periodic(n_fft)
    n_frames = 1 + y.size // hop
    frames = np.stack([ypad[t * hop:t * hop + n_fft] * window for t in range(n_frames)])
    return np.abs(np.fft.rfft(frames, axis=1)) ** 2


def power_to_db(S, amin=1e-10, top_db=80.0):
    """librosa.power_to_db(S, ref=np.max, amin, top_db)."""
    log_spec = 10.0 * np.log10(np.maximum(amin, S)) - 10.0 * np.log10(np.maximum(amin, S.max()))
    return np.maximum(log_spec, log_spec.max() - top_db)


def log_melspec(y, sr=16000, n_fft=1024, hop=1096, n_mels=128, amin=1e-10, top_db=80.0):
    """y (n_samples,) -> (1 + n_samples // hop, n_mels) float64."""
    mel = stft_power(y, n_fft, hop) @ mel_basis(sr, n_fft, n_mels).T            # (frames, mels)
    return power_to_db(mel, amin, top_db)
