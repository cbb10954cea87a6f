"""CPU: what independent implementations in this container can pin of the log-mel restatement (oracle/mel_ref.py; reference call:
data_loader/lmdb_data_loader.py:216-218, librosa 0.8.1 -- not importable here).

* framing, periodic Hann window, reflect padding, FFT: `mel_ref.stft_power` against torch.stft (an independent implementation of
  center=True / pad_mode='reflect' / periodic Hann) and the window against scipy.signal.get_window('hann', N, fftbins=True), which
  is the call librosa's `window='hann'` resolves to;
* the dB stage: `mel_ref.power_to_db` against the closed form of librosa.power_to_db(S, ref=np.max, amin=1e-10, top_db=80);
* the Slaney filter bank stays UNPINNED (no independent implementation of librosa.filters.mel is importable): only the published
  structural properties of norm='slaney' triangles are checked."""
import numpy as np
import torch

from oracle import mel_ref

N_FFT, HOP, SR, N = 1024, 1096, 16000, 36267


def _clips():
    rng = np.random.default_rng(7)
    t = np.arange(N) / SR
    return [rng.standard_normal(N) * 0.1, np.sin(2 * np.pi * 440.0 * t) + 0.3 * np.sin(2 * np.pi * 3000.0 * t),
            np.concatenate([np.zeros(N // 2), rng.standard_normal(N - N // 2)])]


def test_window_is_scipy_hann_fftbins():
    from scipy.signal import get_window
    w = get_window("hann", N_FFT, fftbins=True)
    assert np.abs(mel_ref.hann_periodic(N_FFT) - w).max() <= 1e-15
    assert np.abs(torch.hann_window(N_FFT, periodic=True, dtype=torch.float64).numpy() - w).max() <= 1e-15


def test_stft_power_vs_torch_stft():
    win = torch.hann_window(N_FFT, periodic=True, dtype=torch.float64)
    for y in _clips():
        want = torch.stft(torch.from_numpy(y), n_fft=N_FFT, hop_length=HOP, win_length=N_FFT, window=win, center=True, pad_mode="reflect",
                          normalized=False, onesided=True, return_complex=True).abs().pow(2).t().numpy()      # (frames, 513)
        got = mel_ref.stft_power(y, N_FFT, HOP)
        assert got.shape == want.shape == (34, 513)               # 1 + 36267 // 1096 frames: the 34 frames of a clip
        assert np.abs(got - want).max() <= 1e-9 * max(want.max(), 1e-30)


def test_power_to_db_closed_form():
    rng = np.random.default_rng(1)
    S = np.abs(rng.standard_normal((34, 128))) ** 2 * 10.0 ** rng.uniform(-14, 3, (34, 128))    # spans amin and the 80 dB floor
    got = mel_ref.power_to_db(S)
    ref = max(S.max(), 1e-10)
    want = 10.0 * np.log10(np.maximum(S, 1e-10) / ref)
    want = np.maximum(want, want.max() - 80.0)
    assert np.abs(got - want).max() <= 1e-9
    assert got.max() == 0.0 and got.min() == -80.0                # ref = max; values below max - 80 dB sit on the floor
    assert np.all(mel_ref.power_to_db(np.zeros((4, 8))) == 0.0)   # silence: everything at amin = the reference


def test_slaney_filter_bank_structure():
    """Published properties of librosa.filters.mel(sr, n_fft, n_mels=128, htk=False, norm='slaney'): triangles on the Slaney mel
    scale (linear below 1 kHz at 200/3 Hz per mel, logarithmic above with step log(6.4)/27), peak-normalised by 2 / bandwidth
    (unit area in Hz), float32 weights.  NOT a pin of the numbers."""
    W = mel_ref.mel_basis(SR, N_FFT, 128)
    assert W.shape == (128, 513) and W.min() >= 0.0
    assert np.all(W.astype(np.float32).astype(np.float64) == W)
    df = SR / N_FFT
    centres = (W * np.arange(513)[None, :]).sum(1) / W.sum(1) * df
    assert np.all(np.diff(centres) > 0)                                             # bands ordered by frequency
    wide = (W > 0).sum(1) >= 8                                                       # enough bins for the Riemann sum to mean something
    area = W.sum(1) * df
    assert wide.sum() >= 40 and np.abs(area[wide] - 1.0).max() <= 0.08            # unit area (Slaney normalisation)
    assert abs(mel_ref._hz_to_mel(1000.0) - 15.0) <= 1e-12 and abs(mel_ref._mel_to_hz(mel_ref._hz_to_mel(4000.0)) - 4000.0) <= 1e-9
    lo = centres < 900.0
    assert np.abs(np.diff(centres[lo], 2)).max() <= 3.0                              # equally spaced below 1 kHz
