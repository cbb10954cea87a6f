"""CPU: what independent implementations in this container can pin of the log-mel restatement (oracle/mel_ref.py; reference call:
data_loader/lmdb_data_loader.py:216-218, librosa 0.8.1 -- not importable here).

* framing, periodic Hann window, reflect padding, FFT: `mel_ref.stft_power` against torch.stft (an independent implementation of
  center=True / pad_mode='reflect' / periodic Hann) and the window against scipy.signal.get_window('hann', N, fftbins=True), which
  is the call librosa's `window='hann'` resolves to;
* the dB stage: `mel_ref.power_to_db` against the closed form of librosa.power_to_db(S, ref=np.max, amin=1e-10, top_db=80);
* the Slaney filter bank: `mel_ref.mel_basis` against transformers.audio_utils.mel_filter_bank(norm='slaney', mel_scale='slaney') --
  Hugging Face's own numpy implementation of librosa.filters.mel (the one Whisper's feature extractor is built on and that their
  tests hold against librosa) -- and its published structural properties;
* the whole feature: `mel_ref.log_melspec` against transformers.audio_utils.spectrogram (their framing, window, FFT and filter
  bank) followed by the closed-form dB stage.
None of this is librosa itself: three independent implementations (torch, scipy, transformers) of the same published definitions."""
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


def test_slaney_filter_bank_vs_transformers():
    """The numbers of the filter bank against an independent implementation: float64 triangles of transformers' mel_filter_bank
    vs. the oracle's (which rounds to float32 as librosa 0.8.1's filters.mel returns float32): equal to float32 rounding."""
    from transformers.audio_utils import mel_filter_bank
    W = mel_ref.mel_basis(SR, N_FFT, 128)
    F = mel_filter_bank(num_frequency_bins=1 + N_FFT // 2, num_mel_filters=128, min_frequency=0.0, max_frequency=SR / 2.0,
                        sampling_rate=SR, norm="slaney", mel_scale="slaney").T
    assert F.shape == W.shape == (128, 513)
    assert np.abs(F - W).max() <= 2.0 ** -23 * W.max()
    assert np.array_equal(F > 0, W > 0) or np.abs(F - W)[(F > 0) != (W > 0)].max() <= 1e-12        # same support (up to an edge bin at ~0)


def test_log_mel_vs_transformers_spectrogram():
    """The whole feature (lmdb_data_loader.py:216-218) against transformers.audio_utils.spectrogram -- their framing, reflect
    padding, periodic Hann window, FFT and Slaney filter bank, float64 -- followed by power_to_db(ref=max, amin=1e-10, top_db=80)
    in closed form: mel power to float32 rounding of the filter weights, dB to 1e-5 dB."""
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    F = mel_filter_bank(1 + N_FFT // 2, 128, 0.0, SR / 2.0, SR, norm="slaney", mel_scale="slaney")
    win = window_function(N_FFT, "hann", periodic=True)
    for y in _clips():
        want = spectrogram(y, win, frame_length=N_FFT, hop_length=HOP, fft_length=N_FFT, power=2.0, center=True, pad_mode="reflect",
                           mel_filters=F, mel_floor=0.0, log_mel=None, dtype=np.float64).T                     # (34, 128) mel power
        got = mel_ref.stft_power(y, N_FFT, HOP) @ mel_ref.mel_basis(SR, N_FFT, 128).T
        assert got.shape == want.shape == (34, 128)
        assert np.abs(got - want).max() <= 4e-7 * want.max()
        wdb = 10.0 * np.log10(np.maximum(want, 1e-10) / max(want.max(), 1e-10))
        wdb = np.maximum(wdb, wdb.max() - 80.0)
        assert np.abs(mel_ref.log_melspec(y) - wdb).max() <= 1e-5
