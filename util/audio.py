"""CPU side of synthesis: linear spectrogram -> waveform (north_star: "Griffin-Lim in util/audio.py stays on CPU").

Only what the Synthesizer needs, on NumPy / SciPy (the reference's util/audio.py:20-36, 65-75, 114-151 uses librosa and a TF graph,
neither of which exists here): de-normalisation and dB -> amplitude with the reference's constants, Griffin-Lim phase reconstruction
on an STFT with the reference's parameters (n_fft = 2 (num_freq - 1), hop = frame_shift_ms, Hann window of frame_length_ms, centred
frames with reflect padding -- librosa's conventions), inverse pre-emphasis and 16-bit wav output.  Not on the training path."""
import numpy as np
from scipy import signal
from scipy.io import wavfile

from hparams import hparams


def stft_parameters(hp=hparams):
    n_fft = (hp.num_freq - 1) * 2
    return n_fft, int(hp.frame_shift_ms / 1000 * hp.sample_rate), int(hp.frame_length_ms / 1000 * hp.sample_rate)


def _window(n_fft, win):
    w = np.zeros(n_fft)
    lo = (n_fft - win) // 2
    w[lo:lo + win] = signal.get_window('hann', win, fftbins=True)
    return w


def stft(y, hp=hparams):
    """[num_freq, frames] complex spectrum of a 1-D signal."""
    n_fft, hop, win = stft_parameters(hp)
    y = np.pad(np.asarray(y, dtype=np.float64), n_fft // 2, mode='reflect')
    frames = 1 + (len(y) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(frames)[:, None]
    return np.fft.rfft(y[idx] * _window(n_fft, win)[None, :], axis=1).T


def istft(S, hp=hparams):
    """least-squares overlap-add inverse of stft()."""
    n_fft, hop, win = stft_parameters(hp)
    w = _window(n_fft, win)
    frames = S.shape[1]
    x = np.fft.irfft(S.T, n=n_fft, axis=1) * w[None, :]
    y = np.zeros(n_fft + hop * (frames - 1))
    norm = np.zeros_like(y)
    for i in range(frames):
        y[i * hop:i * hop + n_fft] += x[i]
        norm[i * hop:i * hop + n_fft] += w * w
    y /= np.maximum(norm, 1e-8)
    return y[n_fft // 2:len(y) - n_fft // 2]


def griffin_lim(magnitude, hp=hparams, seed=0):
    """phase reconstruction for a [num_freq, frames] magnitude spectrogram (hp.griffin_lim_iters iterations)."""
    rng = np.random.RandomState(seed)
    angles = np.exp(2j * np.pi * rng.rand(*magnitude.shape))
    y = istft(magnitude * angles, hp)
    for _ in range(hp.griffin_lim_iters):
        est = stft(y, hp)
        angles = est / np.maximum(1e-8, np.abs(est))
        y = istft(magnitude * angles, hp)
    return y


def inv_preemphasis(x, hp=hparams):
    return signal.lfilter([1], [1, -hp.preemphasis], x)


def inv_spectrogram(spectrogram, hp=hparams):
    """normalised dB spectrogram [num_freq, frames] (the model's linear output, transposed) -> waveform."""
    db = np.clip(spectrogram, 0, 1) * -hp.min_level_db + hp.min_level_db + hp.ref_level_db
    return inv_preemphasis(griffin_lim(np.power(10.0, db * 0.05) ** hp.power, hp), hp)


def save_wav(wav, path, hp=hparams):
    wav = np.asarray(wav, dtype=np.float64)
    wavfile.write(path, hp.sample_rate, (wav * (32767 / max(0.01, np.max(np.abs(wav))))).astype(np.int16))
