// Log-mel spectrogram of the step's audio clips on the GPU, with the parameters of the reference's data loader
// (data_loader/lmdb_data_loader.py:216-218, librosa 0.8.1 per requirements_HOP:35):
//
//   melspec     = librosa.feature.melspectrogram(y, sr=16000, n_fft=1024, hop_length=1096, power=2)
//                   = mel_filter(128 x 513, Slaney scale + norm) . |STFT(y; hann(1024) periodic, center=True, reflect pad)|^2
//   log_melspec = librosa.power_to_db(melspec, ref=np.max).T           10 log10(max(1e-10, S)) - 10 log10(max(1e-10, max S)),
//                                                                      floored at (its maximum - 80 dB); (frames, mels)
//
// so that the input pipeline ships raw audio only and the 34 x 128 feature is made where it is consumed (the reference
// computes it per sample on the host inside Dataset.__getitem__).  Two launches:
//   logmel_power_kernel   one workgroup per (clip, frame): 1024 windowed samples (reflect-padded indexing) -> radix-2
//                         FFT in LDS (fp32, twiddles from sincospif) -> power spectrum -> the 128 triangular filters
//                         (each thread owns one band's compact weight run) -> mel power [B][frames][128]
//   logmel_db_kernel      one workgroup per clip: maximum over the clip's frames x 128 bands (fixed-order tree), dB, floor.
// HBM-bound by definition and tiny (145 KB in, 17 KB out per clip); the point of the kernel is the boundary, not the roofline.
#include "common.h"

namespace hopmi {

constexpr int MEL_NFFT = 1024, MEL_BINS = MEL_NFFT / 2 + 1, MEL_BANDS = 128;

__device__ __forceinline__ int reflect_index(int p, int n) {          // numpy.pad(mode="reflect") index map (n > pad)
  if (p < 0) p = -p;
  if (p >= n) p = 2 * (n - 1) - p;
  return p;
}

__global__ __launch_bounds__(256) void logmel_power_kernel(const float* __restrict__ audio, int n_samples, int hop, int frames,
                                                           const int* __restrict__ band_start, const int* __restrict__ band_len,
                                                           const int* __restrict__ band_off, const float* __restrict__ band_w,
                                                           float* __restrict__ mel_power) {
  __shared__ float re[MEL_NFFT], im[MEL_NFFT];
  __shared__ float pw[MEL_BINS];
  const int tid = threadIdx.x;
  const int clip = blockIdx.x / frames, t = blockIdx.x - clip * frames;
  const float* y = audio + (size_t)clip * n_samples;
  // windowed frame, written in bit-reversed order (decimation in time)
  for (int k = tid; k < MEL_NFFT; k += 256) {
    const int p = reflect_index(t * hop + k - MEL_NFFT / 2, n_samples);
    float s, c;
    sincospif(2.f * (float)k / MEL_NFFT, &s, &c);
    const float w = 0.5f - 0.5f * c;                                   // periodic Hann (scipy get_window(fftbins=True))
    const int r = (int)(__brev((unsigned)k) >> 22);                    // 10-bit reversal
    re[r] = y[p] * w;
    im[r] = 0.f;
  }
  __syncthreads();
  for (int half = 1; half < MEL_NFFT; half <<= 1) {                    // 10 radix-2 stages, 512 butterflies each
    for (int b = tid; b < MEL_NFFT / 2; b += 256) {
      const int j = b & (half - 1);
      const int i0 = ((b - j) << 1) + j, i1 = i0 + half;
      float s, c;
      sincospif(-(float)j / (float)half, &s, &c);                      // exp(-i pi j / half)
      const float xr = re[i1] * c - im[i1] * s, xi = re[i1] * s + im[i1] * c;
      const float ar = re[i0], ai = im[i0];
      re[i0] = ar + xr; im[i0] = ai + xi;
      re[i1] = ar - xr; im[i1] = ai - xi;
    }
    __syncthreads();
  }
  for (int k = tid; k < MEL_BINS; k += 256) pw[k] = re[k] * re[k] + im[k] * im[k];
  __syncthreads();
  if (tid < MEL_BANDS) {
    const int s0 = band_start[tid], n = band_len[tid];
    const float* w = band_w + band_off[tid];
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc += w[i] * pw[s0 + i];              // ascending bin order
    mel_power[((size_t)clip * frames + t) * MEL_BANDS + tid] = acc;
  }
}

__global__ __launch_bounds__(256) void logmel_db_kernel(const float* __restrict__ mel_power, int frames, float amin, float top_db,
                                                        float* __restrict__ out) {
  __shared__ float red[256];
  const int tid = threadIdx.x, n = frames * MEL_BANDS;
  const float* p = mel_power + (size_t)blockIdx.x * n;
  float m = 0.f;
  for (int i = tid; i < n; i += 256) m = fmaxf(m, p[i]);
  red[tid] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
    __syncthreads();
  }
  const float ref_db = 10.f * log10f(fmaxf(amin, red[0]));
  // log_spec = 10 log10(max(amin, S)) - ref_db; its maximum is 10 log10(max(amin, max S)) - ref_db = 0 -> floor at -top_db
  float* o = out + (size_t)blockIdx.x * n;
  for (int i = tid; i < n; i += 256) o[i] = fmaxf(10.f * log10f(fmaxf(amin, p[i])) - ref_db, -top_db);
}

}  // namespace hopmi

using namespace hopmi;

extern "C" int hopmi_logmel(const float* audio, int B, int n_samples, int hop, const int* band_start, const int* band_len,
                            const int* band_off, const float* band_w, float* mel_power_ws, float* out, void* stream) {
  if (!audio || !band_start || !band_len || !band_off || !band_w || !mel_power_ws || !out) { set_error("hopmi_logmel: null pointer argument"); return HOPMI_EINVAL; }
  if (B <= 0 || hop <= 0 || n_samples <= MEL_NFFT / 2) { set_error("hopmi_logmel: bad sizes B=%d n_samples=%d hop=%d", B, n_samples, hop); return HOPMI_EINVAL; }
  const int frames = 1 + n_samples / hop;                               // librosa: center=True
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(logmel_power_kernel, dim3(B * frames), dim3(256), 0, st, audio, n_samples, hop, frames, band_start, band_len,
                     band_off, band_w, mel_power_ws);
  if (int e = check_launch("hopmi_logmel(power)")) return e;
  hipLaunchKernelGGL(logmel_db_kernel, dim3(B), dim3(256), 0, st, mel_power_ws, frames, 1e-10f, 80.f, out);
  return check_launch("hopmi_logmel(db)");
}
