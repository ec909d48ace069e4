"""SpectrogramGenerator -- the reference's stage 1 (processors/spectrogram_generator.py:18-146 of
danavery/audio-tokens) on the MI355X log-mel kernel.

Same constructor, methods, outputs (spectrograms/{train,validation}/<ytid>.npy, float32
[n_mels, T]) and skip-and-continue error behaviour.  Differences in HOW, not WHAT:
  * clips of one batch are decoded on the host, stacked by length and pushed through ONE fused
    STFT -> mel -> dB launch per length (the reference launches a dozen kernels per clip);
  * the whole batch comes back in one device->host copy before the per-file np.save.
  * clips at another sample rate go through the device polyphase resampler (ops.Resample).
Audio decoding stays on the CPU: torchaudio.load if torchaudio is importable, otherwise PCM/float
.wav through the standard library and raw float32 .npy waveforms (used by the synthetic tests).
"""
import json
import logging
import os
import shutil
import wave as _wave
from pathlib import Path

import numpy as np
import torch
from tqdm import tqdm

from ..audio_tokens_config import AudioTokensConfig
from ..ops import LogMelSpectrogram, Resample
from .dataset_splitter import load_split

try:  # optional: only used for decoding when it exists
    import torchaudio as _torchaudio
except Exception:  # pragma: no cover - not installed in the build image
    _torchaudio = None


def _load_audio(path: Path):
    """-> (waveform float32 [C, L] on the host, sample_rate).  RuntimeError("Failed to decode audio.")
    is the one failure the reference skips silently (spectrogram_generator.py:98-103)."""
    suffix = path.suffix.lower()
    if suffix == ".npy":
        w = np.load(path)
        w = w[None, :] if w.ndim == 1 else w
        sr_file = path.with_suffix(".sr")
        sr = int(sr_file.read_text()) if sr_file.exists() else 22050
        return torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)), sr
    if _torchaudio is not None:
        return _torchaudio.load(path)
    if suffix == ".wav":
        try:
            with _wave.open(str(path), "rb") as f:
                sr, ch, width, nframes = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
                raw = f.readframes(nframes)
        except (_wave.Error, EOFError) as e:
            raise RuntimeError("Failed to decode audio.") from e
        if width == 2:
            a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
        elif width == 4:
            a = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
        elif width == 1:
            a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        else:
            raise RuntimeError("Failed to decode audio.")
        return torch.from_numpy(a.reshape(-1, ch).T.copy()), sr
    raise RuntimeError("Failed to decode audio.")


class SpectrogramGenerator:
    def __init__(self, config):
        self.config = config
        self.logger = logging.getLogger(__name__)
        self.spec_transformer = LogMelSpectrogram(
            sample_rate=self.config.common_sr,
            n_mels=self.config.n_mels,
            n_fft=self.config.n_fft,
            hop_length=self.config.hop_length,
        )
        # AmplitudeToDB is fused into the kernel; kept as an attribute name for drop-in code
        self.amplitude_to_db_transformer = None
        self.device = self.spec_transformer.backend.device

        self.data_split = load_split(config.split_file)

    def run(self):
        for split in ["train", "validation"]:
            self.logger.info(f"Creating {split} spectrograms")
            output_dir = Path(self.config.dest_spec_path) / split
            shutil.rmtree(output_dir, ignore_errors=True)
            output_dir.mkdir(parents=True)

            ytids = self.data_split[split]
            for i in tqdm(
                range(0, len(ytids), self.config.spectrogram_batch_size),
                total=len(ytids) // self.config.spectrogram_batch_size,
                position=0,
            ):
                batch_ytids = ytids[i: i + self.config.spectrogram_batch_size]
                specs = self.populate_specs(batch_ytids)

                for spec in specs:
                    ytid = os.path.splitext(spec["filename"])[0]
                    output_file = output_dir / f"{ytid}.npy"
                    np.save(output_file, spec["spec"].cpu())
            self.logger.info(f"{split.capitalize()} spectrograms saved to: {output_dir}")

    def populate_specs(self, source_files):
        """-> [{"filename": str, "spec": Tensor[n_mels, T]}] in input order, bad clips skipped.
        The tensors are views of per-length batch results that already live on the host side of
        one bulk copy when `.cpu()` is called on them (they share storage per batch)."""
        waves, names = [], []
        for i, ytid in enumerate(source_files):
            audio_file_path = self.find_audio_file(ytid)
            if not audio_file_path:
                continue
            waveform = self.preprocess_waveform(audio_file_path)
            if waveform is None:
                continue
            waves.append(waveform)
            names.append((i, audio_file_path))

        # one launch per distinct clip length
        by_len = {}
        for j, w in enumerate(waves):
            by_len.setdefault(w.shape[-1], []).append(j)
        specs_by_j = {}
        for L, js in by_len.items():
            if L <= self.config.n_fft // 2:
                self.logger.debug(f"clips shorter than the reflect padding skipped: {len(js)}")
                continue
            batch = torch.stack([waves[j].reshape(-1).to(self.device) for j in js])
            if self.config.normalize:
                # the batch's log-mel with the per-clip (spec - min) / (max - min) behind it in one call: the extremes come
                # out of the log-mel kernel itself (the reference: three torch reductions and two passes per clip)
                st = self.spec_transformer
                out = st.backend.logmel_minmax(batch, st.sample_rate, st.n_fft, st.hop_length, st.n_mels, fb=st.fb)
            else:
                out = self.spec_transformer(batch)                   # [B, n_mels, T] on the GPU
            finite = torch.isfinite(out).flatten(1).all(dim=1).cpu()
            out = out.cpu()  # one bulk device->host copy; the per-file .cpu() in run() is then free
            for b, j in enumerate(js):
                if not bool(finite[b]):
                    self.check_for_nan_inf(out[b], f"spectrogram {names[j][0]}")
                    self.logger.debug(f"Bad file: {names[j][1]}")
                    continue
                specs_by_j[j] = out[b]
        specs = []
        for j in sorted(specs_by_j):
            specs.append({"filename": os.path.basename(names[j][1]), "spec": specs_by_j[j]})
        return specs

    def find_audio_file(self, ytid):
        audio_file_path = None
        for source_set in self.config.audio_source_sets:
            stem = f"{self.config.audio_source_path}/{source_set}/{ytid[:2]}/{ytid}"
            for ext in (".flac", ".wav", ".npy"):  # the reference looks for .flac only
                audio_file_path = Path(stem + ext)
                if audio_file_path.exists():
                    return audio_file_path
        self.logger.debug(f"Audio file not found: {audio_file_path}")
        return None

    def preprocess_waveform(self, audio_file_path):
        try:
            waveform, sr = _load_audio(Path(audio_file_path))
        except RuntimeError as e:
            if str(e) == "Failed to decode audio.":
                self.logger.info(f"skipping {audio_file_path}: {e}")
                return None
            raise
        waveform = self.convert_to_mono(waveform)
        waveform = self.resample(waveform, sr)
        return waveform

    @staticmethod
    def convert_to_mono(waveform):
        if waveform.shape[0] > 1:  # stereo or surround
            return torch.mean(waveform, dim=0, keepdim=True)
        return waveform

    def resample(self, waveform, sr):
        """torchaudio.transforms.Resample(sr, common_sr)(waveform) on the device kernel; the taps of
        the last rate pair stay resident, so a dataset at one native rate builds them once."""
        if sr != self.config.common_sr:
            waveform = Resample(sr, self.config.common_sr, backend=self.spec_transformer.backend)(waveform)
        return waveform

    def generate_mel_spectrogram(self, audio):
        """audio [1, L] -> [n_mels, T] dB (MelSpectrogram + AmplitudeToDB in one kernel)."""
        return self.spec_transformer(audio).squeeze(0)

    @staticmethod
    def normalize_spectrogram(spec):
        return (spec - torch.min(spec)) / (torch.max(spec) - torch.min(spec))

    def check_for_nan_inf(self, data, name="data"):
        if torch.isnan(data).any():
            self.logger.debug(f"Warning: NaN values found in {name}")
            return True
        if torch.isinf(data).any():
            self.logger.debug(f"Warning: Inf values found in {name}")
            return True
        return False


if __name__ == "__main__":
    SpectrogramGenerator(AudioTokensConfig()).run()
