"""audio_tokens_amd -- the audio-tokenisation hot path of danavery/audio-tokens, rebuilt for MI355X.

waveform -> STFT/log-mel -> K-means (FAISS-style Lloyd) -> nearest-centroid tokens, as hand-written
gfx950 HIP kernels behind a C ABI (include/audio_tokens_amd.h), with the reference's Python
surface on top:

    audio_tokens_amd.audio_tokens_config.AudioTokensConfig
    audio_tokens_amd.processors.{SpectrogramGenerator, ClusterCreator, SpecTokenizer}
    audio_tokens_amd.ops.{LogMelSpectrogram, Kmeans, IndexFlatL2}   (torchaudio / faiss stand-ins)
    audio_tokens_amd.run_pipeline.main

Importing the package does not need a GPU; running any operator does (there is no CPU fallback).
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401  (ctypes table; loads the .so lazily)
