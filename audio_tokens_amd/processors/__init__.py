"""The three hot-path stage classes of danavery/audio-tokens (processors/), MI355X-native."""
from .cluster_creator import ClusterCreator
from .spec_tokenizer import SpecTokenizer
from .spectrogram_generator import SpectrogramGenerator

__all__ = ["SpectrogramGenerator", "ClusterCreator", "SpecTokenizer"]
