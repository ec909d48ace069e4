"""run_pipeline.py of the reference (run_pipeline.py:8-14 of danavery/audio-tokens), hot-path
stages only: spectrograms -> k-means vocabulary -> tokens.  ModelTrainer (the classifier that
consumes the tokens) is plain PyTorch in the reference and runs on ROCm unchanged; it is not part of
this package, so it is chained only if the caller passes one in."""
from .audio_tokens_config import AudioTokensConfig
from .processors.cluster_creator import ClusterCreator
from .processors.spec_tokenizer import SpecTokenizer
from .processors.spectrogram_generator import SpectrogramGenerator


def main(config=None, model_trainer_cls=None):
    config = config or AudioTokensConfig()

    SpectrogramGenerator(config).run()
    ClusterCreator(config).run()
    SpecTokenizer(config).run()
    if model_trainer_cls is not None:
        model_trainer_cls(config).run()


if __name__ == "__main__":
    main()
