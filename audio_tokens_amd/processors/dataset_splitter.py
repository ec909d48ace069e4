"""DatasetSplitter -- writes the split JSON SpectrogramGenerator reads ({"train": [...], "validation": [...]}).

Stands in for processors/dataset_splitter.py:11-42 of danavery/audio-tokens (stdlib only, left of the
hot path): python's `random` seeded with config.random_seed shuffles the ytids in place, the first
`dataset_ratio` of them are kept and the last `validation_ratio` of those become the validation set.
tests/test_reference_fixtures.py holds it to lists the reference's own class produced
(tests/golden/reference_split.json).
"""
import json
import logging
import random
from pathlib import Path


def split_ytids(ytids, random_seed, dataset_ratio, validation_ratio):
    """-> (train, validation).  `ytids` is shuffled in place, as the reference does."""
    random.seed(random_seed)
    random.shuffle(ytids)
    kept = ytids[: int(len(ytids) * dataset_ratio)]
    n_train = int(len(kept) * (1 - validation_ratio))
    return kept[:n_train], kept[n_train:]


def load_split(path):
    """The document create_split_file wrote -> {"train": [ytid...], "validation": [ytid...]}."""
    with open(path, "r") as f:
        return json.load(f)


class DatasetSplitter:
    def __init__(self, config):
        self.config = config
        self.logger = logging.getLogger(__name__)

    def _split_data(self, ytids):
        c = self.config
        return split_ytids(ytids, c.random_seed, c.dataset_ratio, c.validation_ratio)

    def create_split_file(self, metadata_processor):
        """metadata_processor: anything with get_all_ytids() (the reference's AudiosetMetadataProcessor)."""
        train, validation = self._split_data(metadata_processor.get_all_ytids())
        out = Path(self.config.split_file)
        out.parent.mkdir(parents=True, exist_ok=True)
        with out.open("w") as f:
            json.dump({"train": train, "validation": validation}, f)
        self.logger.info(f"Split file created at {out}: {len(train)} train / {len(validation)} validation")
