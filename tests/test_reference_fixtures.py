"""Holds the package to fixtures produced by the reference's OWN code (tests/golden/reference_*.json,
written by tests/golden/make_reference_fixtures.py in the build container, where
audio_tokens_config.py, processors/dataset_splitter.py and utils/set_seed.py of danavery/audio-tokens
import and run).  These are the only reference-held vectors that exist for this path: everything else
the reference computes lives in torchaudio / faiss, absent here."""
import dataclasses
import json
import random
from pathlib import Path

import numpy as np
import pytest
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"


def _load(name):
    return json.loads((GOLDEN / name).read_text())


def _expand(v, base):
    if isinstance(v, dict) and "__path__" in v:
        return Path(_expand(v["__path__"], base))
    if isinstance(v, str):
        return v.replace("{BASE_DIR}", base)
    if isinstance(v, list):
        return [_expand(e, base) for e in v]
    return v


def test_config_matches_the_reference_dataclass_field_for_field():
    from audio_tokens_amd import audio_tokens_config as mine
    fx = _load("reference_config_fields.json")
    fields = dataclasses.fields(mine.AudioTokensConfig)
    assert [f.name for f in fields] == [f["name"] for f in fx["fields"]]          # names AND order
    cfg = mine.AudioTokensConfig()
    for f, ref in zip(fields, fx["fields"]):
        tname = str(f.type).replace("typing.", "") if str(f.type).startswith("typing.") else f.type.__name__
        assert tname == ref["type"], f.name
        want = _expand(ref["default"], mine.BASE_DIR)
        got = getattr(cfg, f.name)
        assert type(got) is type(want) and got == want, (f.name, got, want)
    names = {f.name for f in fields}
    attrs = {k: v for k, v in vars(mine.AudioTokensConfig).items()
             if not k.startswith("_") and k not in names and not callable(v)}
    assert attrs == fx["class_attributes"]                                          # un-annotated class attributes
    # keyword construction as the reference's callers use it
    assert mine.AudioTokensConfig(vocab_size=8192, n_mels=128).vocab_size == 8192


@pytest.mark.parametrize("case", range(3))
def test_split_matches_the_reference_splitter(case):
    from audio_tokens_amd.audio_tokens_config import AudioTokensConfig
    from audio_tokens_amd.processors.dataset_splitter import DatasetSplitter
    fx = _load("reference_split.json")
    c = fx["cases"][case]
    cfg = AudioTokensConfig(random_seed=c["random_seed"], dataset_ratio=c["dataset_ratio"],
                            validation_ratio=c["validation_ratio"])
    train, val = DatasetSplitter(cfg)._split_data(list(fx["ytids"]))
    assert train == c["train"] and val == c["validation"]


def test_split_file_document_and_reader(tmp_path):
    """The file the reference's create_split_file wrote is byte-for-byte what ours writes, and the
    generator's reader (SpectrogramGenerator.__init__ -> load_split) takes the reference's bytes."""
    from audio_tokens_amd.audio_tokens_config import AudioTokensConfig
    from audio_tokens_amd.processors.dataset_splitter import DatasetSplitter, load_split
    fx = _load("reference_split.json")
    w = fx["written_file"]

    class Meta:
        def get_all_ytids(self):
            return list(fx["ytids"])

    cfg = AudioTokensConfig(split_file=str(tmp_path / "out" / "split.json"), dataset_ratio=w["dataset_ratio"],
                            validation_ratio=w["validation_ratio"], random_seed=w["random_seed"])
    DatasetSplitter(cfg).create_split_file(Meta())
    assert Path(cfg.split_file).read_text() == w["text"]
    ref_file = tmp_path / "reference_written.json"
    ref_file.write_text(w["text"])
    split = load_split(ref_file)
    assert sorted(split) == ["train", "validation"]
    assert len(split["train"]) == 90 and len(split["validation"]) == 10
    assert all(isinstance(y, str) for y in split["train"] + split["validation"])


def test_set_seed_draws_match_the_reference():
    from audio_tokens_amd.utils.set_seed import set_seed
    fx = _load("reference_set_seed.json")
    same_libs = (fx["versions"]["numpy"] == np.__version__ and fx["versions"]["torch"] == torch.__version__.split("+")[0])
    for d in fx["draws"]:
        set_seed(d["seed"])
        assert [random.random() for _ in range(4)] == d["python_random"]
        got_np = np.random.rand(4).tolist()
        got_ri = np.random.randint(0, 1000, size=4).tolist()
        got_t = torch.rand(4, dtype=torch.float64).tolist()
        got_p = torch.randperm(8).tolist()
        if same_libs:   # (numpy / torch streams are only promised within a version)
            assert got_np == d["numpy_rand"] and got_ri == d["numpy_randint_1000"]
            assert got_t == d["torch_rand_cpu"] and got_p == d["torch_randperm_8"]
