#!/usr/bin/env python3
"""Writes the reference-held fixtures: tests/golden/reference_*.json.

Runs in the BUILD CONTAINER ONLY (it needs /root/reference; the GPU box has no copy).  It imports
the three pieces of danavery/audio-tokens that do import here -- everything else on the hot path
needs torchaudio / faiss, which are absent -- runs them, and stores what they produce as plain data:

  reference_config_fields.json   audio_tokens_config.py:14-81   dataclass field names, declared
                                 types and defaults, the two un-annotated class attributes, with
                                 BASE_DIR factored out of the path-valued defaults
  reference_split.json           processors/dataset_splitter.py:20-42   `_split_data` on a fixed
                                 list of ytids for three (seed, dataset_ratio, validation_ratio)
                                 settings, and the JSON document `create_split_file` writes
  reference_set_seed.json        utils/set_seed.py:6-12   the first draws of python / numpy / torch
                                 (CPU generator) after set_seed(4242) and set_seed(7)

Only outputs travel; no source text of the reference is stored.

    python tests/golden/make_reference_fixtures.py
"""
import dataclasses
import sys as _sys
_sys.dont_write_bytecode = True   # nothing is written under /root/reference
import json
import random
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")


def _jsonable(v, base_dir):
    if isinstance(v, Path):
        return {"__path__": _jsonable(str(v), base_dir)}
    if isinstance(v, str):
        return v.replace(base_dir, "{BASE_DIR}")
    if isinstance(v, (list, tuple)):
        return [_jsonable(e, base_dir) for e in v]
    return v


def config_fixture():
    import audio_tokens_config as ref_cfg
    base = ref_cfg.BASE_DIR
    cfg = ref_cfg.AudioTokensConfig()
    fields = []
    for f in dataclasses.fields(ref_cfg.AudioTokensConfig):
        tname = getattr(f.type, "__name__", None) or str(f.type).replace("typing.", "")
        if str(f.type).startswith("typing."):
            tname = str(f.type).replace("typing.", "")
        fields.append({"name": f.name, "type": tname, "default": _jsonable(getattr(cfg, f.name), base)})
    names = {f["name"] for f in fields}
    class_attrs = {k: v for k, v in vars(ref_cfg.AudioTokensConfig).items()
                   if not k.startswith("_") and k not in names and not callable(v)}
    return {"source": "audio_tokens_config.py:14-81 (danavery/audio-tokens), imported and instantiated",
            "fields": fields, "class_attributes": class_attrs}


class _FixedYtids:
    """Stands in for AudiosetMetadataProcessor: create_split_file only calls get_all_ytids()."""

    def __init__(self, ytids):
        self._y = list(ytids)

    def get_all_ytids(self):
        return list(self._y)


def split_fixture():
    import audio_tokens_config as ref_cfg
    from processors.dataset_splitter import DatasetSplitter
    rng = random.Random(20241016)
    alphabet = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789-_"
    ytids = ["".join(rng.choice(alphabet) for _ in range(11)) for _ in range(500)]
    cases = []
    for seed, dratio, vratio in ((4242, 0.1, 0.1), (4242, 1.0, 0.1), (7, 0.37, 0.25)):
        cfg = ref_cfg.AudioTokensConfig(random_seed=seed, dataset_ratio=dratio, validation_ratio=vratio)
        train, val = DatasetSplitter(cfg)._split_data(list(ytids))
        cases.append({"random_seed": seed, "dataset_ratio": dratio, "validation_ratio": vratio,
                      "train": train, "validation": val})
    # the document create_split_file writes (what SpectrogramGenerator.__init__ json.load()s)
    with tempfile.TemporaryDirectory() as tmp:
        cfg = ref_cfg.AudioTokensConfig(split_file=f"{tmp}/out/split.json", dataset_ratio=0.2)
        DatasetSplitter(cfg).create_split_file(_FixedYtids(ytids))
        written = Path(cfg.split_file).read_text()
    return {"source": "processors/dataset_splitter.py:20-42 (danavery/audio-tokens), imported and run",
            "ytids": ytids, "cases": cases,
            "written_file": {"dataset_ratio": 0.2, "validation_ratio": 0.1, "random_seed": 4242, "text": written}}


def set_seed_fixture():
    import numpy as np
    import torch
    from utils.set_seed import set_seed
    out = []
    for seed in (4242, 7):
        set_seed(seed)
        out.append({"seed": seed,
                    "python_random": [random.random() for _ in range(4)],
                    "numpy_rand": np.random.rand(4).tolist(),
                    "numpy_randint_1000": np.random.randint(0, 1000, size=4).tolist(),
                    "torch_rand_cpu": torch.rand(4, dtype=torch.float64).tolist(),
                    "torch_randperm_8": torch.randperm(8).tolist()})
    import numpy, torch as _t
    return {"source": "utils/set_seed.py:6-12 (danavery/audio-tokens), imported and run",
            "versions": {"numpy": numpy.__version__, "torch": _t.__version__.split("+")[0], "python": sys.version.split()[0]},
            "draws": out}


def main():
    if not REF.is_dir():
        raise SystemExit("make_reference_fixtures.py runs in the build container only (/root/reference is absent)")
    sys.path.insert(0, str(REF))
    for name, fn in (("reference_config_fields.json", config_fixture), ("reference_split.json", split_fixture),
                     ("reference_set_seed.json", set_seed_fixture)):
        (HERE / name).write_text(json.dumps(fn(), indent=1, sort_keys=False) + "\n")
        print("wrote", HERE / name)


if __name__ == "__main__":
    main()
