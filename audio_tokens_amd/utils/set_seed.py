"""utils/set_seed.py of the reference (utils/set_seed.py:6-12), unchanged in behaviour: seeds python,
numpy and torch.  It does not touch the k-means RNG, which -- as in FAISS -- has its own seed 1234."""
import random

import numpy as np
import torch


def set_seed(seed=42):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
