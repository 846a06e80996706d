"""Seeding as the reference does it before every run (utils/reproducibility.py:6-24)."""
import random

import numpy as np
import torch


def set_seed_and_cudnn(seed_value=42):
    random.seed(seed_value)
    np.random.seed(seed_value)
    torch.manual_seed(seed_value)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed_value)
    # the HIP kernels of this engine are deterministic by construction (no float atomics)
