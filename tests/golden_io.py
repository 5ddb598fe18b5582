"""Load tests/golden/*.npz back into torch tensors (bit patterns -> dtypes)."""
from pathlib import Path

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"


def load(name: str):
    return np.load(GOLDEN / f"{name}.npz")


def bf16(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)


def fp8(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.copy()).view(torch.float8_e4m3fn)


def i32(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.astype(np.int32, copy=True))


def i64(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.astype(np.int64, copy=True))


def f32(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.astype(np.float32, copy=True))
