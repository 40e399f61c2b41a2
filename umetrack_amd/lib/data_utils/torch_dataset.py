"""Synchronous access to torch_data folders.

The reference wraps the same files in an asyncio dataset + prefetching sampler
(lib/data_utils/async_dataset.py:43-413); that plumbing is host I/O scheduling and out of scope.
What is kept is the on-disk contract: ``{root}/.../{training|testing}/{field}.torch.{idx,bin}``,
a folder counts only if it holds every requested field, all fields of a folder have the same
length, and sample ``i`` is ``{field: frame_i}``.  Bin files are memory-mapped, so a sample is a
set of zero-copy views.
"""
import os
from typing import Any, Dict, Iterable, List, Sequence, Union

import numpy as np

from .idxbinfile import BIN_SUFFIX, IDX_SUFFIX, TorchIdx
from .split import Split


def find_torchdata_folders(root: str) -> Dict[str, List[str]]:
    """Leaf folders named after a split that contain at least one .torch.bin (async_dataset.py:43-81)."""
    names = {s.value for s in Split}
    out: Dict[str, List[str]] = {}
    for path, _, files in os.walk(root, topdown=False):
        if os.path.basename(path) in names and any(f.endswith(BIN_SUFFIX) for f in files):
            out[path] = sorted(files)
    return out


class TorchDataFolder:
    """One folder, several fields (async_dataset.py:246-322 without the awaitables)."""

    def __init__(self, folder: str, fields: Iterable[str]):
        self.folder = folder
        self._idx: Dict[str, TorchIdx] = {}
        self._bin: Dict[str, np.ndarray] = {}
        self.n = None
        for field in fields:
            path = os.path.join(folder, field + IDX_SUFFIX)
            if not os.path.exists(path):
                raise ValueError(f"required field {field} not found in folder {folder}, which only has {os.listdir(folder)}")
            idx = TorchIdx(path)
            if self.n is not None and self.n != len(idx):
                raise ValueError(f"folder has mismatched field sizes: {field}")
            self.n = len(idx)
            self._idx[field] = idx
            self._bin[field] = np.memmap(idx.bin_path, dtype=np.uint8, mode="r")

    def __len__(self) -> int:
        return int(self.n or 0)

    def __getitem__(self, i: int) -> Dict[str, Any]:
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return {f: idx.view_buffer_at(i, self._bin[f]) for f, idx in self._idx.items()}

    def data_size_bytes(self) -> int:
        return sum(idx.data_size_bytes() for idx in self._idx.values())


class ConcatFolders:
    """Concatenation of folders in discovery order (torch ConcatDataset semantics)."""

    def __init__(self, parts: Sequence[TorchDataFolder]):
        self.parts = list(parts)
        self._ends = np.cumsum([len(p) for p in self.parts]) if self.parts else np.zeros(0, dtype=np.int64)

    def __len__(self) -> int:
        return int(self._ends[-1]) if len(self._ends) else 0

    def __getitem__(self, i: int):
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        k = int(np.searchsorted(self._ends, i, side="right"))
        return self.parts[k][i - (int(self._ends[k - 1]) if k else 0)]

    def __iter__(self):
        for p in self.parts:
            for i in range(len(p)):
                yield p[i]


def find_dataset(roots: Union[str, List[str]], fields: Iterable[str], splits: Iterable[Split] = Split
                 ) -> Dict[Split, ConcatFolders]:
    """async_dataset.py:325-413: {split: dataset} over every qualifying folder under ``roots``."""
    fields = list(fields)
    roots = [roots] if isinstance(roots, str) else list(roots)
    wanted = {f + ext for f in fields for ext in (IDX_SUFFIX, BIN_SUFFIX)}
    per_split: Dict[str, List[TorchDataFolder]] = {s.value: [] for s in splits}
    for root in roots:
        found = find_torchdata_folders(root)
        print(f"Found {len(found)} data folders in {root}")
        for folder, files in found.items():
            if wanted - set(files):
                continue                                  # a required field is missing: skip the folder
            name = os.path.basename(folder)
            if name in per_split:
                per_split[name].append(TorchDataFolder(folder, fields))
    return {s: ConcatFolders(per_split[s.value]) for s in splits if per_split[s.value]}
