"""Reader (and a small writer) for the ``.torch.idx`` / ``.torch.bin`` pair.

Same attribute and method names as ``TorchIdx`` in lib/data_utils/idxbinfile.py:48-347, so code
written against the reference keeps working.  The layout, from idxbinfile.py:111-121: the index
file is a flat int64 array

    [magic 0x584449544E54, version 1, dtype code, itemsize, N, S,
     N+1 dim offsets, N+1 data offsets (units of itemsize), S sizes]

(version 0 files carry magic 0), and the bin file is the frames back to back.  dtype codes 1..7
are uint8, int8, int16, int32, int64, float32, float64; code 8 frames are msgpack documents
(idxbinfile.py:368-377).

Pure host I/O: nothing here touches the GPU.  Frames are returned as zero-copy views of the
buffer handed in (a ``numpy.memmap`` in ``TorchDataFolder``), so a frame reaches HBM with one
host-to-device copy.
"""
import math
import os
from enum import Enum
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import msgpack
import numpy as np

Buffer = Union[bytes, bytearray, memoryview, np.ndarray]

IDX_MAGIC = 0x584449544E54
IDX_SUFFIX, BIN_SUFFIX = ".torch.idx", ".torch.bin"


class BinFormat(Enum):
    TENSOR = 0
    MSGPACK = 1


_CODE_TO_TYPE: Dict[int, Tuple[str, BinFormat]] = {
    1: ("uint8", BinFormat.TENSOR), 2: ("int8", BinFormat.TENSOR), 3: ("int16", BinFormat.TENSOR),
    4: ("int32", BinFormat.TENSOR), 5: ("int64", BinFormat.TENSOR), 6: ("float32", BinFormat.TENSOR),
    7: ("float64", BinFormat.TENSOR), 8: ("object", BinFormat.MSGPACK),
}
_TYPE_TO_CODE = {name: code for code, (name, _) in _CODE_TO_TYPE.items()}
OBJECT_DTYPE = np.dtype("object")


def _bin_path_for(idx_path: str) -> str:
    if not idx_path.endswith(IDX_SUFFIX):
        raise AssertionError(f"{idx_path} does not end with {IDX_SUFFIX}")      # idxbinfile.py:41
    return idx_path[: -len(IDX_SUFFIX)] + BIN_SUFFIX


def _as_int64(buffer: Buffer, path: str) -> np.ndarray:
    """idxbinfile.py:329-347: bytes whose length is a multiple of 8, or an int64 array."""
    view = memoryview(buffer)
    if view.ndim != 1:
        raise ValueError(f".torch.idx data has invalid shape: {view.shape}; require ndim=1")
    if view.format == "B":
        if view.nbytes % 8:
            raise ValueError(f".torch.idx data has invalid length: {view.nbytes}%8 != 0")
    elif view.format not in ("q", "l") or view.itemsize != 8:
        raise ValueError(f".torch.idx data has invalid format {view.format}: expected 'B' (bytes)  or 'q' or 'l' (int64)")
    return np.frombuffer(view, dtype=np.int64)


class TorchIdx:
    """Parsed ``.torch.idx`` file: ``len(idx)`` frames, ``idx.dims[i]`` the shape of frame ``i``,
    ``idx.shape`` = ``(N, *frame_shape)`` when every frame has the same shape, else ``None``."""

    def __init__(self, path: str, bin_path: Optional[str] = None, buffer: Optional[Buffer] = None) -> None:
        self.source = path
        self.bin_path = _bin_path_for(path) if bin_path is None else bin_path
        if buffer is None:
            with open(path, "rb") as fh:
                buffer = fh.read()
        words = _as_int64(buffer, path)
        if words.size < 6:
            raise ValueError(f".torch.idx file {path} is too short ({words.size} words)")
        magic, version = int(words[0]), int(words[1])
        if version not in (0, 1):
            raise ValueError(f"unsupported version {version} in .torch.idx file {path}")
        if magic != (IDX_MAGIC if version == 1 else 0):
            raise ValueError(f"bad magic number in .torch.idx file {path}")
        code = int(words[2])
        self.itemsize = int(words[3])
        if code not in _CODE_TO_TYPE:
            raise KeyError(f"unrecognized type {code}")
        type_name, self._bin_format = _CODE_TO_TYPE[code]
        self.dtype = np.dtype(type_name)
        if self.dtype != OBJECT_DTYPE and self.dtype.itemsize != self.itemsize:
            raise ValueError(f"item size {self.itemsize} not compatible with dtype {type_name}.itemsize={self.dtype.itemsize}")
        n, s = int(words[4]), int(words[5])
        need = 6 + 2 * (n + 1) + s
        if words.size < need:
            raise ValueError(f".torch.idx file {path} is truncated: {words.size} words, header promises {need}")
        dim_off = words[6:6 + n + 1]
        data_off = words[6 + n + 1:6 + 2 * (n + 1)]
        sizes = words[6 + 2 * (n + 1):need]
        dims: Union[np.ndarray, List[Tuple[int, ...]]] = [
            tuple(int(v) for v in sizes[dim_off[i]:dim_off[i + 1]]) for i in range(n)]
        self.is_uniform = False
        self.shape: Optional[Tuple[int, ...]] = None
        self.byte_offsets_array: Optional[np.ndarray] = data_off * self.itemsize
        if n and self.dtype != OBJECT_DTYPE and len({len(d) for d in dims}) == 1:
            first = dims[0]
            if all(d == first for d in dims):
                # one frame shape for the whole file: keep a broadcast view instead of N copies, and
                # derive offsets arithmetically when the data starts at 0 (idxbinfile.py:176-187)
                self.is_uniform = True
                self.shape = (n, *first)
                if not np.all(np.diff(data_off) == math.prod(first)):
                    raise AssertionError(f"{path}: data offsets disagree with the uniform frame shape {first}")
                if data_off[0] == 0:
                    self.byte_offsets_array = None
                dims = np.broadcast_to(np.array(first, dtype=np.int64), (n, len(first)))
            else:
                dims = np.array(dims, dtype=np.int64)
        self.dims = dims

    def __len__(self) -> int:
        return len(self.dims)

    def __repr__(self) -> str:
        if self.is_uniform:
            return f"TorchIdx(shape={self.shape}, dtype={self.dtype}, source={self.source})"
        return f"TorchIdx(N={len(self)}, dtype={self.dtype}, is_uniform=False, source={self.source})"

    # ---- offsets (byte positions in the .torch.bin file) ----
    def _frame_bytes(self) -> int:
        assert self.shape is not None
        return math.prod(self.shape[1:]) * self.itemsize

    def byte_offset(self, i: int) -> int:
        """Start of frame ``i``; ``-1`` (or N) is the end of the data."""
        if self.byte_offsets_array is not None:
            return int(self.byte_offsets_array[i])
        n = self.shape[0]
        i = n if i == -1 else i
        assert i <= n
        return self._frame_bytes() * i

    def byte_offsets(self, start: int, end: int) -> np.ndarray:
        if self.byte_offsets_array is not None:
            return self.byte_offsets_array[start:end]
        n = self.shape[0]
        end = n if end == -1 else end
        assert start <= n and end <= n + 1
        return self._frame_bytes() * np.arange(start, end, dtype=np.ulonglong)

    def data_size_bytes(self) -> int:
        return self.byte_offset(-1) - self.byte_offset(0)

    def frame_size_bytes(self) -> int:
        if not self.is_uniform:
            raise ValueError("frame_size_bytes requires that all frames are the same size")
        return self.byte_offset(1) - self.byte_offset(0)

    def item_shape(self, i: Optional[int] = None) -> Tuple[int, ...]:
        if i is None:
            if self.shape is None:
                raise ValueError(f"Dataset does not have uniform shape: {self.source}")
            return tuple(self.shape[1:])
        return tuple(int(v) for v in self.dims[i])

    # ---- data ----
    def _check_size(self, istart: int, istop: int, buffer: Buffer) -> None:
        have = memoryview(buffer).nbytes
        want = self.byte_offset(istop) - self.byte_offset(istart)
        if have != want:
            raise ValueError(f"expected {want} bytes but got {have} for {self.bin_path}[{istart}:{istop}]")

    def view_frame(self, index: int, buffer: Buffer):
        """One frame from exactly its bytes: an ndarray view, or the decoded msgpack document."""
        self._check_size(index, index + 1, buffer)
        if self._bin_format == BinFormat.MSGPACK:
            return msgpack.unpackb(bytes(buffer) if isinstance(buffer, np.ndarray) else buffer)
        return np.ndarray(shape=self.item_shape(index), dtype=self.dtype, buffer=buffer)

    def view_buffer_at(self, index: int, buffer: Buffer):
        """Frame ``index`` out of a buffer that holds the whole bin file."""
        lo, hi = (int(v) for v in self.byte_offsets(index, index + 2))
        return self.view_frame(index, memoryview(buffer)[lo:hi])

    def view_buffer(self, buffer: Buffer):
        """The whole file: one ``(N, ...)`` array when uniform, else a lazy list of frames."""
        self._check_size(0, -1, buffer)
        if self.shape is not None:
            return np.ndarray(shape=self.shape, dtype=self.dtype, buffer=buffer)
        return _LazyFrames(self, buffer)

    def read_bin(self):
        lo, hi = self.byte_offset(0), self.byte_offset(-1)
        with open(self.bin_path, "rb") as fh:
            fh.seek(lo)
            return self.view_buffer(fh.read(hi - lo))


class _LazyFrames(Sequence):
    def __init__(self, idx: TorchIdx, buffer: Buffer):
        self._idx, self._buffer = idx, buffer

    def __len__(self) -> int:
        return len(self._idx)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._idx.view_buffer_at(i, self._buffer)


def write_torch_bin(idx_path: str, frames: Sequence[Any], dtype: Optional[str] = None) -> None:
    """Write ``frames`` as a version-1 idx/bin pair (used for fixtures and synthetic datasets; the
    reference ships no writer).  ``dtype='object'`` packs each frame with msgpack; otherwise every
    frame is an array of ``dtype`` (default: the first frame's)."""
    bin_path = _bin_path_for(idx_path)
    if dtype is None:
        dtype = "object" if not isinstance(frames[0], np.ndarray) else str(frames[0].dtype)
    code = _TYPE_TO_CODE[dtype]
    dim_off, data_off, sizes = [0], [0], []
    with open(bin_path, "wb") as fh:
        for f in frames:
            if dtype == "object":
                blob = msgpack.packb(f)
                sizes.append(len(blob))
                itemsize = 1
            else:
                arr = np.ascontiguousarray(f, dtype=dtype)
                blob = arr.tobytes()
                sizes.extend(arr.shape)
                itemsize = arr.dtype.itemsize
            fh.write(blob)
            dim_off.append(len(sizes))
            data_off.append(data_off[-1] + len(blob) // itemsize)
    itemsize = 1 if dtype == "object" else np.dtype(dtype).itemsize
    header = [IDX_MAGIC, 1, code, itemsize, len(frames), len(sizes)]
    np.array(header + dim_off + data_off + sizes, dtype=np.int64).tofile(idx_path)
