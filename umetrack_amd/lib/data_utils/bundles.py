"""Recursive helpers over dataclass / NamedTuple / Mapping / tuple / list bundles.

Same public names and semantics as lib/data_utils/bundles.py:19-244 (map_fields,
to_device, collate, group, asdict, asdict_rec, field_names, is_dictlike).
"""
import dataclasses
from typing import Any, Mapping

import numpy as np
import torch


def is_dictlike(obj: Any) -> bool:
    return dataclasses.is_dataclass(obj) or hasattr(obj, "_asdict") or isinstance(obj, Mapping)


def field_names(ty):
    return [f.name for f in dataclasses.fields(ty)] if dataclasses.is_dataclass(ty) else ty._fields


def asdict(obj):
    if dataclasses.is_dataclass(obj):
        return {f.name: getattr(obj, f.name) for f in dataclasses.fields(obj)}
    if hasattr(obj, "_asdict"):
        return obj._asdict()
    if isinstance(obj, Mapping):
        return obj
    raise TypeError("asdict() requires a Mapping, dataclass, or NamedTuple")


def asdict_rec(obj):
    if is_dictlike(obj):
        return {k: asdict_rec(v) for k, v in asdict(obj).items()}
    if isinstance(obj, tuple):
        return tuple(asdict_rec(v) for v in obj)
    if isinstance(obj, list):
        return [asdict_rec(v) for v in obj]
    return obj


def map_fields(func, obj, only_type=object):
    """Apply func to every leaf that is an instance of only_type; containers keep their type."""
    def rec(o):
        if is_dictlike(o):
            items = {k: rec(v) for k, v in asdict(o).items()}
            return type(o)(items.items()) if isinstance(o, Mapping) else type(o)(**items)
        if isinstance(o, tuple):
            return tuple(rec(v) for v in o)
        if isinstance(o, list):
            return [rec(v) for v in o]
        return func(o) if isinstance(o, only_type) else o
    return rec(obj)


def to_device(obj, device):
    return map_fields(lambda t: t.to(device), obj, only_type=torch.Tensor)


def group(batch, group_fn):
    """Turn a list of N equally structured items into one item whose leaves are group_fn(list)."""
    head = batch[0]
    if isinstance(head, (np.ndarray, np.generic, torch.Tensor)):
        return group_fn(batch)
    if is_dictlike(head):
        dicts = [asdict(x) for x in batch]
        return type(head)(**{k: group([d[k] for d in dicts], group_fn) for k in dicts[0]})
    if isinstance(head, tuple):
        return tuple(group([b[i] for b in batch], group_fn) for i in range(len(head)))
    if isinstance(head, list):
        return [group([b[i] for b in batch], group_fn) for i in range(len(head))]
    if any(b is None for b in batch):
        if head is None and all(b is None for b in batch):
            return None
        raise TypeError("Some items are None and others are not")
    return batch


def collate(batch, device=None):
    """Stack arrays / tensors along a new leading dimension, field by field."""
    def stack(items):
        if isinstance(items[0], (np.ndarray, np.generic)):
            return np.stack(items)
        if isinstance(items[0], torch.Tensor):
            t = torch.stack(items)
            return t if device is None else t.to(device)
        raise TypeError(f"Can't stack tensors: unknown type {type(items[0])} found")
    return group(batch, stack)
