"""Zero-copy joins of device arrays.  The batch steps hand whole planes / row blocks of one buffer from step to step
([d] in front of the l planes [beta_i]; [[zeta_1]], [[zeta_2]], [[delta_B]] as three row blocks of one encryption call);
joining them again must not cost a 0.5 GB device copy."""
from __future__ import annotations

from typing import Sequence

import torch


def cat_rows(parts: Sequence[torch.Tensor]) -> torch.Tensor:
    """torch.cat(parts, dim=0) -- as a VIEW when the parts are consecutive, contiguous blocks of one buffer (the common case:
    they were sliced from one array), as a copy otherwise.  All parts share the trailing dimensions."""
    first = parts[0]
    tail = tuple(first.shape[1:])
    if all(p.is_contiguous() and p.dtype == first.dtype and p.device == first.device and tuple(p.shape[1:]) == tail for p in parts):
        store = first.untyped_storage().data_ptr()
        offset, ok = first.storage_offset(), True
        for p in parts:
            if p.untyped_storage().data_ptr() != store or p.storage_offset() != offset:
                ok = False
                break
            offset += p.numel()
        if ok:
            rows = sum(p.shape[0] for p in parts)
            row = 1
            for d in tail:
                row *= d
            strides, acc = [], 1
            for d in reversed(tail):
                strides.append(acc)
                acc *= d
            return first.as_strided((rows,) + tail, (row,) + tuple(reversed(strides)), first.storage_offset())
    return torch.cat(list(parts), dim=0)
