"""On-disk formats of the reference's data pipeline, host side only.

EnCodec and g2p_en -- the third-party front-ends that *produce* these files -- are not part of this build; what
they write is:

  * `<name>.qnt.pt`   `torch.save` of the EnCodec codes, int64 `[1, n_q = 8, t]`
                      (/root/reference/vall_e/emb/qnt.py:68,93); the models consume the `(t, n_q)` transpose
                      (/root/reference/vall_e/data.py:31-37 `_load_quants`);
  * `<name>.phn.txt`  space-separated phone symbols; the loader wraps them in `<s> ... </s>`
                      (/root/reference/vall_e/data.py:40-45 `_get_phones`);
  * phone symmap      `{symbol: id}` with ids starting at 1 so that 0 is padding, built from the sorted set of
                      all phones of the corpus (/root/reference/vall_e/data.py:125-127); the trained AR module
                      carries it as `ar.phone_symmap` (/root/reference/vall_e/__main__.py:49).
"""
from __future__ import annotations

import json
from pathlib import Path

import torch

BOS, EOS = "<s>", "</s>"


def load_quants(path) -> torch.Tensor:
    """`.qnt.pt` -> int64 (t, n_q), as data.py:_load_quants."""
    q = torch.load(Path(path), map_location="cpu")
    if q.dim() != 3 or q.shape[0] != 1:
        raise ValueError(f"{path}: expected an int64 [1, n_q, t] tensor, got {tuple(q.shape)}")
    return q[0].t().contiguous().long()


def save_quants(codes: torch.Tensor, path) -> None:
    """(t, n_q) codes -> `.qnt.pt` in the layout EnCodec's decoder side reads (`[1, n_q, t]` int64;
    qnt.py:decode_to_file rearranges "t q -> 1 q t" itself, so this is the same file encode() writes)."""
    if codes.dim() == 1:
        codes = codes.unsqueeze(-1)
    torch.save(codes.t().contiguous().long().unsqueeze(0).cpu(), Path(path))


def read_phones(path) -> list[str]:
    """`.phn.txt` -> ['<s>', ..., '</s>'] as data.py:_get_phones."""
    return [BOS] + Path(path).read_text(encoding="utf8").split() + [EOS]


def build_symmap(phone_lists) -> dict[str, int]:
    """Sorted set of all symbols, ids from 1 (0 = padding) -- data.py:_get_phone_symmap."""
    return {s: i for i, s in enumerate(sorted({p for phones in phone_lists for p in phones}), 1)}


def load_symmap(path) -> dict[str, int]:
    """A symmap saved as JSON ({symbol: id}); upstream keeps it inside the pickled module, which needs upstream's
    classes to load -- export it once there with `json.dump(ar.phone_symmap, f)`."""
    m = json.loads(Path(path).read_text(encoding="utf8"))
    if not all(isinstance(v, int) and v >= 1 for v in m.values()):
        raise ValueError(f"{path}: symmap ids must be integers >= 1")
    return m


def phones_to_ids(phones, symmap) -> torch.Tensor:
    """int64 ids; an unknown symbol is an error (upstream's `map(symmap.get, ...)` would put a None in the list
    and fail inside torch.tensor with a less helpful message)."""
    missing = [p for p in phones if p not in symmap]
    if missing:
        raise KeyError(f"phones not in the symmap: {sorted(set(missing))}")
    return torch.tensor([symmap[p] for p in phones], dtype=torch.long)
