"""Prompt cache.  The reference runs the (frozen) CLIP text encoder on the category prompts of EVERY batch
(DiFashion/models/difashion.py:218-224 training, :340-353 sampling), although the prompts are a closed set: one sentence per
category (``data_utils.py:96-111``) plus the empty "null" prompt.  ``PromptTable`` encodes that set ONCE with whatever text
encoder the caller owns (the encoder itself is out of scope here: any callable ``input_ids -> (hidden_states, ...)``) and keeps
the (n_categories + 1, 77, D) table in HBM; the per-batch work becomes an index_select on the device."""
from __future__ import annotations

from typing import Dict, Sequence

import torch

from .data import category_prompt


class PromptTable:
    def __init__(self, table: torch.Tensor, cate_ids: Sequence[int]):
        self.table = table                                   # [len(cate_ids) + 1][T][D]; the last row is the null prompt
        self._row = {int(c): i for i, c in enumerate(cate_ids)}
        self._lut = None

    @classmethod
    @torch.no_grad()
    def build(cls, text_encoder, tokenizer, id_cate_dict: Dict[int, str], device, batch_size: int = 64) -> "PromptTable":
        cate_ids = sorted(id_cate_dict)
        prompts = [category_prompt(id_cate_dict[c]) for c in cate_ids] + [""]
        rows = []
        for s in range(0, len(prompts), batch_size):
            ids = tokenizer(prompts[s:s + batch_size], padding="max_length", max_length=tokenizer.model_max_length,
                            truncation=True, return_tensors="pt").input_ids
            rows.append(text_encoder(ids.to(device))[0].to(device))
        return cls(torch.cat(rows, dim=0).contiguous(), cate_ids)

    @property
    def null_prompt(self) -> torch.Tensor:                   # (1, T, D), what text_encoder(tokenizer([""]))[0] returns
        return self.table[-1:]

    def lookup(self, category_ids: torch.Tensor) -> torch.Tensor:
        """(...,) category ids -> (N, T, D) encoder_hidden_states in row-major order of the ids (device gather, no host sync)."""
        if self._lut is None or self._lut.device != self.table.device:
            lut = torch.full((max(self._row) + 1,), -1, dtype=torch.long)
            for c, i in self._row.items():
                lut[c] = i
            self._lut = lut.to(self.table.device)
        rows = self._lut[category_ids.reshape(-1).to(self.table.device).long()]
        return self.table.index_select(0, rows)

    def to(self, device):
        self.table = self.table.to(device)
        self._lut = None
        return self
