"""``**/*.pt`` prompt-embedding loader (file format of the reference:
``ecad/dataset_utils/prompt_embedding_dataset.py:10-60`` -- one dict of tensors per prompt,
each tensor squeezed, plus ``name`` and ``relative_path`` entries)."""
from __future__ import annotations

from pathlib import Path

import torch
from torch.utils.data import Dataset


class PromptEmbeddingDataset(Dataset):
    def __init__(self, embedding_dir: Path):
        self.root = Path(embedding_dir)
        self.files = sorted(self.root.glob("**/*.pt"))
        if not self.files:
            raise FileNotFoundError(f"no .pt prompt embeddings under {self.root}")

    def __len__(self) -> int:
        return len(self.files)

    def __getitem__(self, idx: int) -> dict:
        path = self.files[idx]
        data = torch.load(path, map_location="cpu", weights_only=True)
        item = {k: (v.squeeze() if torch.is_tensor(v) else v) for k, v in data.items()}
        item["name"] = path.stem
        rel = path.parent.relative_to(self.root)
        item["relative_path"] = str(rel) if str(rel) != "." else ""
        return item
