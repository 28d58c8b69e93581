"""Stand-in for the two members of lightning.pytorch.Trainer the reference's processors use
(processors/vcfprocessor.py:252-265): `.precision` (read at model_combined_modulator.py:737) and
`.predict(model, dataloader, ckpt_path=None) -> list[dict]`."""
from __future__ import annotations

import torch


class Trainer:
    def __init__(self, accelerator="gpu", devices=1, logger=False, precision="bf16-mixed", enable_checkpointing=False, **kw):
        self.accelerator, self.devices, self.precision = accelerator, devices, precision

    def predict(self, model, dataloaders, ckpt_path=None):
        """Runs predict_step over the loader.  `ckpt_path` is accepted for signature parity and ignored: the
        reference re-reads the 14 GB checkpoint on every predict call (vcfprocessor.py:262); the weights
        loaded by ModelManager are already resident."""
        model.trainer = self
        model.eval()
        out = []
        with torch.no_grad():
            for i, batch in enumerate(dataloaders):
                out.append(model.predict_step(batch, i))
        return out
