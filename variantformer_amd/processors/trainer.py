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
        pipelined = (not getattr(model, "vep", False)) and hasattr(model, "predict_launch")
        with torch.no_grad():
            if not pipelined:
                for i, batch in enumerate(dataloaders):
                    out.append(model.predict_step(batch, i))
                return out
            # Software pipeline over batches: the forward of batch i is enqueued (no host wait), then batch i+1 is
            # fetched from the loader and prepared on the host (concatenation, structure arrays, H2D) while the GPU
            # works, and only then are batch i's results copied back.  Same results, same order as predict_step.
            it = iter(dataloaders)
            batch = next(it, None)
            pb = model.prepare_batch(batch) if batch is not None else None
            i = 0
            while pb is not None:
                handle = model.predict_launch(pb)
                batch = next(it, None)
                pb = model.prepare_batch(batch) if batch is not None else None
                out.append(model.predict_finish(handle, i))
                i += 1
        try:       # what the self-healing LayerNorm fold did during this pass (finished / recomputed batches, switched off?)
            from ..seq2gene.model_combined_modulator import ln_fold_state
            self.ln_fold_state = ln_fold_state(model)         # this model's own bookkeeping (per model since round 6)
        except Exception:                                     # a model class without the fold
            self.ln_fold_state = None
        return out
