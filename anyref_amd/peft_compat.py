"""Stand-in for the two `peft` calls the reference's callers make (`eval_referseg.py:84-85`,
`eval_avs_object.py:78-79`, `merge_lora.py:58-59`):

    model = PeftModel.from_pretrained(model, lora_name)
    model = model.merge_and_unload()

`peft` walks `nn.Module` trees; the MI355X backend keeps its weights in HBM behind a C-ABI handle, so the same
two lines are served here by merging the adapter into the host-side state dict before the handle is built
(`anyref_amd.checkpoint.merge_lora`: W += (lora_alpha / r) * B @ A on `q_proj` / `v_proj`, `modules_to_save`
overrides -- `train.py:371-396`).  A caller switches with one import:  `from anyref_amd.peft_compat import PeftModel`.
"""
from __future__ import annotations


class PeftModel:
    def __init__(self, model, adapter_dir: str):
        self.base_model = model
        self.adapter_dir = adapter_dir

    @classmethod
    def from_pretrained(cls, model, model_id: str, **_kw) -> "PeftModel":
        return cls(model, model_id)

    def merge_and_unload(self):
        self.base_model.merge_adapter(self.adapter_dir)
        return self.base_model

    def __getattr__(self, name):            # `.eval()`, `.generate(...)` straight on the un-merged wrapper
        return getattr(self.base_model, name)
