"""Checkpoint and prototype files in the reference's formats.

Reference: VL-T5/src/trainer_base.py:246-269 (`save` / `load`: `torch.save(model.state_dict())` into `{output}/{name}.pth`, where
the model is DDP-wrapped so every key starts with `module.`; `load` renames the legacy `vis_encoder.` prefix and loads with
`strict=False`), src/vqacl.py:419-424, 540-542 (`Q_prototype.pt` / `V_prototype.pt`: the two prototype tensors saved whole).

The engine-backed model keeps the reference's parameter names, so a reference checkpoint loads as is and a checkpoint written
here loads into the reference.
"""
import os

import torch


def to_reference_keys(state_dict, prefix="module."):
    """Keys as the reference writes them from its DDP-wrapped model."""
    return {(k if k.startswith(prefix) else prefix + k): v for k, v in state_dict.items()}


def from_reference_keys(state_dict):
    """Undo the DDP prefix and the legacy renames of `TrainerBase.load` (trainer_base.py:256-264): `module.vis_encoder.*` and
    `module.model.vis_encoder.*` become `encoder.*`; `module.` / `module.model.` are stripped."""
    out = {}
    for k, v in state_dict.items():
        for p in ("module.model.", "module."):
            if k.startswith(p):
                k = k[len(p):]
                break
        if k.startswith("vis_encoder."):
            k = "encoder." + k[len("vis_encoder."):]
        out[k] = v
    return out


def save_checkpoint(model, output, name, ddp_prefix=True):
    """`{output}/{name}.pth` (name = `{task}_LAST` after every task, vqacl.py:416)."""
    os.makedirs(output, exist_ok=True)
    sd = {k: v.detach().to("cpu") for k, v in model.state_dict().items()}
    path = os.path.join(output, f"{name}.pth")
    torch.save(to_reference_keys(sd) if ddp_prefix else sd, path)
    return path


def load_checkpoint(model, path, map_location="cpu"):
    """Loads `{path}.pth` (the reference passes the path without extension) with strict=False; returns the missing/unexpected
    key report like `load_state_dict`."""
    if not path.endswith(".pth"):
        path = path + ".pth"
    sd = torch.load(path, map_location=map_location)
    return model.load_state_dict(from_reference_keys(sd), strict=False)


def save_prototypes(model, output):
    """`Q_prototype.pt` [n_ques, d] and `V_prototype.pt` [n_cate, d] (vqacl.py:419-424)."""
    os.makedirs(output, exist_ok=True)
    torch.save(model.Q_prototype.detach().to("cpu"), os.path.join(output, "Q_prototype.pt"))
    torch.save(model.V_prototype.detach().to("cpu"), os.path.join(output, "V_prototype.pt"))


def load_prototypes(model, output, map_location="cpu"):
    """vqacl.py:540-542: assigned through the model's attribute setters."""
    model.Q_prototype = torch.load(os.path.join(output, "Q_prototype.pt"), map_location=map_location)
    model.V_prototype = torch.load(os.path.join(output, "V_prototype.pt"), map_location=map_location)
