"""Inference wrapper and weight-format helpers ("next" row f3 of SURVEY section 8).

`enhance` restates what the reference's single-image scripts do around the model (eval_hf.py:44-58, app.py:33-53,
eval.py:60-66): reflect-pad the bottom / right edge up to a multiple of 8, `model(input ** gamma)` with the HVI
post-scales `trans.alpha_s` / `trans.alpha` set from the caller, clamp to [0, 1], crop back.  `load_weights` /
`save_pretrained` cover the two checkpoint formats the reference loads: a pickled `state_dict` (.pth, train.py:93-101,
eval.py:42, strict) and the Hugging Face layout `model.safetensors` + `config.json` (eval_hf.py:21-35, strict=False there).
The pad / power / clamp / crop are plain tensor plumbing on the caller's device; the model call is the HIP path."""
import json
import os

import torch
import torch.nn.functional as F


def pad_to_multiple(x, factor=8):
    """(B,3,h,w) -> reflect-padded (B,3,H,W) and (h, w).  Exactly the reference's arithmetic: H = ((h + f) // f) * f and the
    pad is applied only when h % f != 0 (eval_hf.py:47-51)."""
    h, w = x.shape[-2], x.shape[-1]
    H, W = ((h + factor) // factor) * factor, ((w + factor) // factor) * factor
    padh = H - h if h % factor != 0 else 0
    padw = W - w if w % factor != 0 else 0
    if padh or padw:
        x = F.pad(x, (0, padw, 0, padh), "reflect")
    return x, (h, w)


@torch.no_grad()
def enhance(model, img, gamma=1.0, alpha_s=1.0, alpha_i=1.0):
    """img: (3,h,w) or (B,3,h,w) float in [0,1] on the model's device.  Returns the enhanced image(s), same shape."""
    squeeze = img.dim() == 3
    x = img.unsqueeze(0) if squeeze else img
    x, (h, w) = pad_to_multiple(x, 8)
    was_training = model.training
    model.eval()
    model.trans.alpha_s = alpha_s
    model.trans.alpha = alpha_i
    out = model(x ** gamma)
    if isinstance(out, tuple):                       # CIDNet_TNSM returns (rgb, noise) in training mode only
        out = out[0]
    out = torch.clamp(out, 0, 1)[:, :, :h, :w]
    model.train(was_training)
    return out.squeeze(0) if squeeze else out


def load_weights(model, path, strict=None):
    """.pth / .pt: pickled state_dict, strict (train.py:99, eval.py:42);  .safetensors (or a directory holding
    model.safetensors): strict=False as eval_hf.py:33 does.  Returns the (missing, unexpected) key lists."""
    if os.path.isdir(path):
        path = os.path.join(path, "model.safetensors")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
        res = model.load_state_dict(sd, strict=False if strict is None else strict)
    else:
        sd = torch.load(path, map_location="cpu")
        res = model.load_state_dict(sd, strict=True if strict is None else strict)
    return list(res.missing_keys), list(res.unexpected_keys)


def save_pretrained(model, directory, config=None):
    """Hugging Face layout: model.safetensors (the state_dict, reference key names) + config.json (the constructor
    arguments, which is what PyTorchModelHubMixin stores for the reference class, net/CIDNet.py:6-13)."""
    from safetensors.torch import save_file
    os.makedirs(directory, exist_ok=True)
    sd = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    save_file(sd, os.path.join(directory, "model.safetensors"))
    if config is None:
        config = {"channels": [36, 36, 72, 144], "heads": [1, 2, 4, 8], "norm": False}
    with open(os.path.join(directory, "config.json"), "w", encoding="utf-8") as f:
        json.dump(config, f)
    return directory
