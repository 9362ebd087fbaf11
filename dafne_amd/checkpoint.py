"""Weight loading for the engine.

Reference: `DetectionCheckpointer(model).resume_or_load(cfg.MODEL.WEIGHTS)`
(tools/plain_train_net.py:577-579) reads detectron2 `.pth` checkpoints
({"model": state_dict, ...}) whose parameter names are the ones this engine's
modules use, and `.pkl` model-zoo files (pickled {"model": {name:
ndarray}}).  MSRA `R-50.pkl` / `R-101.pkl` trunks use Caffe2 names; they are
mapped to `backbone.bottom_up.*` with the rules of detectron2's
`c2_model_loading.convert_basic_c2_names` [recalled, SURVEY appendix B].

Loading is host-side plumbing: tensors land in the fp32 parameter containers;
`model.invalidate()` re-packs the bf16 engine weights on the next forward.
"""
import pickle
import re

import numpy as np
import torch


def _c2_to_d2(name):
    """Caffe2 ResNet trunk name -> detectron2 name (without the backbone prefix), or None."""
    n = name.replace("_", ".")
    n = re.sub(r"\.b$", ".bias", n)
    n = re.sub(r"\.w$", ".weight", n)
    n = re.sub(r"bn\.s$", "norm.weight", n)
    n = re.sub(r"bn\.bias$", "norm.bias", n)
    n = re.sub(r"bn\.rm", "norm.running_mean", n)
    n = re.sub(r"bn\.running.mean$", "norm.running_mean", n)
    n = re.sub(r"bn\.riv$", "norm.running_var", n)
    n = re.sub(r"bn\.running.var$", "norm.running_var", n)
    n = re.sub(r"bn\.gamma$", "norm.weight", n)
    n = re.sub(r"bn\.beta$", "norm.bias", n)
    n = re.sub(r"^res\.conv1\.norm\.", "conv1.norm.", n)
    n = re.sub(r"^conv1\.", "stem.conv1.", n)
    n = n.replace(".branch1.", ".shortcut.")
    n = n.replace(".branch2a.", ".conv1.").replace(".branch2b.", ".conv2.").replace(".branch2c.", ".conv3.")
    n = re.sub(r"^(res\d)\.(\d+)\.", r"\1.\2.", n)
    if n.startswith(("fc1000", "pred")):
        return None
    return n


def load_checkpoint_file(path):
    """-> flat {name: torch.Tensor} from a .pth (torch) or .pkl (detectron2 / MSRA) file."""
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1")
        sd = data["model"] if isinstance(data, dict) and "model" in data else data
        if isinstance(data, dict) and data.get("matching_heuristics") or any("branch2a" in k or k.endswith("_w") for k in sd):
            conv = {}
            for k, v in sd.items():
                nk = _c2_to_d2(k)
                if nk is not None:
                    conv["backbone.bottom_up." + nk] = v
            sd = conv
        return {k: torch.from_numpy(np.asarray(v)).clone() if not isinstance(v, torch.Tensor) else v
                for k, v in sd.items()}
    data = torch.load(path, map_location="cpu", weights_only=False)
    sd = data["model"] if isinstance(data, dict) and "model" in data else data
    return {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))) for k, v in sd.items()}


FP8_SCALES_KEY = "fp8_act_scales"       # engine-specific entry beside "model" in a .pth checkpoint


def load_checkpoint_with_scales(path):
    """ONE read of the file -> (state dict, fp8 activation scales or None): the scales an fp8 model was calibrated with travel
    beside "model" in the .pth ({weight key: in_qscale})."""
    if path.endswith(".pkl"):
        return load_checkpoint_file(path), None
    data = torch.load(path, map_location="cpu", weights_only=False)
    sd = data["model"] if isinstance(data, dict) and "model" in data else data
    sd = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))) for k, v in sd.items()}
    sc = data.get(FP8_SCALES_KEY) if isinstance(data, dict) else None
    return sd, (None if sc is None else {str(k): float(v) for k, v in dict(sc).items()})


def load_fp8_act_scales(path):
    """The fp8 model's calibrated activation scales stored beside the weights ({weight key: in_qscale}), or None."""
    if not isinstance(path, str) or path.endswith(".pkl"):
        return None
    return load_checkpoint_with_scales(path)[1]


def save_checkpoint(model, path):
    """{"model": state_dict} in detectron2's .pth layout; an fp8 model's calibrated activation scales travel with the
    weights (they are part of the model: ADVICE round 2), so a served model never depends on what it sees first."""
    data = {"model": {k: v.detach().cpu() for k, v in model.state_dict().items()}}
    sc = model.fp8_act_scales() if hasattr(model, "fp8_act_scales") else None
    if sc is not None:
        data[FP8_SCALES_KEY] = dict(sc)
    torch.save(data, path)


def load_weights(model, path_or_state, strict=False):
    """Copy matching tensors into `model`; returns (missing, unexpected) key lists.
    Caffe2-style BGR stems etc. are taken as they are (the released DAFNe configs use
    INPUT.FORMAT BGR with d2's ImageNet trunks)."""
    sc = None
    if isinstance(path_or_state, str):
        sd, sc = load_checkpoint_with_scales(path_or_state)          # one read: weights and the scales calibrated for them
    else:
        sd = dict(path_or_state)
    own = model.state_dict()
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own]
    bad = [k for k in own if k in sd and tuple(sd[k].shape) != tuple(own[k].shape)]
    if bad:
        raise ValueError("shape mismatch for %s" % bad[:5])
    if strict and (missing or unexpected):
        raise KeyError("missing %s unexpected %s" % (missing[:5], unexpected[:5]))
    takes_scales = sc is not None and getattr(getattr(model, "cfg", None), "ENGINE", None) is not None \
        and model.cfg.ENGINE.WEIGHT_DTYPE == "fp8_e4m3" and model.device.type == "cuda"
    if takes_scales:
        # everything that can be refused is refused BEFORE the model is touched (a failure used to leave it half-loaded:
        # new weights, no scales)
        if missing:
            # the scales were calibrated for the checkpoint's weights: a partial load serves other weights
            raise ValueError("fp8 activation scales in the checkpoint, but %d model keys are missing from it: the scales belong to "
                             "ITS weights (load them explicitly with set_fp8_act_scales if that is intended)" % len(missing))
        from . import engine
        engine.check_act_qscales(sc)
        model.check_fp8_act_scale_keys(sc)     # the layer set too (it follows from the architecture, not from the values)
    with torch.no_grad():
        for k, v in own.items():
            if k in sd:
                v.copy_(sd[k].to(v.dtype))
    if hasattr(model, "invalidate"):
        model.invalidate()
    if takes_scales:
        model.set_fp8_act_scales(sc)
    return missing, unexpected
