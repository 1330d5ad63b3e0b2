"""Checkpoint IO in the reference's on-disk format (/root/reference/train_lm.py:100-154,292-296):
`{log_dir}/{obj_name}/geomatch_{epoch:02d}.pth.tar` holding {'epoch','model_state','optimizer_state'},
copied to `{log_dir}/{obj_name}/geomatch.pth.tar`; on load a DDP 'module.' prefix is stripped."""
import os
import shutil

import torch


def checkpoint_state(model=None, optimizer=None, epoch=None):
    m = model.module if hasattr(model, "module") and isinstance(model.module, torch.nn.Module) else model
    return {"epoch": epoch,
            "model_state": m.state_dict() if m is not None else None,
            "optimizer_state": optimizer.state_dict() if optimizer is not None else None}


def save_checkpoint(model, optimizer, epoch, log_dir, obj_name, name="geomatch"):
    d = os.path.join(log_dir, obj_name)
    os.makedirs(d, exist_ok=True)
    filename = os.path.join(d, "%s_%02d.pth.tar" % (name, epoch))
    torch.save(checkpoint_state(model, optimizer, epoch), filename)
    shutil.copyfile(filename, filename[:-11] + ".pth.tar")          # train_lm.py:153-154
    return filename


def load_checkpoint(model=None, optimizer=None, filename="checkpoint", device="cpu", strict=True):
    """Returns the stored epoch, or None when the file is missing (as the reference prints and goes on)."""
    filename = "{}.pth.tar".format(filename)
    if not os.path.isfile(filename):
        print("==> Checkpoint '{}' not found".format(filename))
        return None
    ck = torch.load(filename, map_location=device, weights_only=False)
    st = ck.get("model_state")
    if model is not None and st is not None:
        if "module" in list(st.keys())[0]:
            st = {k.replace("module.", ""): v for k, v in st.items()}
        m = model.module if hasattr(model, "module") and isinstance(model.module, torch.nn.Module) else model
        m.load_state_dict(st, strict=strict)
    if optimizer is not None and ck.get("optimizer_state") is not None:
        optimizer.load_state_dict(ck["optimizer_state"])
    return ck["epoch"]
