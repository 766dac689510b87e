"""Checkpoints keyed by the reference's TF variable names.

The reference saves with tf.train.Saver under `gnet-<iteration>` and links the best model as `./gnet_best`
(train.py:56-61,288,337,345).  TensorFlow checkpoints cannot be read without TensorFlow; the exchange
format here is an .npz whose keys are exactly the TF variable names (`gnet/block3/pw_fc1/weights`, ...,
`global_step`), which is what `tf.train.load_variable` / a one-line export script produces on the
reference side.  Optimizer slots use TF's slot naming (`<var>/Adam`, `<var>/Adam_1`, `<var>/Momentum`).
"""
import os

import numpy as np


def save(net, path, global_step=0, optimizer=None):
    out = {k: v for k, v in net.state_dict().items()}
    out["global_step"] = np.int64(global_step)
    if optimizer is not None:
        off = 0
        m = optimizer.m.cpu().numpy()
        v = optimizer.v.cpu().numpy() if optimizer.v is not None else None
        for name, shape in net._spec:
            k = int(np.prod(shape))
            if optimizer.kind == "adam":
                out[name + "/Adam"] = m[off:off + k].reshape(shape)
                out[name + "/Adam_1"] = v[off:off + k].reshape(shape)
            else:
                out[name + "/Momentum"] = m[off:off + k].reshape(shape)
            off += k
    if not path.endswith(".npz"):
        path += ".npz"
    np.savez(path, **out)
    return path


def load(net, path, optimizer=None):
    """Restores variables (and optimizer slots when present); returns global_step.  Unknown keys are
    ignored, missing variables raise KeyError -- like a TF restore of a partial checkpoint would."""
    import torch
    z = np.load(path if path.endswith(".npz") else path + ".npz")
    missing = [n for n, _ in net._spec if n not in z.files]
    if missing:
        raise KeyError("checkpoint lacks variables: %s" % ", ".join(missing[:4]))
    net.load_params({n: z[n] for n, _ in net._spec})
    if optimizer is not None:
        slot_m = "/Adam" if optimizer.kind == "adam" else "/Momentum"
        if all((n + slot_m) in z.files for n, _ in net._spec):
            flat = np.concatenate([z[n + slot_m].reshape(-1) for n, _ in net._spec]).astype(np.float32)
            optimizer.m.copy_(torch.from_numpy(flat).to(optimizer.m.device))
            if optimizer.kind == "adam":
                flat = np.concatenate([z[n + "/Adam_1"].reshape(-1) for n, _ in net._spec]).astype(np.float32)
                optimizer.v.copy_(torch.from_numpy(flat).to(optimizer.v.device))
        optimizer.global_step = int(z["global_step"]) if "global_step" in z.files else 0
    return int(z["global_step"]) if "global_step" in z.files else 0


def checkpoint_name(iteration, directory="."):
    return os.path.join(directory, "gnet-%d" % iteration)        # train.py:337,345


def link_best(models, link="./gnet_best"):
    """ModelManager.write_link_to_best (train.py:56-61): models = [(iteration, ap, file), ...]."""
    best = max((ap, f) for _, ap, f in models)[1]
    if os.path.lexists(link):
        os.remove(link)
    os.symlink(best, link)
    return best
