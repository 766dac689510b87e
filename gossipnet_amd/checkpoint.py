"""Checkpoints keyed by the reference's TF variable names.

The reference saves with tf.train.Saver under `gnet-<iteration>` and links the best model as `./gnet_best`
(train.py:56-61,288,337,345).  Two on-disk formats, same keys (exactly the TF variable names,
`gnet/block3/pw_fc1/weights`, ..., `global_step`; optimizer slots `<var>/Adam`, `<var>/Adam_1`, `<var>/Momentum`):
  * `<path>.npz` (default of save());
  * the Saver V2 bundle itself, `<prefix>.index` + `<prefix>.data-00000-of-00001`, read and written WITHOUT TensorFlow
    by gossipnet_amd/tf_bundle.py -- load() picks it when `<prefix>.index` exists, save(..., fmt="tf") writes it.
    (Format restated from the TF sources; not validated against a real checkpoint: none is available here.)
"""
import os

import numpy as np


def latest_checkpoint(directory="."):
    """tf.train.get_checkpoint_state(dir).model_checkpoint_path (train.py:274-277): first line of the `checkpoint` file."""
    f = os.path.join(directory, "checkpoint")
    if not os.path.exists(f):
        return None
    for line in open(f):
        if line.startswith("model_checkpoint_path:"):
            p = line.split(":", 1)[1].strip().strip('"')
            return p if os.path.isabs(p) else os.path.join(directory, p)
    return None


def save(net, path, global_step=0, optimizer=None, fmt="npz"):
    out = {k: v for k, v in net.state_dict().items()}
    out["global_step"] = np.int64(global_step)
    if optimizer is not None:
        m = {k: t.cpu().numpy() for k, t in net.flat_to_named(optimizer.m).items()}
        v = {k: t.cpu().numpy() for k, t in net.flat_to_named(optimizer.v).items()} if optimizer.v is not None else None
        for name, _ in net._spec:
            if optimizer.kind == "adam":
                out[name + "/Adam"] = m[name]
                out[name + "/Adam_1"] = v[name]
            else:
                out[name + "/Momentum"] = m[name]
    if fmt == "tf":
        from . import tf_bundle
        tf_bundle.write_bundle(path, {k: np.asarray(v) for k, v in out.items()})
        with open(os.path.join(os.path.dirname(path) or ".", "checkpoint"), "w") as f:     # CheckpointState, as Saver.save
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (os.path.basename(path), os.path.basename(path)))
        return path
    if not path.endswith(".npz"):
        path += ".npz"
    np.savez(path, **out)
    return path


def load(net, path, optimizer=None):
    """Restores variables (and optimizer slots when present); returns global_step.  Unknown keys are
    ignored, missing variables raise KeyError -- like a TF restore of a partial checkpoint would."""
    import torch
    from . import tf_bundle

    class _Bundle(dict):
        files = property(lambda self: list(self.keys()))

    if tf_bundle.is_bundle(path):
        z = _Bundle(tf_bundle.read_bundle(path))
    else:
        z = np.load(path if path.endswith(".npz") else path + ".npz")
    missing = [n for n, _ in net._spec if n not in z.files]
    if missing:
        raise KeyError("checkpoint lacks variables: %s" % ", ".join(missing[:4]))
    net.load_params({n: z[n] for n, _ in net._spec})
    if optimizer is not None:
        slot_m = "/Adam" if optimizer.kind == "adam" else "/Momentum"
        if all((n + slot_m) in z.files for n, _ in net._spec):
            def restore(flat, suffix):
                for n, t in net.flat_to_named(flat).items():
                    t.copy_(torch.from_numpy(np.asarray(z[n + suffix], np.float32)).to(flat.device))
            restore(optimizer.m, slot_m)
            if optimizer.kind == "adam":
                restore(optimizer.v, "/Adam_1")
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
