"""Checkpoint compatibility with the reference (wetectron/utils/checkpoint.py:41-104,156-178 and
utils/model_serialization.py:11-80): `.pth` files hold {"model": state_dict, "optimizer": ..., "scheduler": ...,
"iteration": ...}; keys may carry DistributedDataParallel's "module." prefix; pretrained backbones carry SHORTER
keys that are matched to the model's keys by longest suffix.  Parameter and buffer names of this build equal the
reference's, so its checkpoints load unchanged (and ours load into wetectron)."""
from collections import OrderedDict

import torch


def strip_prefix_if_present(state_dict, prefix="module."):
    keys = sorted(state_dict.keys())
    if not keys or not all(k.startswith(prefix) for k in keys):
        return state_dict
    return OrderedDict((k.replace(prefix, ""), v) for k, v in state_dict.items())


def align_and_update_state_dicts(model_state_dict, loaded_state_dict):
    """For every model key take the loaded key that is its LONGEST suffix (model_serialization.py:11-62)."""
    loaded_keys = sorted(loaded_state_dict.keys())
    matched = {}
    for key in sorted(model_state_dict.keys()):
        best = None
        for lk in loaded_keys:
            if key.endswith(lk) and (best is None or len(lk) > len(best)):
                best = lk
        if best is not None:
            model_state_dict[key] = loaded_state_dict[best]
            matched[key] = best
    return matched


def load_state_dict(model, loaded_state_dict):
    """model_serialization.py:73-80.  Parameters that live in engine.FlatSGD's flat buffers are updated in place
    (copy_ into the views); call FlatSGD.sync_from_params() afterwards to refresh the bf16 shadows."""
    model_state_dict = model.state_dict()
    loaded_state_dict = strip_prefix_if_present(loaded_state_dict, prefix="module.")
    matched = align_and_update_state_dicts(model_state_dict, loaded_state_dict)
    model.load_state_dict(model_state_dict)
    return matched


def load_checkpoint(model, path, map_location="cpu"):
    """DetectronCheckpointer._load_file + _load_model for native `.pth` files (checkpoint.py:169-178): a bare
    state-dict is wrapped as {"model": ...}.  Returns the rest of the checkpoint (optimizer, scheduler, iteration)."""
    loaded = torch.load(path, map_location=map_location)
    if "model" not in loaded:
        loaded = dict(model=loaded)
    load_state_dict(model, loaded.pop("model"))
    return loaded


def save_checkpoint(model, path, optimizer=None, iteration=None, **extra):
    """Checkpointer.save (checkpoint.py:41-63): {"model", "optimizer", "scheduler", "iteration"} in the reference's
    layout, loadable by wetectron.  `optimizer` = engine.FlatSGD (its momenta are written as a torch.optim.SGD
    state_dict, one group per trainable parameter)."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()        # the optimiser may still be rewriting the head on its side stream (engine.FlatSGD.step)
    data = {"model": model.state_dict()}
    if optimizer is not None:
        data["optimizer"] = optimizer.state_dict(model)
        data["scheduler"] = optimizer.scheduler_state(iteration or 0)
    if iteration is not None:
        data["iteration"] = int(iteration)
    data.update(extra)
    torch.save(data, path)


def last_checkpoint(output_dir):
    """Checkpointer.has_checkpoint / get_checkpoint_file (checkpoint.py:106-123): the path recorded in
    OUTPUT_DIR/last_checkpoint, or None."""
    import os
    tag = os.path.join(output_dir, "last_checkpoint") if output_dir else ""
    if not tag or not os.path.exists(tag):
        return None
    with open(tag) as f:
        path = f.read().strip()
    return path if path and os.path.exists(path) else None


def tag_last_checkpoint(output_dir, path):
    import os
    with open(os.path.join(output_dir, "last_checkpoint"), "w") as f:
        f.write(path)


def restore_training_state(optimizer, model, rest):
    """Apply what load_checkpoint returned to an engine.FlatSGD: momenta, schedule position; returns the iteration
    to continue from.  Without this a resumed run silently restarts with zero momentum."""
    iteration = int(rest.get("iteration", 0) or 0)
    if rest.get("optimizer") is not None:
        optimizer.load_state_dict(model, rest["optimizer"])
    # the schedule position is the scheduler's own count (one step per SOLVER.ITER_SIZE group, engine/trainer.py:86-91),
    # NOT the iteration index: with ITER_SIZE > 1 the two differ, and a learning-rate factor taken from the micro-iteration
    # count would cross a STEPS milestone early and trigger a spurious momentum correction on the next step
    sched = rest.get("scheduler")
    iter_size = max(1, int(optimizer.cfg.SOLVER.ITER_SIZE))
    derived = (iteration + iter_size - 1) // iter_size
    if sched is not None and "last_epoch" in sched:
        position = int(sched["last_epoch"])
        # Checkpoints written before the "unit" marker existed stored the micro-iteration index in last_epoch.  With
        # ITER_SIZE > 1 that places the schedule ITER_SIZE times too far (an early STEPS milestone, a spurious momentum
        # correction): a position no run of `iteration` iterations can have reached falls back to the derived one.
        if sched.get("unit") != "sched_steps" and iteration > 0 and position > derived:
            import logging
            logging.getLogger("od_wscl_amd").warning(
                "checkpoint: scheduler.last_epoch = %d is not a scheduler-step count for iteration %d with ITER_SIZE %d "
                "(an older checkpoint format); resuming the schedule at %d", position, iteration, iter_size, derived)
            position = derived
    else:
        position = derived
    optimizer.sync_from_params(model)
    optimizer.resume(position)
    return iteration
