"""`cfg` with the reference's key names for the hot path (wetectron/config/defaults.py).

A small attribute-dict with yacs' surface (merge_from_file / merge_from_list /
freeze / clone); the reference's shipped yaml files (configs/voc/*.yaml,
configs/coco/*.yaml) merge unchanged -- keys the hot path does not read are
stored, not rejected."""
from .node import CfgNode
from .defaults import make_defaults

cfg = make_defaults()

__all__ = ["cfg", "CfgNode", "make_defaults"]
