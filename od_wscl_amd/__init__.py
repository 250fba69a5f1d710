"""od_wscl_amd -- MI355X-native (gfx950) implementation of OD-WSCL's
proposal-feature hot path, behind the reference's `wetectron.layers` /
`wetectron.modeling` operator API.  See DESIGN.md."""
__version__ = "0.1.0"

import os as _os

# One process driving one GPU: TWO hardware queues for its HIP streams instead of the runtime's default four.  The step uses
# three streams that wait on each other every few hundred microseconds (the step's, the optimiser's, the contrastive branch's);
# measured on two MI355X boxes, alternating runs of the driver's bench command (profiles/r06/ab_hwq*.txt): 1 queue 8.93 ms,
# 2 queues 8.53-8.66, 3 queues 8.74, 4 (default) 8.67-8.78, 8 queues 8.65-8.71 per step.  Read by the HIP runtime when it
# starts, so it only takes effect if this package is imported before the first device call; an explicit setting wins.  With
# more than one rank per job the default stays (RCCL's own streams want their queues).
if int(_os.environ.get("WORLD_SIZE", "1") or 1) == 1:
    if "GPU_MAX_HW_QUEUES" not in _os.environ:
        _os.environ["GPU_MAX_HW_QUEUES"] = "2"
        _os.environ["ODW_HWQ_DEFAULTED"] = "1"        # (so that a rank launched FROM this process does not inherit the default)
elif _os.environ.get("ODW_HWQ_DEFAULTED") == "1":
    _os.environ.pop("GPU_MAX_HW_QUEUES", None)
    _os.environ.pop("ODW_HWQ_DEFAULTED", None)
