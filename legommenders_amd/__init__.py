"""legommenders_amd -- MI355X-native (gfx950) training hot path for Legommenders' two-tower
content recommenders (NAML / NRMS): HIP kernels behind a C ABI (include/lego_hip.h) plus the
host-side mirror of the reference's operator / predictor plug-in interface."""
__version__ = "0.1.0"

import os as _os

# The training step runs on three HIP streams (main, weight-gradient side stream, prefetch stream).  The ROCm runtime maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); once RCCL has created its own streams under torch.distributed
# two of ours land on ONE queue and run back to back: the prefetch chain (sample, plan, gather, keep bits, 83 us) then sits on
# the main queue and a data-parallel step takes 0.78 instead of 0.705 ms (profiles/r02_hw_queues.txt).  Eight queues keep them
# apart.  Read by the runtime when HIP initialises, so it has to be in the environment before the first torch.cuda call; a value
# set by the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
