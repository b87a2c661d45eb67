"""legommenders_amd -- MI355X-native (gfx950) training hot path for Legommenders' two-tower
content recommenders (NAML / NRMS): HIP kernels behind a C ABI (include/lego_hip.h) plus the
host-side mirror of the reference's operator / predictor plug-in interface."""
__version__ = "0.1.0"
