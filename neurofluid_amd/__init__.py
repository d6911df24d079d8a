"""neurofluid_amd — MI355X-native implementation of NeuroFluid's data-parallel hot path.

Host-side mirror of the reference's operator interface (models/renderer.py RenderNet,
models/nerf.py NeRF/Embedding, models/transmodel.py ParticleNet, utils/ray_utils.py) over
hand-written HIP kernels behind a C ABI (include/neurofluid_hip.h).  No CPU / PyTorch fallback:
importing works anywhere, running needs the built library and a gfx950 GPU.
"""
__version__ = "0.1.0"

from ._hostcpu import effective_cpus, limit_host_threads  # noqa: E402,F401

limit_host_threads()
