"""
torchoptics_amd -- MI355X-native differentiable sequential ray tracer.

Drop-in for the hot path of OceanT-shirt/TorchOptics (`torchlens.ray_tracing_lite`,
`torchlens.lens_modeling`): same Python API, the per-surface loop replaced by hand-written
HIP kernels for gfx950 behind a C ABI (include/tl_trace.h).  See DESIGN.md.
"""
from . import graphs, lens_modeling, metrics, paraxial, ray_tracing  # noqa: F401
from .lens_modeling import Lens, Specs, Structure  # noqa: F401
from .ops import get_default_mode, set_default_mode  # noqa: F401
from .ray_tracing import RayTracer, compute_rms2d, compute_rms2d_batch, trace_skew  # noqa: F401

__version__ = "0.1.0"
