"""raytracer_amd -- MI355X-native stochastic progressive photon mapping (the SPPM hot path of
0V/raytracer) behind a C ABI (include/sppm_hip.h).  This package is the Python host-side mirror of
the reference's interface; the compute is in raytracer_amd/libsppm_hip.so (hand-written HIP for
gfx950).  There is no CPU fallback: importing works anywhere, creating a context needs a GPU."""
from .scene import CORNELL_SPHERES, DIFFUSE, SPECULAR, REFRACTION  # noqa: F401
from .sppm import StochasticPpm, SppmContext, progressive_render, save_hdr_file, save_ppm_file  # noqa: F401
