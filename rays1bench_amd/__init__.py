"""rays1bench_amd — MI355X-native implementation of rays1bench's step13 path-tracing hot path.

The product is native: `csrc/` holds the hand-written HIP kernels (gfx950) and the C-ABI
(`include/rays1.h`) in `lib/librays1.so`, plus `lib/rayweek1_hip`, the drop-in host program
with the reference's own entry points (create_*_scene / benchmark / main -w -n).  This Python
package is only the thin host-side binding used by tests and bench.py: it mirrors the
reference's interface for the path — `create_small_scene()`, `create_medium_scene()`,
`create_large_scene()`, `benchmark(scene, ...)` — and forwards everything to the C-ABI.
There is no CPU fallback: if librays1.so or a HIP device is missing, calls raise.
"""
from .binding import (  # noqa: F401
    R1Error, Renderer, Scene, Params, RESULT, benchmark, build, create_grid_scene, create_large_scene,
    create_medium_scene, create_small_scene, device_count, lib, lib_path, log_results, make_params, tga_write_rgb24,
)
