"""bella_amd: MI355X-native overlap detection (BELLA's SpGEMM + X-drop hot path) behind a C ABI.

The compute lives in csrc/ (hand-written gfx950 HIP kernels, built into libbella_hip.so); this package is the
host-side mirror of the reference's call surface for that path.  Nothing here computes on the CPU."""
from .api import BellaPars, BellaHipError, Engine, hash_spgemm  # noqa: F401
