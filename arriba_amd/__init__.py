"""arriba_amd -- MI355X-native hot path of a chimeric-read fusion caller (drop-in for suhrig/arriba's
read_chimeric_alignments -> find_fusions -> filter cascade).  See DESIGN.md and include/arriba_gpu.h."""
from ._capi import FILTER_NAMES  # noqa: F401
from .pipeline import ArribaError, DevicePipeline, HostSession  # noqa: F401
