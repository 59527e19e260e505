"""checkm_amd -- MI355X-native marker-gene scan + marker-set reduction behind CheckM's own interfaces."""
import os as _os

# The rare stages launch one kernel per register class on up to 8 streams; the HIP runtime's default of 4 hardware queues
# would serialise half of them.  Must be in the environment before the runtime initialises (first HIP call of the process).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
