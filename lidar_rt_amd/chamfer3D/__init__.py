"""MI355X Chamfer distance (counterpart of the reference's lib/utils/chamfer3D package)."""
from .dist_chamfer_3D import chamfer_3DDist, chamfer_3DFunction  # noqa: F401
