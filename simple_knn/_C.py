"""`simple_knn._C.distCUDA2` -> MI355X implementation (das3r_amd.knn)."""
from das3r_amd.knn import distCUDA2  # noqa: F401
