"""simple_knn._C of the reference (submodules/simple-knn/ext.cpp:15, spatial.cu:14-25) over the C ABI."""
from ed3dgs_amd.knn import distCUDA2  # noqa: F401
