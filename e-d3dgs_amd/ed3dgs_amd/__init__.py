"""ed3dgs_amd -- helpers of the MI355X-native E-D3DGS hot path (loader of the HIP C-ABI library, synthetic inputs)."""
