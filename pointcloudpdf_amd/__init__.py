"""pointcloudpdf_amd -- MI355X-native PointTransformer-V1 / PDF U-decoder hot path of JinfengX/PointCloudPDF.

Python host code on PyTorch-ROCm over a C-ABI HIP library (include/pdfops.h, csrc/*.hip, gfx950 only).
"""
__version__ = "0.1.0"
