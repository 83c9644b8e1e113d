"""Drop-in for the window-attention part of the reference's ``pointops2.pointops`` module
(libs/pointops2/functions/pointops.py) -- SURVEY.md 8 row f-1.  ``import pointops2.pointops as pointops`` in
stratified_transformer_v1m1_origin.py:21 resolves to :mod:`pointcloudpdf_amd.pointops2.pointops`."""
from . import pointops  # noqa: F401
