"""Install the MI355X path behind the reference's plugin surface."""
import sys


def install_pointops():
    """Make ``import pointops`` (libs/pointops/functions/__init__.py:1-14 names) and ``import pointops2.pointops``
    (the window-attention part, libs/pointops2/functions/pointops.py) resolve to this package's drop-ins."""
    from . import pointops, pointops2

    sys.modules["pointops"] = pointops
    # libs/pointops2: ``import pointops2.pointops as pointops`` (stratified_transformer_v1m1_origin.py:21)
    sys.modules["pointops2"] = pointops2
    sys.modules["pointops2.pointops"] = pointops2.pointops
    return pointops


def register_into_pointcept(force=True):
    """Register our classes under the reference's names in a live pointcept install's registries."""
    from pointcept.models.builder import MODELS as PC_MODELS  # noqa: import error = pointcept not importable
    from . import point_transformer, recognizer, segmentor, model_hook, stratified  # noqa: F401  (fills the registries)
    from .registry import MODELS, RECOGNIZER, MODELHOOKS

    for name in ["PointTransformer-Seg26", "PointTransformer-Seg38", "PointTransformer-Seg50",
                 "PointTransformer-Recognizer", "ST-v1m1", "ST-v1m1-Recognizer", "DefaultSegmentor"]:
        PC_MODELS.register_module(name=name, force=force, module=MODELS.get(name))
    try:
        from pointcept.recognizers.builder import RECOGNIZER as PC_REC

        for name in ["PointPdf-v1m1", "MaxProbability"]:
            PC_REC.register_module(name=name, force=force, module=RECOGNIZER.get(name))
    except ImportError:
        pass
    try:
        from pointcept.models.utils.model_hook import MODELHOOKS as PC_HOOKS

        PC_HOOKS.register_module(name="ModelHook", force=force, module=MODELHOOKS.get("ModelHook"))
    except ImportError:
        pass
