"""Install the MI355X path behind the reference's plugin surface."""
import sys


def install_pointops():
    """Make ``import pointops`` resolve to this package's drop-in (libs/pointops/functions/__init__.py:1-14 names)."""
    from . import pointops

    sys.modules["pointops"] = pointops
    return pointops


def register_into_pointcept(force=True):
    """Register our classes under the reference's names in a live pointcept install's registries."""
    from pointcept.models.builder import MODELS as PC_MODELS  # noqa: import error = pointcept not importable
    from . import point_transformer, recognizer, segmentor, model_hook
    from .registry import MODELS, RECOGNIZER, MODELHOOKS

    for name in ["PointTransformer-Seg26", "PointTransformer-Seg38", "PointTransformer-Seg50",
                 "PointTransformer-Recognizer", "DefaultSegmentor"]:
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
