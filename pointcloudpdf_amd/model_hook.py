"""Forward/backward hook tap feeding the recognizer -- pointcept/models/utils/model_hook.py:18-200.

Same configuration surface (``hook_config`` = {module name: ["forward_output", "backward_outputGrad", ...]},
``clone_tensor``, ``exclude_clone``), same result layout (``hooks[name]["forward_output"]``), same context-manager
use.  One addition: when a captured coordinate tensor is cloned, its Geometry tag travels with the clone, so the
U-decoder reuses the batch's interpolation tables instead of re-running kNN (results are identical either way).
"""
from collections import defaultdict

import torch

from .geometry import propagate_tag
from .registry import MODELHOOKS

_ACTION_TO_HOOK = {"inputGrad": "backward", "outputGrad": "backward", "input": "forward", "output": "forward"}


class _SilentLogger:
    def info(self, *a, **k): pass
    def debug(self, *a, **k): pass
    def warning(self, *a, **k): pass
    def error(self, *a, **k): pass
    def exception(self, *a, **k): pass


@MODELHOOKS.register_module("ModelHook")
class BaseModelHook:
    def __init__(self, hook_config, clone_tensor=True, exclude_clone=None, logger=None):
        self.clone_tensor = clone_tensor
        self.exclude_clone = exclude_clone or {}
        self.logger = logger or _SilentLogger()
        self.model = None
        self.model_name = "Model"
        self.output = defaultdict(dict)
        self.hooks = defaultdict(dict)
        self._clone = {}
        self._hooked = set()
        self._registered = False
        for module_name, actions in hook_config.items():
            for action_str in ([actions] if isinstance(actions, str) else actions):
                hook, action = action_str.split("_")
                assert hook == _ACTION_TO_HOOK.get(action), f"Invalid hook-action combo: {action_str}"
                self.hooks[module_name][action_str] = None
                self.output[module_name][action_str] = None
                self._clone[(module_name, action_str)] = clone_tensor and action_str not in self.exclude_clone.get(module_name, [])

    def set_model(self, model):
        self.model = model
        return self

    def set_logger(self, logger):
        self.logger = logger or _SilentLogger()

    def register_hooks(self, model):
        if self._registered:
            self.logger.warning(f"Hooks already registered for {self.model_name}")
            return
        self.model = getattr(model, "module", model)
        seen = set()
        # (the hooks are entered and left around EVERY training step: walking named_modules() of the whole model each time cost 0.7 ms of
        #  host time per step; the name -> module table of the hooked names is kept per model object)
        cache = self.__dict__.get("_module_cache")
        if cache is None or cache[0] is not model:
            cache = (model, [(name, module) for name, module in model.named_modules() if name in self.hooks])
            self._module_cache = cache
        for name, module in cache[1]:
            if name in self.hooks and id(module) not in self._hooked:
                for action_str in self.hooks[name]:
                    hook, action = action_str.split("_")
                    reg = module.register_forward_hook if hook == "forward" else module.register_full_backward_hook
                    self.hooks[name][action_str] = reg(self._closure(name, hook, action))
                self._hooked.add(id(module))
                seen.add(name)
        self.model_name = model.__class__.__name__
        self._registered = True
        missing = set(self.hooks) - seen
        if missing:
            self.logger.warning(f"Hooks not registered for modules: {', '.join(sorted(missing))}")

    def remove_hooks(self):
        for per_module in self.hooks.values():
            for action_str, h in per_module.items():
                if h is not None:
                    h.remove()
                    per_module[action_str] = None
        self._hooked.clear()
        self._registered = False

    def _closure(self, module_name, hook, action):
        key = f"{hook}_{action}"
        clone = self._clone[(module_name, key)]

        def fwd(module, inp, out):
            self.output[module_name][key] = self._capture(inp if action == "input" else out, clone)

        def bwd(module, grad_in, grad_out):
            self.output[module_name][key] = self._capture(grad_in if action == "inputGrad" else grad_out, clone)

        return fwd if hook == "forward" else bwd

    def _capture(self, obj, clone):
        if isinstance(obj, tuple):
            items = tuple(self._capture(o, clone) for o in obj)
            return items[0] if len(items) == 1 else items
        if isinstance(obj, torch.Tensor):
            return propagate_tag(obj, obj.clone()) if clone else obj
        return obj  # lists ([p, x, o]) are captured by reference, as upstream

    def __getitem__(self, key):
        return self.output[key]

    def __enter__(self):
        if self.model is None:
            raise RuntimeError("Model not set. Call `set_model(model)` first.")
        self.register_hooks(self.model)
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.remove_hooks()
        return False
