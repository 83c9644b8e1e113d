"""Name -> class registries with the reference's build contract (pointcept/utils/registry.py:9-56:
``REG.build(dict(type="Name", **kwargs))`` -> ``cls(**kwargs)``), so reference config dicts build our classes.
The same classes can also be registered into a live pointcept install (see integration.py / INTEGRATION.md)."""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._classes = {}

    @property
    def name(self):
        return self._name

    def __contains__(self, key):
        return key in self._classes

    def __len__(self):
        return len(self._classes)

    def get(self, key):
        return self._classes.get(key)

    def register_module(self, name=None, force=False, module=None):
        """Decorator (``@REG.register_module("Name")``) or direct call (``module=cls``), as upstream."""
        if isinstance(name, type):  # used as a bare decorator
            cls, name = name, None
            self._register(cls, None, force)
            return cls

        def deco(cls):
            self._register(cls, name, force)
            return cls

        if module is not None:
            return deco(module)
        return deco

    def _register(self, cls, name, force):
        if not inspect.isclass(cls):
            raise TypeError(f"module must be a class, but got {type(cls)}")
        names = [cls.__name__] if name is None else ([name] if isinstance(name, str) else list(name))
        for n in names:
            if not force and n in self._classes:
                raise KeyError(f"{n} is already registered in {self._name}")
            self._classes[n] = cls

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict):
            raise TypeError(f"cfg must be a dict, but got {type(cfg)}")
        args = dict(cfg)
        for k, v in (default_args or {}).items():
            args.setdefault(k, v)
        if "type" not in args:
            raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}\n{default_args}')
        obj_type = args.pop("type")
        if isinstance(obj_type, str):
            cls = self.get(obj_type)
            if cls is None:
                raise KeyError(f"{obj_type} is not in the {self._name} registry")
        elif inspect.isclass(obj_type):
            cls = obj_type
        else:
            raise TypeError(f"type must be a str or valid type, but got {type(obj_type)}")
        try:
            return cls(**args)
        except Exception as e:  # same courtesy as upstream: say which class failed
            raise type(e)(f"{cls.__name__}: {e}")


MODELS = Registry("models")
LOSSES = Registry("losses")
RECOGNIZER = Registry("recognizer")
MODELHOOKS = Registry("modelhook")


def build_model(cfg):
    return MODELS.build(cfg)
