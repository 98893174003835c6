"""Tiny stand-in for gym's env registry (gym is not installed in this image): `make('Q1PhysEnv-v0')`
constructs what reference env.py:516-521 registers."""
_REGISTRY = {}


def register(env_id, entry_point, kwargs=None):
    _REGISTRY[env_id] = (entry_point, dict(kwargs or {}))


def make(env_id, **overrides):
    if env_id not in _REGISTRY:
        raise KeyError(f"unknown env id {env_id!r}; registered: {sorted(_REGISTRY)}")
    entry, kwargs = _REGISTRY[env_id]
    kwargs = {**kwargs, **overrides}
    return entry(**kwargs)


def registered():
    return dict(_REGISTRY)
