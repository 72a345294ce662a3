"""Plug-in registries with the reference's contract (apps/api/src/register/__init__.py:8-290):
`FunctionRegister` — decorator registration by key with an `available` flag, `call(*args, key=None)`
dispatching to the default key, KeyError on duplicate keys unless overwrite, RuntimeError when the
key is unavailable; `ClassRegister` — the same for component classes (`TRANSFORMERS_REGISTRY`).

When this backend is dropped into the reference tree the reference's own registries are used
(INTEGRATION.md); these mirrors let the backend be exercised stand-alone with identical semantics.
"""
from __future__ import annotations

from typing import Any, Dict, Optional


class _Registry:
    def __init__(self, *, allow_overwrite: bool = False):
        self._store: Dict[str, Dict[str, Any]] = {}
        self._allow_overwrite = allow_overwrite

    def __call__(self, key_or_obj=None, *, overwrite: Optional[bool] = None, available: bool = True):
        if callable(key_or_obj) and not isinstance(key_or_obj, str) and overwrite is None:
            self._add(key_or_obj.__name__, key_or_obj, self._allow_overwrite, available)
            return key_or_obj
        key = key_or_obj

        def deco(obj):
            self._add(key if key is not None else obj.__name__, obj,
                      self._allow_overwrite if overwrite is None else overwrite, available)
            return obj

        return deco

    def _add(self, key: str, obj, allow: bool, available: bool):
        if not allow and key in self._store:
            raise KeyError(f"Key '{key}' already registered. Use overwrite=True to replace.")
        self._store[key] = {"obj": obj, "available": available}

    def get(self, key: str):
        if key not in self._store:
            raise KeyError(f"Key '{key}' not found in registry.")
        return self._store[key]["obj"]

    def is_available(self, key: str) -> bool:
        return key in self._store and self._store[key]["available"]

    def set_availability(self, key: str, available: bool):
        if key not in self._store:
            raise KeyError(f"Key '{key}' not found in registry.")
        self._store[key]["available"] = available

    def all(self):
        return {k: v["obj"] for k, v in self._store.items()}

    def all_available(self):
        return {k: v["obj"] for k, v in self._store.items() if v["available"]}

    def set_default(self, key: str):
        self._default = key

    def get_default(self) -> str:
        return self._default

    __getitem__ = get

    def __iter__(self):
        return iter(self._store)

    def __len__(self):
        return len(self._store)

    def __contains__(self, key):
        return key in self._store


class FunctionRegister(_Registry):
    def call(self, *args, key: Optional[str] = None, **kwargs):
        if key is None and hasattr(self, "_default"):
            key = self._default
        if not self.is_available(key):
            raise RuntimeError(f"Function '{key}' is not available.")
        return self.get(key)(*args, **kwargs)


class ClassRegister(_Registry):
    def create(self, key: str, *args, **kwargs):
        if not self.is_available(key):
            raise RuntimeError(f"Class '{key}' is not available.")
        return self.get(key)(*args, **kwargs)


attention_register = FunctionRegister()
TRANSFORMERS_REGISTRY = ClassRegister()
VAE_REGISTRY = ClassRegister()
