"""Plug-in discovery (mirror of the reference's loader/class_hub.py:47-165): every `*_operator.py` /
`*_predictor.py` next to the base class is imported and each subclass is registered under its lower-cased
class name minus the suffix, so `meta.item: CNN`, `meta.user: Ada`, `meta.predictor: Dot` resolve exactly as
in the reference.  The glob is anchored at this package (the reference uses a cwd-relative glob,
class_hub.py:54,63)."""
from __future__ import annotations

import glob
import importlib
import os
from typing import Dict, List, Type

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class ClassHub:
    @staticmethod
    def operators() -> "ClassHub":
        from legommenders_amd.model.operators.base_operator import BaseOperator
        return ClassHub(BaseOperator, os.path.join("model", "operators"), "Operator")

    @staticmethod
    def predictors() -> "ClassHub":
        from legommenders_amd.model.predictors.base_predictor import BasePredictor
        return ClassHub(BasePredictor, os.path.join("model", "predictors"), "Predictor")

    def __init__(self, base_class: Type, module_dir: str, module_type: str):
        self.base_class = base_class
        self.module_dir = module_dir
        self.module_type = module_type.lower()
        self.upper_module_type = self.module_type[0].upper() + self.module_type[1:]
        self.class_list: List[Type] = self._get_class_list()
        self.class_dict: Dict[str, Type] = {
            c.__name__.replace(self.upper_module_type, "").lower(): c for c in self.class_list}

    def _get_class_list(self):
        out = []
        for path in sorted(glob.glob(os.path.join(_PKG, self.module_dir, f"*_{self.module_type}.py"))):
            name = os.path.splitext(os.path.basename(path))[0]
            mod = importlib.import_module(".".join(["legommenders_amd"] + self.module_dir.split(os.sep) + [name]))
            for obj in mod.__dict__.values():
                if isinstance(obj, type) and issubclass(obj, self.base_class) and obj is not self.base_class:
                    if obj not in out:
                        out.append(obj)
        return out

    def __call__(self, name: str):
        return self.class_dict[name.lower()]

    __getitem__ = __call__

    def __contains__(self, name: str) -> bool:
        return name.lower() in self.class_dict

    def list(self):
        return list(self.class_dict.keys())
