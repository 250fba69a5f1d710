"""Name -> factory registries with the reference's registry names
(wetectron/modeling/registry.py:9-19, utils/registry.py)."""


class Registry(dict):
    def register(self, name):
        def deco(fn):
            self[name] = fn
            return fn
        return deco


BACKBONES = Registry()
ROI_BOX_FEATURE_EXTRACTORS = Registry()
ROI_WEAK_PREDICTOR = Registry()
ROI_WEAK_LOSS = Registry()
