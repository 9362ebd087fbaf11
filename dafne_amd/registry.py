"""Registries with detectron2's names and API (register()/get()).

The reference resolves its classes through META_ARCH_REGISTRY["OneStageDetector"]
(dafne/modeling/one_stage_detector.py:34), BACKBONE_REGISTRY[
"build_dafne_resnet_fpn_backbone"] (dafne/modeling/backbone/fpn.py:58) and
PROPOSAL_GENERATOR_REGISTRY["DAFNe"] (dafne/modeling/dafne/dafne.py:69).  When
detectron2 is importable we register into ITS registries so cfg-driven
build_model() picks this engine; otherwise a look-alike is used.
"""


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, "'%s' already registered in '%s'" % (name, self._name)
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(f):
                self._do_register(f.__name__, f)
                return f
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def get(self, name):
        if name not in self._obj_map:
            raise KeyError("No object named '%s' found in '%s' registry!" % (name, self._name))
        return self._obj_map[name]

    def __contains__(self, name):
        return name in self._obj_map


try:  # pragma: no cover - detectron2 is absent in the build image
    from detectron2.modeling.backbone.build import BACKBONE_REGISTRY
    from detectron2.modeling.meta_arch.build import META_ARCH_REGISTRY
    from detectron2.modeling.proposal_generator.build import PROPOSAL_GENERATOR_REGISTRY
except Exception:  # noqa: BLE001
    META_ARCH_REGISTRY = Registry("META_ARCH")
    BACKBONE_REGISTRY = Registry("BACKBONE")
    PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")


def build_model(cfg):
    """Counterpart of detectron2.modeling.build_model for this engine."""
    return META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
