"""Plugin registry with the reference's VO-model API
(/root/reference/pointnav_vo/utils/baseline_registry.py:81-109: register_vo_model / get_vo_model,
register_vo_engine / get_vo_engine).  When habitat is installed, the reference's own ``baseline_registry`` object
can be handed to :func:`install_into` so the HIP models replace the stock ones under the same names
(see INTEGRATION.md)."""
from typing import Optional


class BaselineRegistry:
    mapping = {}

    @classmethod
    def _register_impl(cls, _type, to_register, name, assert_type=None):
        def wrap(to_register):
            if assert_type is not None:
                assert issubclass(to_register, assert_type), f"{to_register} must be a subclass of {assert_type}"
            cls.mapping.setdefault(_type, {})[name if name is not None else to_register.__name__] = to_register
            return to_register

        return wrap if to_register is None else wrap(to_register)

    @classmethod
    def _get_impl(cls, _type, name):
        return cls.mapping.get(_type, {}).get(name, None)

    @classmethod
    def register_vo_model(cls, to_register=None, *, name: Optional[str] = None):
        return cls._register_impl("vo_model", to_register, name)

    @classmethod
    def get_vo_model(cls, name):
        return cls._get_impl("vo_model", name)

    @classmethod
    def register_policy(cls, to_register=None, *, name: Optional[str] = None):
        """baseline_registry.py: register_policy / get_policy (ddppo_trainer.py:115)."""
        return cls._register_impl("policy", to_register, name)

    @classmethod
    def get_policy(cls, name):
        return cls._get_impl("policy", name)

    @classmethod
    def register_vo_engine(cls, to_register=None, *, name: Optional[str] = None):
        return cls._register_impl("vo_engine", to_register, name)

    @classmethod
    def get_vo_engine(cls, name):
        return cls._get_impl("vo_engine", name)


baseline_registry = BaselineRegistry()


def install_into(other_registry):
    """Re-register every HIP VO model into another registry object that exposes ``register_vo_model``
    (the reference's ``pointnav_vo.utils.baseline_registry.baseline_registry``), overriding the stock classes."""
    for name, cls_ in BaselineRegistry.mapping.get("vo_model", {}).items():
        other_registry.register_vo_model(cls_, name=name)
    if hasattr(other_registry, "register_policy"):
        for name, cls_ in BaselineRegistry.mapping.get("policy", {}).items():
            other_registry.register_policy(cls_, name=name)
