"""Drop-in wiring for code written against the reference's package names.

``install()`` makes ``fuxictr.pytorch.models.RAT_m2`` (and ``RAT_m0``, ``RAT_m1``, ``RAT_m3``) resolve to the HIP-backed plugins:
  * if the real FuxiCTR fork is importable it only swaps those attributes (every other model stays the reference's);
  * otherwise it registers a minimal ``fuxictr`` namespace (version 1.2.3, ``fuxictr.pytorch.models``,
    ``fuxictr.pytorch.torch_utils.seed_everything``, ``fuxictr.features.FeatureMap``, ``fuxictr.utils``) backed by
    this package, which is all the reference's ``run_expid.py`` touches on the RAT_m2 path."""
import sys
import types


def install():
    from . import base_model, config, features, models
    try:
        import fuxictr.pytorch.models as ref_models          # the real fork, if present
        ref_models.RAT_m2 = models.RAT_m2
        ref_models.RAT_m1 = models.RAT_m1
        ref_models.RAT_m3 = models.RAT_m3
        ref_models.RAT_m0 = models.RAT_m0
        return "patched"
    except Exception:
        pass
    fux = types.ModuleType("fuxictr")
    fux.__version__ = "1.2.3"
    pyt = types.ModuleType("fuxictr.pytorch")
    tu = types.ModuleType("fuxictr.pytorch.torch_utils")
    tu.seed_everything = base_model.seed_everything
    tu.get_device = base_model.get_device
    pyt.models, pyt.torch_utils = models, tu
    fux.pytorch, fux.features, fux.utils = pyt, features, config
    sys.modules.update({"fuxictr": fux, "fuxictr.pytorch": pyt, "fuxictr.pytorch.models": models,
                        "fuxictr.pytorch.torch_utils": tu, "fuxictr.features": features, "fuxictr.utils": config})
    return "registered"
