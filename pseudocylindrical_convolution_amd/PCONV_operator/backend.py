"""Which native modules sit under the operator layer.

The reference's wrappers do `import PCONV` / `import coder` at import time.  Here
the two modules are looked up when an operator is constructed, so the same
nn.Module layer can be driven by the HIP shim (the product, the default) or by a
module that a TEST injects with `use(...)` -- the parity tests run the graph a
second time on the CPU oracle this way.  The product never calls `use`.
"""
_native_ops = None
_native_coder = None


def use(ops_module=None, coder_module=None):
    """Override the native modules (test / baseline harnesses only)."""
    global _native_ops, _native_coder
    if ops_module is not None:
        _native_ops = ops_module
    if coder_module is not None:
        _native_coder = coder_module


def reset():
    global _native_ops, _native_coder
    _native_ops = None
    _native_coder = None


def ops():
    """The module that provides the 21 `PCONV.*Op` classes."""
    global _native_ops
    if _native_ops is None:
        from .. import PCONV as hip_ops  # raises at first use when libpconv_hip.so is absent
        _native_ops = hip_ops
    return _native_ops


def coder():
    """The module that provides `coder.coder`."""
    global _native_coder
    if _native_coder is None:
        from .. import coder as native_coder
        _native_coder = native_coder
    return _native_coder


def device_of(gid):
    """torch device string an op keyed by `gid` works on under the active backend."""
    mod = ops()
    return getattr(mod, "DEVICE_FMT", "cuda:{}").format(gid)


# -- derived-parameter caches ---------------------------------------------------
# Packed convolution slabs, the effective GDN parameters and the entropy engine's
# repacked weights are cached per parameter tensor, keyed on (data_ptr, _version).
# A write through `.data` does not bump `_version`, so every cache key also
# carries this epoch: `invalidate_derived()` drops them all at once, and the codec
# modules call it from a load_state_dict post-hook (`watch_state_dict`).
_param_epoch = [0]


def param_epoch():
    return _param_epoch[0]


def invalidate_derived():
    """Drop every cache derived from module parameters (call after editing a
    parameter through `.data`; load_state_dict does it by itself)."""
    _param_epoch[0] += 1


def watch_state_dict(module):
    """make `module.load_state_dict` invalidate the derived caches"""
    module.register_load_state_dict_post_hook(lambda mod, incompatible: invalidate_derived())
    return module
