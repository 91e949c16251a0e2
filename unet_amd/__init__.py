"""MI355X-native U-Net segmentation hot path."""


def _arch(name):
    def f(pretrained=False, **kw):     # fastai passes architectures as callables (params_and_main.py:99: ARCHITECTURE = xresnet34)
        raise RuntimeError(f"{name} is an architecture token; HipDynamicUnet builds the encoder itself")
    f.__name__ = name
    return f


# the constructors the reference imports (params_and_main.py:12)
xresnet18, xresnet34, xresnet50 = _arch("xresnet18"), _arch("xresnet34"), _arch("xresnet50")
xresnet101, xresnet34_deep = _arch("xresnet101"), _arch("xresnet34_deep")
