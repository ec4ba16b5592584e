"""Model factory with the reference's interface (models/__init__.py:4-34).

Only the four networks on the cost-volume path exist here; ``dispnet`` (no cost volume)
and the ``*_m1/_m2`` names (whose files are absent from the reference too) raise."""

dict_models = ["dispnetcorr", "iresnet", "gcnet", "psmnet"]


def model_create_by_name(name_model, maxdisparity=192):
    if name_model == "psmnet":
        from .psmnet.stackhourglass import PSMNet
        return PSMNet(maxdisparity)
    if name_model == "gcnet":
        from .gcnet import gcnet
        return gcnet(maxdisparity)
    if name_model == "dispnetcorr":
        from .dispnetcorr import dispnetcorr
        return dispnetcorr(maxdisparity)
    if name_model == "iresnet":
        from .iresnet import iresnet
        return iresnet(maxdisparity)
    raise AssertionError("model %r is not on the MI355X cost-volume path" % (name_model,))
