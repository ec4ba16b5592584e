"""MI355X-native stereo cost-volume path of DSMnet."""


def refold():
    """Drop every cached packed weight / folded BatchNorm affine; they are re-made at the next
    forward.  Needed only after in-place parameter edits made through ``.data`` (which do not
    bump a tensor's version counter); see ``blocks3d.invalidate_folded_caches``."""
    from .blocks3d import invalidate_folded_caches
    invalidate_folded_caches()
