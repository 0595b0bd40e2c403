"""iodine_amd: MI355X-native (gfx950) IODINE refinement step behind the reference's
``IODINE.forward / reconstruct`` module API.  Heavy imports are lazy so that
``iodine_amd.synth`` can be used without the HIP library being built."""

__all__ = ['IODINE', 'synth', 'parallel', 'optim', 'ari', 'checkpoint', 'data', 'engine']


def __getattr__(name):
    if name == 'IODINE':
        from .model import IODINE
        return IODINE
    raise AttributeError(name)
