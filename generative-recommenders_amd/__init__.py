"""MI355X-native Hydra hot path: drop-in modules over hand-written gfx950 kernels (libhmx.so).

    from generative_recommenders_amd import hmx            # ctypes binding of the C ABI
    from generative_recommenders_amd import ops            # torch.autograd wrappers
    from generative_recommenders_amd.hydra import Hydra    # reference-compatible modules
"""
__version__ = "0.1.0"
