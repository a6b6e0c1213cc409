"""srganst - MI355X-native SRGAN-ST training hot path (host side).

Mirrors the reference's Python surface (config.Config, model.Generator/Discriminator,
loss.StructureTensorLoss, train/warmup/validate) over the C ABI of libsrganst.so
(include/srganst.h).  PyTorch is plumbing here: device memory, streams, torch.distributed.
"""
__version__ = "0.1.0"
